#!/usr/bin/env python3
"""Per-kernel resource table from the compiler's reports (flatquant_amd/csrc/build/*.res, written by the Makefile):
tools/kernel_resources.py [> profiles/rNN_kernel_resources.txt].  parse() is also what tests/test_host_cpu.py checks."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = {"VGPRs": "vgpr", "AGPRs": "agpr", "Occupancy [waves/SIMD]": "occupancy", "VGPRs Spill": "vgpr_spill",
          "SGPRs Spill": "sgpr_spill", "LDS Size [bytes/block]": "lds", "ScratchSize [bytes/lane]": "scratch"}


def demangle(names):
    """'_ZN12_GLOBAL__N_119fq_kron_trio_kernelILi4ELb0ELb0EEEvPK...' -> 'fq_kron_trio_kernel<4,0,0>' (integral template arguments
    only, which is all these kernels have; c++filt does not know the _Float16 mangling of the parameter lists)."""
    out = []
    for n in names:
        m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n) or re.match(r"_Z(\d+)", n)
        if not m:
            out.append(n)
            continue
        start = m.end()
        ln = int(m.group(1))
        name, rest = n[start:start + ln], n[start + ln:]
        args = []
        if rest.startswith("I"):
            rest = rest[1:]
            while True:
                a = re.match(r"L[ibjlmxy](n?)(\d+)E", rest)
                if a:
                    args.append(("-" if a.group(1) else "") + a.group(2))
                    rest = rest[a.end():]
                    continue
                t = re.match(r"DF16(_|b)", rest)       # element type of the round-3 templates: _Float16 / __bf16
                if not t:
                    break
                args.append("f16" if t.group(1) == "_" else "bf16")
                rest = rest[t.end():]
        out.append(name + ("<" + ",".join(args) + ">" if args else ""))
    return out


def parse(build_dir=None):
    """-> {demangled kernel name (template arguments included, parameter list cut): {vgpr, agpr, occupancy, ...}}"""
    build_dir = build_dir or os.path.join(ROOT, "flatquant_amd", "csrc", "build")
    raw = {}
    for path in sorted(glob.glob(os.path.join(build_dir, "*.res"))):
        cur = None
        for line in open(path, errors="replace"):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = raw.setdefault(m.group(1), {"file": os.path.basename(path)[:-4]})
                continue
            m = re.search(r"remark:\s+([A-Za-z\[\]/ ]+): (\S+) \[-Rpass", line)
            if m and cur is not None and m.group(1).strip() in FIELDS:
                v = m.group(2)
                cur[FIELDS[m.group(1).strip()]] = int(v) if v.lstrip("-").isdigit() else v
    names = list(raw)
    out = {}
    for mangled, nice in zip(names, demangle(names)):
        out[nice] = raw[mangled]
    return out


if __name__ == "__main__":
    res = parse()
    print(f"{'kernel':70s} {'file':18s} VGPR AGPR occ spill  LDS")
    for name, r in sorted(res.items(), key=lambda kv: (kv[1]["file"], kv[0])):
        print(f"{name[:70]:70s} {r['file']:18s} {r.get('vgpr', -1):4d} {r.get('agpr', -1):4d} {r.get('occupancy', -1):3d} "
              f"{r.get('vgpr_spill', -1):5d} {r.get('lds', -1):6d}")
