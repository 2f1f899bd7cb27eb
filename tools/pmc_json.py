#!/usr/bin/env python3
"""profiles/rNN_kron64_pmc.json from a tools/prof.sh summary (the file bench.py reads `roofline.traffic` from):
    tools/pmc_json.py profiles/r04_kron64_prof_summary.txt r04 > profiles/r04_kron64_pmc.json
HBM bytes per launch of the dominant kernel = 2 x FETCH_SIZE + WRITE_SIZE (KiB; gfx950 reports a wide coalesced read stream at half its
size: MI355X_MICROARCH.md, HBM / rocprofv3 section), from the separate --pmc passes; the kernel-trace line of the same command next to it."""
import json
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
tag = sys.argv[2] if len(sys.argv) > 2 else "rNN"
TOKENS, D = 16384, 4096
vals, trace = {}, None
for l in txt:
    if "fq_kron64_kernel" not in l:
        continue
    m = re.search(r"calls=\s*(\d+)\s+avg_ns=\s*([\d.]+)\s+min_ns=\s*([\d.]+)\s+max_ns=\s*([\d.]+)", l)
    if m:
        trace = {"calls": int(m.group(1)), "avg_ns": float(m.group(2)), "min_ns": float(m.group(3)), "max_ns": float(m.group(4))}
        continue
    m = re.search(r"\s(\S+)\s+avg=\s*([\d.]+)\s+n=(\d+)", l)
    if m:
        vals[m.group(1)] = float(m.group(2))
alg = TOKENS * (2 * D + D // 2 + 2)
out = {
    "kernel": "fq_kron64_kernel<FQ_OUT_PACKED, f16> (prepared 16 KB fragment image)",
    "workload": "C2: 16384 tokens x d=4096, packed out",
    "source": f"tools/prof.sh {tag} (rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline --no-sub-records`; --pmc in separate passes of "
              f"`bench.py --steps 200 --warmup 100 --no-cpu-baseline --no-sub-records`), see {tag}_kron64_prof_summary.txt; assembled by tools/pmc_json.py",
    "FETCH_SIZE_KB_avg": vals.get("FETCH_SIZE"),
    "WRITE_SIZE_KB_avg": vals.get("WRITE_SIZE"),
    "hbm_bytes_per_launch": int(round((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)) if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals else None,
    "note": "gfx950: FETCH_SIZE reports half of a wide coalesced read stream (MI355X_MICROARCH.md HBM) -> doubled; WRITE_SIZE as reported",
    "algorithmic_bytes_per_launch": alg,
    "kernel_trace": dict(trace or {}, command="rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-sub-records (default: 150 ms settle phase "
                                                "+ 200 warm-up + 1000 timed launches + the per-launch pass)"),
    "per_token": {k: round(vals[k] / TOKENS, 1) for k in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD") if k in vals},
    "sq": dict({k: vals[k] for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                                     "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE") if k in vals},
               unit="quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE (cycles); per launch, summed over the chip"),
}
print(json.dumps(out, indent=1))
