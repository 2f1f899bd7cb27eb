#!/usr/bin/env python3
"""ISA check for kernels that load registers with loads the compiler does not track (inline-asm global_load_dwordx4 in
fq_kron_tall.hip): between such a load and the next s_waitcnt vmcnt(0) no instruction may READ or COPY the destination registers
(the data has not arrived; a register copy there — a phi, a spill, an AGPR move — silently takes the old contents).
    tools/check_untracked_loads.py file.s [kernel-name-substring]      exit code 1 and a listing if a hazard is found."""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2] if len(sys.argv) > 2 else ""
bad = 0
kernel = None
pending = {}      # register number -> line of the load
in_asm = False
for i, l in enumerate(src):
    t = l.strip()
    if l.startswith("_Z") and l.rstrip().endswith(":") or (l.startswith("_Z") and ":" in l.split(";")[0]):
        kernel = l.split(":")[0]
        pending = {}
    if kernel is None or key not in kernel:
        continue
    if t.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if t.startswith(";;#ASMEND"):
        in_asm = False
        continue
    if not t or t.startswith((";", ".")):
        continue
    m = re.match(r"global_load_dwordx[24] v\[(\d+):(\d+)\]", t)
    if in_asm and m:
        for r in range(int(m.group(1)), int(m.group(2)) + 1):
            pending[r] = i
        continue
    if "s_waitcnt" in t and "vmcnt(0)" in t:
        pending = {}
        continue
    if pending and not t.startswith(("s_", "global_load", "ds_")):
        ops = t.split(None, 1)[1] if " " in t else ""
        srcs = ops.split(",", 1)[1] if "," in ops else ""        # everything after the destination
        regs = set()
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", srcs):
            regs.update(range(int(a), int(b) + 1))
        regs.update(int(a) for a in re.findall(r"\bv(\d+)\b", srcs))
        hit = regs & set(pending)
        # the MFMAs of GEMM 1 read the OLD contents on purpose? No: GEMM 1 runs in front of the loads; any read here is a hazard
        if hit:
            bad += 1
            print(f"{kernel[:60]} line {i + 1}: `{t}` reads v{sorted(hit)[0]}.. loaded at line {pending[sorted(hit)[0]] + 1} before vmcnt(0)")
print("hazards:", bad)
sys.exit(1 if bad else 0)
