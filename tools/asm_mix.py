#!/usr/bin/env python3
"""Static instruction mix of one kernel from a hipcc -S --cuda-device-only listing:
   tools/asm_mix.py file.s <substring of the mangled kernel name> [--loop]   (counts per opcode, VALU / SALU / VMEM / LDS / MFMA totals)"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if l.startswith("_Z") and key in l.split(":")[0])
end = next(i for i in range(start, len(s)) if "s_endpgm" in s[i])
c = Counter()
for l in s[start + 1:end]:
    l = l.strip()
    if not l or l.startswith((".", ";")) or l.split()[0].endswith(":"):
        continue
    c[l.split()[0]] += 1
cls = Counter()
for op, n in c.items():
    k = ("MFMA" if "mfma" in op else "VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else
         "LDS" if op.startswith("ds_") else "VMEM" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    cls[k] += n
print(dict(cls), "total", sum(c.values()))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f"  {k:28s} {v}")
