#!/usr/bin/env python3
"""Per-(kernel, grid) dispatch durations from a rocprofv3 --kernel-trace csv: tools/trace_kernels.py <dir> <name substring> [...]
-> count, median, min us per (kernel name, grid size, workgroup size), in dispatch order of first appearance."""
import collections, csv, glob, statistics, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
subs = sys.argv[2:]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if subs and not any(s in n for s in subs):
        continue
    key = (n.replace("(anonymous namespace)::", "")[:64], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")))
    d.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print(f"{k[0]:64s} grid {k[1]:>8s} wg {k[2]:>5s}  n {len(v):5d}  median {statistics.median(v):8.2f}  min {min(v):8.2f} us")
