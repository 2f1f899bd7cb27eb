#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output written by tools/prof.sh: per-kernel average duration and PMC values."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    for key in ("fq_had512", "fq_rowmm", "fq_kron_tiles", "fq_kron_general", "fq_fakequant", "fq_silu", "fq_rmsnorm", "fq_gemm_bf6", "fq_kron_tall", "fq_kv_decode", "fq_gemm_i4_skinny", "fq_kron64", "fq_kron_duo", "fq_kron_trio", "fq_kron_fast", "fq_kron_wave", "fq_gemm_i4", "fq_rowquant", "fq_probe_stream", "fq_kron_generic", "fq_hadamard", "fq_block", "copyBuffer"):
        if key in name:
            i = name.find(key)
            return name[i:i + 46]
    return name[:40]


for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, root))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if float(row.get("Percentage", 0) or 0) < 0.5:
                continue
            print(f"  {short(row['Name']):42s} calls={row['Calls']:>5s} avg_ns={float(row['AverageNs']):10.0f} "
                  f"min_ns={float(row['MinNs']):10.0f} max_ns={float(row['MaxNs']):10.0f} pct={row['Percentage']}")

for d in sorted(glob.glob(os.path.join(root, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("== PMC", os.path.basename(d))
        for k, cs in acc.items():
            if not k.startswith("fq_"):
                continue
            for c, vals in cs.items():
                vals = vals[len(vals) // 4:]  # drop warm-up dispatches
                print(f"  {k:42s} {c:34s} avg={sum(vals) / len(vals):16.1f}  n={len(vals)}")
