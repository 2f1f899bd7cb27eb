#!/usr/bin/env python3
"""Where the slow launches of a kernel sit in a rocprofv3 --kernel-trace: tools/trace_outliers.py <trace dir> <kernel substring> [factor]
Prints count / mean / sigma / min / max of the kernel's durations, then every launch slower than `factor` x median (default 1.3) with its
position in the launch sequence, the idle gap in front of it and the phase of the bench command it falls into ("explain the
69 us launches")."""
import csv
import glob
import statistics
import sys

d, key = sys.argv[1], sys.argv[2]
factor = float(sys.argv[3]) if len(sys.argv) > 3 else 1.3
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if key in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
dur = [(e - s) / 1e3 for s, e in rows]
med = statistics.median(dur)
print(f"{key}: n {len(dur)}  mean {statistics.mean(dur):.2f}  sigma {statistics.pstdev(dur):.2f}  median {med:.2f}  min {min(dur):.2f}  max {max(dur):.2f} us")
slow = [i for i, v in enumerate(dur) if v > factor * med]
print(f"launches slower than {factor} x median: {len(slow)} of {len(dur)}")
for i in slow[:60]:
    gap = (rows[i][0] - rows[i - 1][1]) / 1e3 if i else 0.0
    print(f"  launch #{i:5d}  {dur[i]:7.2f} us   idle gap in front {gap:9.1f} us")
# by tenths of the sequence: does the kernel speed up as the part settles?
n = len(dur)
print("mean by tenth of the launch sequence: " + "  ".join(f"{statistics.mean(dur[k * n // 10:(k + 1) * n // 10]):.1f}" for k in range(10)))
