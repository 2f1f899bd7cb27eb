#!/bin/bash
# round 4, GPU call 44: every kernel the layer bench launches, by total time (looking for launches the mirrors add: copies, casts, fills)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; unset FQHIP_LIB
O=$R/gpurun_out/r04c44; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/tools/bench_layer.py --model llama-2-7b --bsz 1 --steps 50 > $O/layer.txt 2> $O/trace.log
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
python - "$f" > $O/kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:45]:
    print(f"{r['Name'][:110]:110s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} pct={r['Percentage']}")
PY
cat $O/kernels.txt
rm -rf $O/trace
