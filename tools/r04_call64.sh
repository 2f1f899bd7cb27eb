#!/bin/bash
# round 4, GPU call 64: final build — the whole parity suite; rocprofv3 kernel trace + stats of the C3 / C4H bench commands and of the default command
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; unset FQHIP_LIB
O=$R/gpurun_out/r04c64; mkdir -p $O
cd $R; timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt; cd /tmp
for c in C3 C4H; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$c -o t -- python $R/bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/trace_$c.log
  f=$(find $O/trace_$c -name "*kernel_stats.csv" | head -1)
  echo "== bench.py --config $c under rocprofv3 --kernel-trace --stats" >> $O/kernel_stats.txt
  python - "$f" >> $O/kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if float(r.get("Percentage", 0) or 0) < 0.5: continue
    n = r["Name"]
    for key in ("fq_kron64", "fq_kron_wave", "fq_kron_duo", "fq_kron_trio", "fq_kron_tiles", "fq_block", "fq_had512", "fq_kron_fast", "fq_rowquant"):
        if key in n:
            i = n.find(key); n = n[i:i + 60]; break
    print(f"  {n[:60]:60s} calls={r['Calls']:>6s} avg_ns={float(r['AverageNs']):10.0f} min_ns={float(r['MinNs']):10.0f} max_ns={float(r['MaxNs']):10.0f} pct={r['Percentage']}")
PY
  python $R/tools/show_bench.py $O/bench_$c.json >> $O/kernel_stats.txt 2>&1
  rm -rf $O/trace_$c
done
cat $O/kernel_stats.txt | cut -c1-200
