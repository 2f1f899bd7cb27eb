#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c7; mkdir -p $O; unset FQHIP_LIB
timeout 600 python -m pytest tests/test_gpu_kron64.py -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
tail -1 $O/bench_default.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value', 'ms_per_step', 'per_rank_ms_per_step', 'broadcast_ms')})
print('roofline', {k: d['roofline'][k] for k in ('frac', 'launch_us')})
print('strong', d.get('strong')); print('c4', d.get('c4')); print('cpu', d.get('cpu_baseline'))
"
tail -3 $O/bench_default.err
