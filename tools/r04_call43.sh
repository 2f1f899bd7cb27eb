#!/bin/bash
# round 4, GPU call 43: prepared images shared across streams / graph capture: parity; C4 / C5 / C3 steps again
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c43; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_kron_generic.py tests/test_gpu_gemm_bf6.py -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for c in C5 C4 C3; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$c.json; done
python tools/show_bench.py $O/bench_C5.json $O/bench_C4.json $O/bench_C3.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C2', d['ms_per_step']*1e3, d['roofline']['frac'], 'strong', d['strong']['ms_per_step'], 'c4', d['c4']['ms_per_step'], d['c4']['step_hbm_frac_rank0'])"
