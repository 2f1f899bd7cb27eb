#!/bin/bash
# round 4, GPU call 56: the whole parity suite on the build with the K * 1024 Hadamard kernel and the staged packed stores; bench C3 / C4H / C4
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c56; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 200 python bench.py --config C3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_C3.json
timeout 200 python bench.py --config C4H --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_C4H.json
timeout 200 python bench.py --config C4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_C4.json
python tools/show_bench.py $O/bench_C3.json $O/bench_C4H.json $O/bench_C4.json 2>&1
