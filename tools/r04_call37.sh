#!/bin/bash
# round 4, GPU call 37: stdout of a dist run = exactly one JSON line (RCCL's banner goes to stderr)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c38; mkdir -p $O
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err
wc -l $O/bench_torchrun1.json; python -c "
import json; d=json.loads(open('$O/bench_torchrun1.json').read()); print(d['dist'], round(d['ms_per_step']*1e3,2), 'us/step', round(d['broadcast_ms'],3), 'ms broadcast', d['roofline']['frac'], d['strong']['ms_per_step'], d['c4']['ms_per_step'])"
grep -c "RCCL version" $O/bench_torchrun1.err
timeout 300 python bench.py --force-dist --config C5 --no-cpu-baseline > $O/bench_C5_force.json 2>/dev/null; wc -l $O/bench_C5_force.json; python tools/show_bench.py $O/bench_C5_force.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records | wc -l
