#!/bin/bash
# round 4, GPU call 36: the N > 1 code path on one GPU: bench.py inside a ONE-rank RCCL group (plain and under torch.distributed.run); C5 on bf16
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c36; mkdir -p $O
timeout 300 python bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_force_dist.json 2> $O/bench_force_dist.err; tail -c 600 $O/bench_force_dist.json; tail -3 $O/bench_force_dist.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; tail -c 400 $O/bench_torchrun1.json; tail -3 $O/bench_torchrun1.err
timeout 300 python bench.py --config C5 --dtype bf16 --no-cpu-baseline > $O/bench_C5_bf16.json 2> $O/c5.err; python tools/show_bench.py $O/bench_C5_bf16.json; tail -2 $O/c5.err
timeout 300 python bench.py --config C5 --force-dist --no-cpu-baseline > $O/bench_C5_force.json 2>/dev/null; python tools/show_bench.py $O/bench_C5_force.json
