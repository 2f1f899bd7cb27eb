#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c6; mkdir -p $O; unset FQHIP_LIB
for cfg in C2 C3 C4 C5; do timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 > $O/bench_$cfg.json 2> $O/bench_$cfg.err; tail -1 $O/bench_$cfg.json | cut -c1-1200; done
