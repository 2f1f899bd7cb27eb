#!/bin/bash
# round 4, GPU call 62: layer bench again (median of three timed windows per piece)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c62; mkdir -p $O
timeout 400 python tools/bench_layer.py --model llama-3-8b --bsz 8 > $O/layer_l3_bs8.txt 2>&1; grep "one launch\|seven linears\|fp16 layer" $O/layer_l3_bs8.txt | cut -c1-900
timeout 400 python tools/bench_layer.py --model llama-2-7b --bsz 1 > $O/layer_l2_bs1.txt 2>&1; tail -1 $O/layer_l2_bs1.txt | cut -c1-900
