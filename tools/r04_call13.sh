#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c13; mkdir -p $O
timeout 200 python tools/host_overhead.py > $O/host.txt 2>&1; grep -v amdgpu.ids $O/host.txt
timeout 400 python tools/bench_layer.py --model llama-3-8b --bsz 8 > $O/layer_l3_bs8.txt 2>&1; tail -25 $O/layer_l3_bs8.txt
timeout 400 python tools/bench_layer.py --model llama-2-7b --bsz 8 > $O/layer_l2_bs8.txt 2>&1; tail -12 $O/layer_l2_bs8.txt
timeout 400 python tools/bench_layer.py --model llama-2-7b --bsz 1 > $O/layer_l2_bs1.txt 2>&1; tail -12 $O/layer_l2_bs1.txt
timeout 400 python tools/bench_layer.py --model llama-2-7b --bsz 1 --graph > $O/layer_l2_bs1_graph.txt 2>&1; tail -6 $O/layer_l2_bs1_graph.txt
