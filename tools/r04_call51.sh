#!/bin/bash
# round 4, GPU call 51: one conversion launch for the distinct activations of a multi-problem GEMM call: parity, timing, layer benches (plain and --graph)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c51; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_gemm_bf6.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 300 python tools/time_gate_up.py 2048 11008 4096 2048 4096 4096 16384 14336 4096 2>&1 | grep -v amdgpu.ids > $O/time.txt; cat $O/time.txt
timeout 400 python tools/bench_layer.py --model llama-2-7b --bsz 1 > $O/layer_l2_bs1.txt 2>&1; tail -17 $O/layer_l2_bs1.txt | cut -c1-700
timeout 400 python tools/bench_layer.py --model llama-2-7b --bsz 1 --graph > $O/layer_l2_bs1_graph.txt 2>&1; tail -17 $O/layer_l2_bs1_graph.txt | cut -c1-700
timeout 400 python tools/bench_layer.py --model llama-3-8b --bsz 8 > $O/layer_l3_bs8.txt 2>&1; tail -3 $O/layer_l3_bs8.txt | cut -c1-700
