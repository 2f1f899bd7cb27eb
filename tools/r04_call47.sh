#!/bin/bash
# round 4, GPU call 47: SiLU.mul in the epilogue of the gate / up GEMM launch (fq_int4_linear_fp6_gate_up_f16): parity, layer benches
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c47; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_gemm_bf6.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
timeout 400 python tools/bench_layer.py --model llama-3-8b --bsz 8 > $O/layer_l3_bs8.txt 2>&1; tail -14 $O/layer_l3_bs8.txt | cut -c1-400
timeout 400 python tools/bench_layer.py --model llama-2-7b --bsz 1 > $O/layer_l2_bs1.txt 2>&1; tail -14 $O/layer_l2_bs1.txt | cut -c1-400
