#!/bin/bash
# round 4, GPU call 20: tiles kernel on bf16 (+ N = 128 on bf16): parity, timing
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c20; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kron_tiles.py tests/test_gpu_bf16.py tests/test_gpu_kron_generic.py -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 300 python tools/time_kron.py 112 128 16384 packed bf16 112 128 16384 packedr bf16 86 128 16384 packed bf16 128 144 8192 packed bf16 80 112 16384 packed bf16 144 192 8192 packed bf16 112 128 16384 packed f16 128 144 8192 packed f16 2>&1 | grep -v amdgpu.ids > $O/time_bf16.txt; cat $O/time_bf16.txt
