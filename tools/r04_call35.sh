#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c35; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kron_duo.py tests/test_gpu_bf16.py -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 200 python tools/time_kron.py 128 224 8192 packed bf16 128 224 8192 packedr bf16 128 224 8192 packed f16 2>&1 | grep -v amdgpu.ids > $O/time.txt; cat $O/time.txt
