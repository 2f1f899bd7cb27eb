#!/bin/bash
# round 4, GPU call 27: 128 x 148 on the tiles kernel (N % 16 != 0): parity, timing
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c27; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kron_tiles.py tests/test_gpu_kron_generic.py tests/test_gpu_round2.py -x -q > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
timeout 300 python tools/time_kron.py 128 148 8192 packed f16 128 148 8192 packedr f16 128 148 8192 packed bf16 128 144 8192 packed f16 2>&1 | grep -v amdgpu.ids > $O/time.txt; cat $O/time.txt
