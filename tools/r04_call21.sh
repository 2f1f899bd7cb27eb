#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c21; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kron_tiles.py tests/test_gpu_kron_trio.py tests/test_gpu_kron_generic.py tests/test_gpu_round2.py -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 300 python tools/time_kron.py 86 128 16384 packed f16 86 128 16384 packedr f16 86 128 16384 packed bf16 108 128 16384 packed f16 96 128 16384 packed f16 2>&1 | grep -v amdgpu.ids > $O/time_86.txt; cat $O/time_86.txt
