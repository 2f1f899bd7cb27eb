#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c10; mkdir -p $O; unset FQHIP_LIB
timeout 600 python -m pytest tests/test_gpu_hadamard.py tests/test_gpu_kron_tall.py tests/test_gpu_had_mfma.py tests/test_gpu_single128.py tests/test_gpu_silu.py tests/test_gpu_kvcache.py -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
