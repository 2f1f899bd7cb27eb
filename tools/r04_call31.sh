#!/bin/bash
# round 4, GPU call 31: launch plans / static-output modules: parity and host cost; GEMM multi + hadamard tests after the last edits
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c34; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_gemm_bf6.py tests/test_gpu_hadamard.py -x -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
timeout 300 python tools/host_overhead.py 2>&1 | grep -v amdgpu.ids > $O/host.txt; cat $O/host.txt
