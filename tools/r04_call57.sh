#!/bin/bash
# round 4, GPU call 57: SiLU.mul input on the structured Hadamard kernel (fq_silu_mul_hadamard_quant_mfma_f16): parity, timing against the dense launch
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c57; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_had_mfma.py tests/test_gpu_hadamard.py tests/test_gpu_silu.py -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
TIME_HAD_SILU=1 timeout 200 python tools/time_had.py 14336:28 28672:28 2>&1 | grep -v amdgpu.ids > $O/time_silu.txt; cat $O/time_silu.txt
