#!/bin/bash
# round 4, GPU call 63: Hadamard + Quantizer(lac=False) as one launch (fq_hadamard_quantizer_mfma_f16): parity; timing
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c63; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_had_mfma.py tests/test_gpu_hadamard.py tests/test_gpu_silu.py tests/test_gpu_quant.py tests/test_gpu_kron_tall.py -q -m gpu > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
