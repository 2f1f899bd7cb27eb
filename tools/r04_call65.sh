#!/bin/bash
# round 4, GPU call 65: the plain-Quantizer routes against the oracle's restatement (tests/test_gpu_had_mfma.py, test_gpu_hadamard.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c65; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_had_mfma.py tests/test_gpu_hadamard.py -q -m gpu > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
