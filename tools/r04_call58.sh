#!/bin/bash
# round 4, GPU call 58: structured Hadamard kernel, next token through registers (a whole iteration in flight) instead of LDS-DMA in phase B: parity, A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c58; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_had_mfma.py tests/test_gpu_hadamard.py tests/test_gpu_silu.py -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
export TIME_HAD_FAST=1
for rep in 1 2; do
for lib in default hmnopf; do
  if [ "$lib" = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=variants/libfqhip_$lib.so; fi
  echo "== $lib"
  timeout 100 python tools/time_had.py 14336:28 6144:12 2>&1 | grep -v amdgpu.ids
done
done > $O/time_had.txt 2>&1
cat $O/time_had.txt
