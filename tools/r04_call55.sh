#!/bin/bash
# round 4, GPU call 55: structured Hadamard kernel with the packed output staged through LDS (1 KB-contiguous stores): parity, A/B against the 8-byte stores
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c55; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_had_mfma.py tests/test_gpu_hadamard.py -q -m gpu -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
export TIME_HAD_FAST=1
for rep in 1 2; do
for lib in default hmnostage; do
  if [ "$lib" = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=variants/libfqhip_$lib.so; fi
  echo "== $lib"
  timeout 100 python tools/time_had.py 28672:28 14336:28 2>&1 | grep -v amdgpu.ids
done
done > $O/time_had.txt 2>&1
cat $O/time_had.txt
