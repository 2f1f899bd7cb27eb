#!/bin/bash
# round 4, GPU call 53: structured Hadamard kernel A/B — default (s_setprio 2 + three-operand packed extrema), without the extrema form, priority 0,
# the mixlo form of the fp16 quotient; parity on the default build and on the mixlo build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c53; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_had_mfma.py -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
FQHIP_LIB=variants/libfqhip_hmmixlo.so timeout 300 python -m pytest tests/test_gpu_had_mfma.py -q -m gpu > $O/pytest_mixlo.txt 2>&1; tail -3 $O/pytest_mixlo.txt
export TIME_HAD_FAST=1
for rep in 1 2; do
for lib in default hmnomax3 hmprio0 hmmixlo; do
  if [ "$lib" = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=variants/libfqhip_$lib.so; fi
  echo "== $lib"
  timeout 100 python tools/time_had.py 28672:28 14336:28 2>&1 | grep -v amdgpu.ids
done
done > $O/time_had.txt 2>&1
cat $O/time_had.txt
