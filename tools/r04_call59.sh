#!/bin/bash
# round 4, GPU call 59: trio kernel (112 x 128, 86..128 x 128) with the packed output staged through LDS: parity, A/B against the 8-byte stores
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c59; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kron_trio.py tests/test_gpu_hadamard.py tests/test_gpu_group128.py -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for rep in 1 2; do
for lib in default trionostage; do
  if [ "$lib" = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=variants/libfqhip_$lib.so; fi
  echo "== $lib"
  timeout 100 python tools/time_kron.py 112 128 16384 packed f16 112 128 16384 packedr f16 112 128 16384 h16 f16 108 128 16384 packed f16 2>&1 | grep -v amdgpu.ids
done
done > $O/time_trio.txt 2>&1
cat $O/time_trio.txt
