#!/bin/bash
# round 4, GPU call 19: workgroup-per-token kernel with LDS-only barriers + the prefetch wait in front of the stores: parity (all Kronecker tests), A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c19; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kron_generic.py tests/test_gpu_group128.py tests/test_gpu_bf16.py tests/test_gpu_silu.py tests/test_gpu_kron_duo.py tests/test_gpu_kron_tiles.py tests/test_gpu_full_size_next.py tests/test_gpu_hadamard.py tests/test_gpu_general.py -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
CASES="112 128 16384 fq f16 112 128 16384 y f16 112 128 16384 fqy f16 128 224 8192 fq f16 128 224 8192 y f16 168 176 8192 packed f16 128 148 8192 packed f16 64 128 16384 fq f16 64 112 16384 fq f16 112 128 16384 packed bf16 128 224 8192 packed bf16 112 256 8192 y f16"
for rep in 1 2; do
for v in default fsync; do
  echo "== $v" >> $O/time_fast_ab.txt
  if [ $v = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=$PWD/variants/libfqhip_$v.so; fi
  timeout 300 python tools/time_kron.py $CASES 2>&1 | grep -v amdgpu.ids >> $O/time_fast_ab.txt
done; done
unset FQHIP_LIB
cat $O/time_fast_ab.txt
