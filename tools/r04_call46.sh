#!/bin/bash
# round 4, GPU call 46: s_setprio in the GEMM phases as the default of the duo / trio / tiles kernels: parity, then A/B of the tiles kernel
# (variant built with -DFQ_PRIO_MFMA=0) and of the structured Hadamard kernel (variant built with -DHM_PRIO_MFMA=2)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c46; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kron_duo.py tests/test_gpu_kron_tiles.py tests/test_gpu_kron_generic.py tests/test_gpu_hadamard.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
CASES="128 224 8192 packed f16 112 128 16384 packed f16 128 144 16384 packed f16 80 112 16384 packed f16 86 128 16384 packed f16 144 192 8192 packed f16 168 176 8192 packed f16 128 148 8192 packed f16"
for rep in 1 2; do
for v in default tiles0; do
  echo "== $v" >> $O/time.txt
  if [ $v = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=$PWD/variants/libfqhip_$v.so; fi
  timeout 300 python tools/time_kron.py $CASES 2>&1 | grep -v amdgpu.ids >> $O/time.txt
done; done
for rep in 1 2; do
for v in default hm2; do
  echo "== $v" >> $O/time.txt
  if [ $v = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=$PWD/variants/libfqhip_$v.so; fi
  TIME_OP=1 timeout 200 python tools/run_op.py hadq14336 2>&1 | grep -v amdgpu.ids | tail -2 >> $O/time.txt
done; done
cat $O/time.txt
