#!/bin/bash
# round 4, GPU call 45: s_setprio for the GEMM phases of the duo / trio kernels (waves in MFMA phases ahead of the quantising ones)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c45; mkdir -p $O
CASES="128 224 8192 packed f16 112 128 16384 packed f16"
for rep in 1 2; do
for v in default prio2 prio3; do
  echo "== $v" >> $O/time.txt
  if [ $v = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=$PWD/variants/libfqhip_$v.so; fi
  timeout 200 python tools/time_kron.py $CASES 2>&1 | grep -v amdgpu.ids >> $O/time.txt
done; done
cat $O/time.txt
