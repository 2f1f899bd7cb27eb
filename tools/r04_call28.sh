#!/bin/bash
# round 4, GPU call 28: 128 x 144 with THREE groups of five waves (R streamed, 128 VGPRs) against two groups
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c28; mkdir -p $O
CASES="128 144 8192 packed f16 128 144 8192 packedr f16 112 144 8192 packed f16"
for rep in 1 2; do
for v in default t3g t3g2; do
  echo "== $v" >> $O/time.txt
  if [ $v = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=$PWD/variants/libfqhip_$v.so; fi
  timeout 200 python tools/time_kron.py $CASES 2>&1 | grep -v amdgpu.ids >> $O/time.txt
done; done
cat $O/time.txt
