#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c41; mkdir -p $O
CASES="64 80 16384 fq f16 64 80 16384 y f16 56 64 16384 fq f16 56 64 16384 y f16 32 64 16384 fq f16 32 64 16384 y f16 32 64 16384 fqy f16 64 128 16384 y f16"
for v in woff wall default; do
  echo "== $v" >> $O/time.txt
  if [ $v = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=$PWD/variants/libfqhip_$v.so; fi
  timeout 200 python tools/time_kron.py $CASES 2>&1 | grep -v amdgpu.ids >> $O/time.txt
done
cat $O/time.txt
