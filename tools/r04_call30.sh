#!/bin/bash
# round 4, GPU call 30: 112 x 128 on FOUR token groups (tiles kernel, R streamed, 128 VGPRs) against the trio kernel
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c30; mkdir -p $O
FQHIP_LIB=$PWD/variants/libfqhip_t4g.so timeout 600 python -m pytest tests/test_gpu_kron_trio.py -x -q > $O/pytest_t4g.txt 2>&1; tail -3 $O/pytest_t4g.txt
CASES="112 128 16384 packed f16 112 128 16384 packedr f16 108 128 16384 packed f16 100 128 16384 packed f16"
for rep in 1 2; do
for v in default t4g t4g2; do
  echo "== $v" >> $O/time.txt
  if [ $v = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=$PWD/variants/libfqhip_$v.so; fi
  timeout 200 python tools/time_kron.py $CASES 2>&1 | grep -v amdgpu.ids >> $O/time.txt
done; done
cat $O/time.txt
