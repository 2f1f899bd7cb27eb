#!/bin/bash
# round 4, GPU call 17: the tall kernel with untracked row loads + an explicit wait in front of the stores: parity, A/B against the tracked loads
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kron_tall.py tests/test_gpu_hadamard.py tests/test_gpu_kron_tiles.py -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
CASES="172 64 16384 h16 f16 172 64 16384 packed f16 172 64 16384 packedr f16 172 64 16384 y f16 140 64 16384 h16 f16 140 64 16384 packedr f16 140 64 16384 y f16 96 64 16384 packed f16 80 64 16384 y f16 80 112 16384 packed f16"
for rep in 1 2; do
for v in default tnoasm; do
  echo "== $v" >> $O/time_tall_ab.txt
  if [ $v = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=$PWD/variants/libfqhip_$v.so; fi
  timeout 200 python tools/time_kron.py $CASES 2>&1 | grep -v amdgpu.ids >> $O/time_tall_ab.txt
done; done
unset FQHIP_LIB
cat $O/time_tall_ab.txt
