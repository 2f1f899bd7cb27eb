#!/bin/bash
# round 4, GPU call 18: tall kernel (unconditional untracked loads) parity; duo kernel with deferred stores: parity + A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kron_tall.py tests/test_gpu_hadamard.py tests/test_gpu_kron_duo.py tests/test_gpu_kron_tiles.py -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
CASES="128 224 8192 packed f16 128 224 8192 packedr f16 120 224 8192 packed f16 172 64 16384 h16 f16 172 64 16384 packed f16 140 64 16384 h16 f16"
for rep in 1 2 3; do
for v in default dnodefer; do
  echo "== $v" >> $O/time_duo_ab.txt
  if [ $v = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=$PWD/variants/libfqhip_$v.so; fi
  timeout 200 python tools/time_kron.py $CASES 2>&1 | grep -v amdgpu.ids >> $O/time_duo_ab.txt
done; done
unset FQHIP_LIB
cat $O/time_duo_ab.txt
