#!/bin/bash
# round 4, GPU call 16: tiles kernel variants (groups / fragments in flight / streamed R) A/B in one call; the tall kernel's routes through the C ABI
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c16; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kron_tiles.py -x -q > $O/pytest_tiles.txt 2>&1; tail -3 $O/pytest_tiles.txt
CASES="80 112 16384 packed f16 80 112 16384 packedr f16 128 144 8192 packed f16 128 144 8192 packedr f16 144 192 8192 packed f16 144 192 8192 packedr f16"
for rep in 1 2; do
for v in default tg4 tda4 trs; do
  echo "== $v" >> $O/time_tiles_variants.txt
  if [ $v = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=$PWD/variants/libfqhip_$v.so; fi
  timeout 200 python tools/time_kron.py $CASES 2>&1 | grep -v amdgpu.ids >> $O/time_tiles_variants.txt
done; done
unset FQHIP_LIB
cat $O/time_tiles_variants.txt
timeout 200 python tools/time_kron.py 172 64 16384 h16 f16 172 64 16384 packed f16 172 64 16384 packedr f16 172 64 16384 y f16 140 64 16384 h16 f16 140 64 16384 packedr f16 2>&1 | grep -v amdgpu.ids > $O/time_tall.txt; cat $O/time_tall.txt
