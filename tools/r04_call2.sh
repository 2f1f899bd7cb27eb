#!/bin/bash
# round 4, GPU call 2: instruction rates of the low-half quantiser's ops, its phase timing, duo with the DMA table; A/B against round 3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c2; mkdir -p $O
timeout 120 tools/scratch/vrate > $O/vrate.txt 2>&1
timeout 120 tools/scratch/quant3 > $O/quant3.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_kron_duo.py -x -q > $O/pytest_duo.txt 2>&1; tail -2 $O/pytest_duo.txt
CASES="128 224 8192 packed f16 128 224 8192 packedr f16 112 128 16384 packed f16 172 64 16384 packed f16 96 64 16384 packed f16"
for rep in 1 2; do
  for lib in default r3 q0; do
    if [ "$lib" = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=variants/libfqhip_$lib.so; fi
    echo "== $lib"; timeout 300 python tools/time_kron.py $CASES
  done
done > $O/ab.txt 2>&1
cat $O/vrate.txt; grep -v MISMATCH $O/quant3.txt; cat $O/ab.txt
