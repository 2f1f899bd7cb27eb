#!/bin/bash
# round 4, GPU call 1: the low-half quantiser — microbenchmark, parity suite on the new default, A/B against the round-3 quantiser
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c1; mkdir -p $O
timeout 120 tools/scratch/quant3 > $O/quant3.txt 2>&1
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
CASES="64 64 16384 packed f16 128 224 8192 packed f16 112 128 16384 packed f16 64 128 16384 packed f16 172 64 16384 packed f16 128 224 8192 packedr f16"
for rep in 1 2; do
  for lib in default q0 q2; do
    if [ "$lib" = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=variants/libfqhip_$lib.so; fi
    echo "== $lib (sig 0.982: no clamp)"; timeout 300 python tools/time_kron.py $CASES
    echo "== $lib (sig 0.7: clamp)"; SIG=0.7 timeout 300 python tools/time_kron.py 64 64 16384 packed f16 128 224 8192 packed f16 112 128 16384 packed f16
  done
done > $O/ab.txt 2>&1
cat $O/quant3.txt; cat $O/ab.txt
