#!/bin/bash
# round 4, GPU call 5: duo (M = 128 specialisation, uniform scale), trio; parity subset; bench C2 / C3 / C4 lines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kron_duo.py tests/test_gpu_kron_trio.py tests/test_gpu_had_mfma.py tests/test_gpu_hadamard.py tests/test_gpu_full_size_next.py tests/test_gpu_kron_generic.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
CASES="128 224 8192 packed f16 128 224 8192 packedr f16 112 128 16384 packed f16 112 128 16384 packedr f16 120 224 8192 packed f16"
for rep in 1 2; do
  for lib in default r3; do
    if [ "$lib" = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=variants/libfqhip_$lib.so; fi
    echo "== $lib"; timeout 300 python tools/time_kron.py $CASES
    echo "== $lib clamp (sig 0.7)"; SIG=0.7 timeout 300 python tools/time_kron.py 128 224 8192 packed f16 112 128 16384 packed f16
  done
done > $O/ab.txt 2>&1
grep -v amdgpu.ids $O/ab.txt
for cfg in C2 C3 C4; do timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 > $O/bench_$cfg.json 2> $O/bench_$cfg.err; tail -1 $O/bench_$cfg.json | cut -c1-900; done
