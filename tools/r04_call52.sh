#!/bin/bash
# round 4, GPU call 52: the structured Hadamard kernel for n = K * 1024 (28672: template NA = 8, two token groups per CU): parity, timing per route
# (default build and HM_PRIO_MFMA=2), bench C4H (C4 with Hadamard 28 x 1024 + Quantizer as the down_proj input) next to C4 on the same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c52; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_had_mfma.py tests/test_gpu_hadamard.py tests/test_gpu_silu.py -q -m gpu > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
for lib in default hmprio2; do
  if [ "$lib" = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=variants/libfqhip_$lib.so; fi
  echo "== $lib"
  timeout 200 python tools/time_had.py 28672:28 14336:28 2>&1 | grep -v amdgpu.ids
done > $O/time_had.txt 2>&1
unset FQHIP_LIB
cat $O/time_had.txt
timeout 200 python bench.py --config C4H --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_C4H.json
timeout 200 python bench.py --config C4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_C4.json
python tools/show_bench.py $O/bench_C4H.json $O/bench_C4.json 2>&1 | tail -20
