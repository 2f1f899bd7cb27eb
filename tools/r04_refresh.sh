#!/bin/bash
# Round-4 measurement refresh, GPU box only (via gpurun). Output: gpurun_out/r4g/  (copied into profiles/r04_* afterwards)
#   bench lines (C1, C2 driver command, C2 on bf16, C3, C4, C5) + per-kernel tables, the C-ABI timing table of the Kronecker
#   shapes (packed / fake-quant / transform, fp16 / bf16), the per-shape table, measured flip rates against the reference's
#   goldens; the parity suite first. (PMC / phase stamps: tools/r04_call11.sh.)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4g
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_C2_driver.json
python bench.py --config C1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_C1.json
python bench.py --dtype bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_C2_bf16.json
python bench.py --config C5 --dtype bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_C5_bf16.json
for c in C3 C4 C5; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_$c.json; done
python tools/show_bench.py $OUT/bench_C2_driver.json $OUT/bench_C1.json $OUT/bench_C2_bf16.json $OUT/bench_C3.json $OUT/bench_C4.json $OUT/bench_C5.json $OUT/bench_C5_bf16.json > $OUT/configs_bench.txt 2>&1
python tools/time_kron.py 2>&1 | grep -v amdgpu.ids > $OUT/kron_contracts_table.txt
python tools/flip_rates.py 2>&1 | grep -v amdgpu.ids > $OUT/flip_rates.txt
python tools/bench_shapes.py 2>&1 | grep -v amdgpu.ids > $OUT/shapes_table.txt
tail -8 $OUT/configs_bench.txt; tail -12 $OUT/flip_rates.txt
