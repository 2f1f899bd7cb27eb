#!/bin/bash
# Round-3 measurement refresh, GPU box only (via gpurun). Output: gpurun_out/r3g/  (copied into profiles/r03_* afterwards)
#   bench lines (C1, C2 driver command, C2 on bf16, C3, C4, C5) + per-kernel tables, the C-ABI timing table of the Kronecker
#   shapes (packed / fake-quant / transform, fp16 / bf16), the per-shape table, measured flip rates against the reference's
#   goldens, PMC of the 128 x 224 launch (fq_kron_duo_kernel) and of the C1 fake-quant launch, phase stamps of the duo kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r3g
mkdir -p $OUT
cd $R
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_C2_driver.json
python bench.py --config C1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_C1.json
python bench.py --dtype bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_C2_bf16.json
for c in C3 C4 C5; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_$c.json; done
python tools/show_bench.py $OUT/bench_C2_driver.json $OUT/bench_C1.json $OUT/bench_C2_bf16.json $OUT/bench_C3.json $OUT/bench_C4.json $OUT/bench_C5.json > $OUT/configs_bench.txt 2>&1
python tools/time_kron.py 2>&1 | grep -v amdgpu.ids > $OUT/kron_contracts_table.txt
python tools/flip_rates.py 2>&1 | grep -v amdgpu.ids > $OUT/flip_rates.txt
python tools/bench_shapes.py 2>&1 | grep -v amdgpu.ids > $OUT/shapes_table.txt
bash tools/pmc_op.sh kron128x224 duo224 > /dev/null 2>&1; cp $R/gpurun_out/pmc_duo224/summary.txt $OUT/pmc_kron_duo_128x224.txt
bash tools/pmc_op.sh kron64fq fq64 > /dev/null 2>&1; cp $R/gpurun_out/pmc_fq64/summary.txt $OUT/pmc_kron64_fakequant.txt
if [ -f variants/libfqhip_dtrace.so ]; then FQHIP_LIB=$R/variants/libfqhip_dtrace.so python tools/microbench/duo_trace.py 2>&1 | grep -v amdgpu.ids > $OUT/duo_phase_trace.txt; fi
tail -8 $OUT/configs_bench.txt; tail -12 $OUT/flip_rates.txt
