#!/bin/bash
# Round-2 measurement refresh, GPU box only (via gpurun): shapes table, the reference's kernel benchmark table, the layer table, bench lines of C3 / C4 / C5, PMC of the 112 x 128 launch
# and of the Hadamard + Quantizer launch, phase stamps + shader clock of fq_kron_trio_kernel. Output: gpurun_out/r2g/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2g
mkdir -p $OUT
cd $R
python tools/bench_shapes.py 2>&1 | grep -v amdgpu.ids > $OUT/shapes_table.txt
python tools/bench_kernel_table.py 2>&1 | grep -v amdgpu.ids > $OUT/kernel_benchmark_table.txt
python tools/bench_layer.py 2>&1 | grep -v amdgpu.ids > $OUT/layer_bench.txt
for c in C3 C4 C5; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_$c.json; done
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_C2_driver.json
python tools/show_bench.py $OUT/bench_C2_driver.json $OUT/bench_C3.json $OUT/bench_C4.json $OUT/bench_C5.json > $OUT/configs_bench.txt 2>&1
bash tools/pmc_op.sh kron112 trio112 > /dev/null 2>&1; cp $R/gpurun_out/pmc_trio112/summary.txt $OUT/pmc_kron_trio_112x128.txt
bash tools/pmc_op.sh hadq14336 hadq > /dev/null 2>&1; cp $R/gpurun_out/pmc_hadq/summary.txt $OUT/pmc_hadamard_quant_14336.txt
for v in trace tracecomp tracemem; do
  if [ -f variants/libfqhip_$v.so ]; then echo "== $v"; FQHIP_LIB=$R/variants/libfqhip_$v.so python tools/microbench/trio_trace.py 2>&1 | grep -v amdgpu.ids; fi
done > $OUT/trio_phase_trace.txt
rocm-smi --showmaxpower 2>&1 | grep -i "power" >> $OUT/trio_phase_trace.txt
tail -3 $OUT/configs_bench.txt; grep "112x128\|14336\|18944\|27648\|29568" $OUT/shapes_table.txt | head -20
