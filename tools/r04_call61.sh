#!/bin/bash
# round 4, GPU call 61: refresh on the final build — whole parity suite, bench lines (default with sub-records, C3, C4, C4H, C5), layer benches, shapes table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c61; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_C2_driver.json
for c in C3 C4 C4H C5; do timeout 200 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$c.json; done
python tools/show_bench.py $O/bench_C2_driver.json $O/bench_C3.json $O/bench_C4.json $O/bench_C4H.json $O/bench_C5.json > $O/configs_bench.txt 2>&1; cat $O/configs_bench.txt
timeout 400 python tools/bench_layer.py --model llama-3-8b --bsz 8 > $O/layer_l3_bs8.txt 2>&1; tail -4 $O/layer_l3_bs8.txt | cut -c1-900
timeout 400 python tools/bench_layer.py --model llama-2-7b --bsz 1 > $O/layer_l2_bs1.txt 2>&1; tail -2 $O/layer_l2_bs1.txt | cut -c1-900
timeout 300 python tools/bench_shapes.py 2>&1 | grep -v amdgpu.ids > $O/shapes_table.txt; grep -i "hadamard\|silu" $O/shapes_table.txt
