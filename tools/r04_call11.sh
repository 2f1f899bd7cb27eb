#!/bin/bash
# round 4, GPU call 11: profiles — bench kernel trace + PMC (kron64), PMC of the duo / structured Hadamard / trio launches, duo phase stamps, clocks
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
timeout 500 bash tools/prof.sh r04 > gpurun_out/prof_r04.log 2>&1; tail -30 gpurun_out/prof_r04/summary.txt
timeout 300 bash tools/pmc_op.sh kron128x224 duo224_r04 > /dev/null 2>&1; cat gpurun_out/pmc_duo224_r04/summary.txt | tail -50
timeout 300 bash tools/pmc_op.sh hadq14336 hadq14336_r04 > /dev/null 2>&1; cat gpurun_out/pmc_hadq14336_r04/summary.txt | tail -45
timeout 300 bash tools/pmc_op.sh kron112 trio112_r04 > /dev/null 2>&1; cat gpurun_out/pmc_trio112_r04/summary.txt | tail -45
mkdir -p gpurun_out/r04c11
FQHIP_LIB=variants/libfqhip_dtrace.so timeout 120 python tools/scratch/duo_trace.py > gpurun_out/r04c11/duo_trace.txt 2>&1; tail -45 gpurun_out/r04c11/duo_trace.txt
for op in kron64 kron128x224 kron112 hadq14336 hadq11008; do timeout 60 bash tools/scratch/clock_under_load.sh $op 5; done > gpurun_out/r04c11/clocks.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04c11/clocks.txt | grep "launches\|Mhz\|Power" | awk '{print}' | head -70
