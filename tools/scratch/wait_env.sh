#!/bin/bash
# A/B: the K = 20 driver command under different host wait modes of the HIP runtime (wall ms_per_step, launch_us)
mkdir -p gpurun_out/s2; out=gpurun_out/s2/wait_env.txt
: > $out
for rep in 1 2 3; do
for env in "X=1" "ROC_ACTIVE_WAIT_TIMEOUT=100000" "ROC_ACTIVE_WAIT_TIMEOUT=100000 ROC_CPU_WAIT_FOR_SIGNAL=0" "ROC_CPU_WAIT_FOR_SIGNAL=0"; do
  r=$(env $env python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), round(d['roofline']['launch_us'],2))")
  echo "$env : $r" >> $out
done
done
cat $out
