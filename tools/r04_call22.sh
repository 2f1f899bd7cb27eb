#!/bin/bash
# round 4, GPU call 22: PMC summaries of 128x144 (tiles kernel), 128x148 (workgroup-per-token kernel), the structured Hadamard kernel
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
for op in kron128x144 kron128x148 hadq14336; do
  timeout 1500 bash tools/pmc_op.sh $op r04_$op > /dev/null 2>&1
  tail -30 gpurun_out/pmc_r04_$op/summary.txt
done
du -sh gpurun_out
