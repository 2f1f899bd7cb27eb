#!/bin/bash
# round 4, GPU call 66: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and instruction counts of the structured Hadamard kernel on the final build:
# 14336 + Quantizer, the same with the SiLU.mul input.
# NOT COMPLETED in round 4: a counter pass of run_op.py takes > 100 s on the box (the per-pass timeout below was 100 s) and the call ran into the
# round's GPU budget; the counters of this kernel before the staged stores are profiles/r04_hadamard_14336_mfma_pmc.txt.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; unset FQHIP_LIB
for OP in hadq14336 hadq14336silu; do
  OUT=$R/gpurun_out/pmc_r04c66_$OP; rm -rf $OUT; mkdir -p $OUT
  CMD="python $R/tools/run_op.py $OP 20"
  i=0
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
  done
  python $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
  rm -rf $OUT/pmc[0-9]* 2>/dev/null; find $OUT -name "*.log" -size +64k -delete 2>/dev/null
  echo "== $OP"; cat $OUT/summary.txt
done
