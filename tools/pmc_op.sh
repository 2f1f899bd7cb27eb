#!/bin/bash
# rocprofv3 counter passes over one op (tools/run_op.py), GPU box only: tools/pmc_op.sh <op> <tag>
# Counters in their own runs with --kernel-trace only (never with sys/runtime/hip traces). Summary -> gpurun_out/pmc_<tag>/summary.txt
OP=$1; TAG=${2:-$1}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/run_op.py $OP 30"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES" \
  "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVES" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
done
python $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
# prune the raw rocprofv3 output (gpurun copies back at most 64 MiB): keep the summary and the kernel-stats csv
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \; 2>/dev/null
rm -rf $OUT/trace $OUT/pmc[0-9]* 2>/dev/null; find $OUT -name "*.log" -size +64k -delete 2>/dev/null
cat $OUT/summary.txt
