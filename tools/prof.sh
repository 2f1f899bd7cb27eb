#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + PMC passes for bench.py. Usage: tools/prof.sh <tag>   (PROF_EXTRA=1: the two cache-counter passes as well)
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# kernel trace: the default bench command (1000 steps after 200 warm-up launches), minus the CPU baseline leg
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --no-cpu-baseline --no-sub-records > $OUT/trace.log 2>&1
CMD="python $R/bench.py --steps 200 --warmup 100 --no-cpu-baseline --no-sub-records"   # counter passes: fewer launches
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAVES" \
  ${PROF_EXTRA:+"TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"} \
  ${PROF_EXTRA:+"TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"} ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
done
python $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
python $R/tools/trace_outliers.py $OUT/trace fq_kron64_kernel > $OUT/outliers.txt 2>&1   # (round 6: where the slow launches of the trace sit)
# prune the raw rocprofv3 output (gpurun copies back at most 64 MiB): keep the summary and the kernel-stats csv
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \; 2>/dev/null
rm -rf $OUT/trace $OUT/pmc[0-9]* 2>/dev/null; find $OUT -name "*.log" -size +64k -delete 2>/dev/null
cat $OUT/summary.txt
