#!/bin/bash
# PMC passes over the int4 GEMM (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_gemm
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/gemm_run.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from flatquant_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
M, N, K = 16384, 4096, 4096
x = torch.randint(0, 256, (M, K // 2), generator=g, device="cuda", dtype=torch.uint8)
w = torch.randint(0, 256, (N, K // 2), generator=g, device="cuda", dtype=torch.uint8)
for _ in range(12): ops.int4_matmul(x, w)
torch.cuda.synchronize()
PY
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES" \
  "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" ; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -o p -- python /tmp/gemm_run.py > $OUT/pmc$i.log 2>&1
done
python $R/tools/prof_summary.py $OUT 2>&1 | grep -E "fq_gemm|PMC"
