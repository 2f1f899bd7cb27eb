#!/bin/bash
# PMC passes over the FP6-path GEMM and the INT4 decode attention (GPU box); summary -> gpurun_out/pmc_bf6/summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_bf6
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/bf6_run.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from flatquant_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
M, N, K = 16384, 4096, 4096
x = torch.randint(0, 256, (M, K // 2), generator=g, device="cuda", dtype=torch.uint8)
w = torch.randint(0, 256, (N, K // 2), generator=g, device="cuda", dtype=torch.uint8)
xb, wb = ops.int4_to_bf6(x), ops.int4_to_bf6(w, weights=True)
for _ in range(12): ops.bf6_matmul(xb, wb, M, N, K)
bsz, seq, heads, hd, page = 64, 2048, 32, 128, 2048
data = torch.randint(0, 256, (bsz, 1, 2, heads, page, hd // 2), generator=g, device="cuda", dtype=torch.uint8)
par = (torch.rand(bsz, 1, 2, heads, page, 2, generator=g, device="cuda") * 0.2 + 0.05).half()
indptr = torch.arange(bsz + 1, device="cuda", dtype=torch.int32)
indices = torch.arange(bsz, device="cuda", dtype=torch.int32)
last = torch.full((bsz,), seq, device="cuda", dtype=torch.int32)
q = torch.randn(bsz, heads, hd, generator=g, device="cuda").half()
for _ in range(12): ops.kv_batch_decode(q, data, par, indptr, indices, last, 0)
torch.cuda.synchronize()
PY
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES" ; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -o p -- python /tmp/bf6_run.py > $OUT/pmc$i.log 2>&1
done
python $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
grep -E "PMC|fq_gemm_bf6|fq_kv_decode" -A14 $OUT/summary.txt | head -80
