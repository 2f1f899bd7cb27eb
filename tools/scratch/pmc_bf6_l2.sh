#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_bf6_l2
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
for _ in range(12): ops.int4_matmul(x, w)
torch.cuda.synchronize()
PY
i=0
for PMC in "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCC_READ_sum" ; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -o p -- python /tmp/bf6_run.py > $OUT/pmc$i.log 2>&1
done
python $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
grep -E "fq_gemm" $OUT/summary.txt | head -30
