#!/bin/bash
# round 4, GPU call 3: the structured Hadamard kernel — parity, then timing against the dense Kronecker launch and the FWHT route
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_had_mfma.py -x -q > $O/pytest_had.txt 2>&1; tail -15 $O/pytest_had.txt
timeout 300 python - > $O/time_had.txt 2>&1 <<'PY'
import torch, statistics, sys
sys.path.insert(0, '.')
from flatquant_amd import ops
from tests.conftest import hadk_matrix
rows, n, K = 16384, 14336, 28
g = torch.Generator(device="cuda").manual_seed(0)
xs = [torch.randn(rows, n, generator=g, device="cuda", dtype=torch.float32).half() for _ in range(2)]
hk = torch.from_numpy(hadk_matrix(K)).cuda()
sig = (0.9820137619972229, 0.9820137619972229)
def timeit(f, steps=50, rounds=5):
    for i in range(10): f(i)
    torch.cuda.synchronize(); ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps): f(i)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / steps * 1e3)
    return statistics.median(ts), min(ts)
pb = rows * (2.5 * n + 2)
for name, f, b in [("hadamard_quant mfma (structured)", lambda i: ops.hadamard_quant(xs[i % 2], K, hk, sig), pb),
                   ("hadamard_quant kron (dense 112x128)", lambda i: ops.hadamard_quant(xs[i % 2], K, hk, sig, route="kron"), pb),
                   ("hadamard_quant fwht", lambda i: ops.hadamard_quant(xs[i % 2], K, hk, sig, route="fwht"), pb),
                   ("hadamard mfma (fp16 out)", lambda i: ops.hadamard(xs[i % 2], K, hk), rows * 4.0 * n),
                   ("hadamard fwht (fp16 out)", lambda i: ops.hadamard(xs[i % 2], K, hk, fwht_route=True), rows * 4.0 * n)]:
    us, mn = timeit(f)
    print(f"{name:40s} {us:8.1f} us (min {mn:.1f})  {b / us / 1e3:7.0f} GB/s  {b / us / 8e6:5.3f} of 8 TB/s", flush=True)
PY
cat $O/time_had.txt
