#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c9; mkdir -p $O; unset FQHIP_LIB
timeout 600 python -m pytest tests/test_gpu_hadamard.py tests/test_gpu_kron_tall.py tests/test_gpu_had_mfma.py -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 300 python - > $O/time_had.txt 2>&1 <<'PY'
import torch, statistics, sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from flatquant_amd import ops
from flatquant_amd._lib import FQ_OUT_TRANSFORM, FQ_ROUND_Y_F16
from tests.conftest import hadk_matrix
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(f, steps=50, rounds=5):
    for i in range(10): f(i)
    torch.cuda.synchronize(); ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps): f(i)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / steps * 1e3)
    return statistics.median(ts), min(ts)
for n, K, r in ((11008, 172, 16384), (8960, 140, 16384), (5120, 40, 16384), (28672, 28, 8192), (14336, 28, 16384)):
    xs = [torch.randn(r, n, generator=g, device="cuda", dtype=torch.float32).half() for _ in range(2)]
    hk = torch.from_numpy(hadk_matrix(K)).cuda()
    cases = [("hadamard default", lambda i: ops.hadamard(xs[i % 2], K, hk)), ("hadamard fwht", lambda i: ops.hadamard(xs[i % 2], K, hk, fwht_route=True))]
    kr = ops._hadamard_as_kron(K, n // K, hk, xs[0].device)
    if kr is not None:
        sc = ops._had_right_div(kr[2]) / n ** 0.5
        cases.append((f"kron transform-only {kr[0].shape[0]}x{kr[2]}", lambda i: ops.kron_quant_ex(xs[i % 2], kr[0], kr[1], sc, [(1.0, 1.0)], FQ_OUT_TRANSFORM | FQ_ROUND_Y_F16)))
    for name, f in cases:
        us, mn = timeit(f)
        b = r * 4.0 * n
        print(f"n={n:5d} rows={r:5d} {name:34s} {us:8.1f} us (min {mn:.1f})  {b / us / 1e3:7.0f} GB/s  {b / us / 8e6:5.3f} of 8 TB/s", flush=True)
    del xs
PY
grep -v amdgpu.ids $O/time_had.txt
