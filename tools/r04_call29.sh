#!/bin/bash
# round 4, GPU call 29: multi-problem GEMM launch: parity, the single-problem kernel unchanged (bench_gemm), layer benches; the standalone Hadamard 14336 route
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c29; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_gemm_bf6.py tests/test_gpu_gemm_i4.py -x -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
timeout 300 python tools/bench_gemm.py > $O/bench_gemm.txt 2>&1; tail -12 $O/bench_gemm.txt
timeout 400 python tools/bench_layer.py --model llama-2-7b --bsz 1 > $O/layer_l2_bs1.txt 2>&1; tail -9 $O/layer_l2_bs1.txt
timeout 400 python tools/bench_layer.py --model llama-3-8b --bsz 8 > $O/layer_l3_bs8.txt 2>&1; tail -9 $O/layer_l3_bs8.txt
timeout 200 python - > $O/had14336.txt 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
from flatquant_amd import ops
from flatquant_amd.flatquant.hadamard_utils import get_hadK
hk, K = get_hadK(14336); hk = hk.half().cuda()
xs = [torch.randn(16384, 14336, device="cuda", dtype=torch.float16) for _ in range(4)]
def timeit(fn, steps=30, warm=5):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3
print("hadamard default", timeit(lambda i: ops.hadamard(xs[i % 4], K, hk)))
print("hadamard fwht   ", timeit(lambda i: ops.hadamard(xs[i % 4], K, hk, fwht_route=True)))
print("hadamard mfma   ", timeit(lambda i: ops.hadamard_mfma(xs[i % 4], K, hk, None, None, True)))
kr = ops._hadamard_as_kron(K, 512, hk, xs[0].device)
print("kron pair", kr[0].shape, kr[1].shape, kr[2])
PY
grep -v amdgpu.ids $O/had14336.txt
