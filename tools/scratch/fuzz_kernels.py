"""Randomised cross-checks of the round's newer kernels against their simpler siblings (GPU box; prints mismatches)."""
import os, sys, random
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flatquant_amd import ops
from flatquant_amd._lib import FQ_OUT_PACKED, FQ_NO_CLAMP0

random.seed(0)
g = torch.Generator(device="cuda").manual_seed(0)
bad = 0
for it in range(60):   # GEMM: FP6 path and skinny path vs the int8 tile kernel
    M = random.choice([1, 2, 7, 31, 32, 33, 64, 100, 128, 129, 255, 257, 511, 777, 1500])
    N = 16 * random.randint(1, 40)
    K = 128 * random.randint(1, 12)
    x = torch.randint(0, 256, (M, K // 2), generator=g, device="cuda", dtype=torch.uint8)
    w = torch.randint(0, 256, (N, K // 2), generator=g, device="cuda", dtype=torch.uint8)
    sx = (torch.rand(M, generator=g, device="cuda") * 0.05 + 1e-3).half()
    sw = (torch.rand(N, generator=g, device="cuda") * 0.05 + 1e-3).half()
    b = torch.randn(N, generator=g, device="cuda").half() if it % 2 else None
    ref = ops.int4_linear(x, sx, w, sw, b) if M > 128 else None
    c_ref = ops.int4_matmul(x, w)
    y6 = ops.bf6_linear(ops.int4_to_bf6(x), sx, ops.int4_to_bf6(w, weights=True), sw, b, M, N, K)
    c6 = ops.bf6_matmul(ops.int4_to_bf6(x), ops.int4_to_bf6(w, weights=True), M, N, K)
    if not torch.equal(c6, c_ref):
        bad += 1; print("bf6 gemm mismatch", M, N, K)
    if M <= 128:
        img = ops.int4_to_frag(w)
        cs = ops.int4_skinny_matmul(x, img, N)
        ys = ops.int4_skinny_linear(x, sx, img, sw, b, N)
        if not torch.equal(cs, c_ref) or not torch.equal(ys, y6):
            bad += 1; print("skinny mismatch", M, N, K)
    elif not torch.equal(y6, ref):
        bad += 1; print("bf6 linear mismatch", M, N, K)
for it in range(30):   # fused SiLU.mul + Kronecker vs two launches, random rows / clip sets
    M, N = random.choice([(112, 128), (86, 128), (96, 128), (128, 224)])
    rows = random.choice([1, 3, 64, 257, 1000])
    gate = (torch.randn(rows, M * N, generator=g, device="cuda") * 3).half()
    up = torch.randn(rows, M * N, generator=g, device="cuda").half()
    L = (torch.randn(M, M, generator=g, device="cuda") / M ** 0.5).half()
    R = (torch.randn(N, N, generator=g, device="cuda") / N ** 0.5).half()
    sig = [(random.uniform(0.3, 1.0), random.uniform(0.3, 1.0))]
    a = ops.silu_mul_kron_quant(gate, up, L, R, sig, FQ_OUT_PACKED | FQ_NO_CLAMP0)
    c = ops.kron_quant(ops.silu_mul(gate, up), L, R, sig, FQ_OUT_PACKED | FQ_NO_CLAMP0)
    if not (torch.equal(a.q[0], c.q[0]) and torch.equal(a.scale[0], c.scale[0])):
        bad += 1; print("silu kron mismatch", M, N, rows)
for it in range(30):   # fused RMSNorm multi-clip == single-clip launches
    rows = random.choice([1, 5, 16, 100, 4097])
    x = (torch.randn(rows, 4096, generator=g, device="cuda") * random.uniform(0.1, 5)).half()
    L = (torch.randn(64, 64, generator=g, device="cuda") / 8).half()
    R = (torch.randn(64, 64, generator=g, device="cuda") / 8).half()
    sigs = [(random.uniform(0.3, 1.0), random.uniform(0.3, 1.0)) for _ in range(random.randint(1, 4))]
    a = ops.rmsnorm_kron_quant(x, 1e-5, L, R, sigs, FQ_OUT_PACKED)
    for ci, sg in enumerate(sigs):
        b1 = ops.rmsnorm_kron_quant(x, 1e-5, L, R, [sg], FQ_OUT_PACKED)
        if not (torch.equal(a.q[ci], b1.q[0]) and torch.equal(a.scale[ci], b1.scale[0])):
            bad += 1; print("rmsnorm multi-clip mismatch", rows, ci)
print("fuzz done, mismatches:", bad)
