#!/usr/bin/env python3
"""Linear4bit FP6-path GEMM against K at 16384 x 4096: the intercept is the per-tile cost (4 tiles of 256 x 256 per CU)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flatquant_amd import ops
from tools.bench_gemm import timeit

M, N = 16384, 4096
g = torch.Generator(device="cuda").manual_seed(0)
for K in (128, 256, 512, 1024, 2048, 4096, 8192):
    x = torch.randint(0, 256, (M, K // 2), generator=g, device="cuda", dtype=torch.uint8)
    w = torch.randint(0, 256, (N, K // 2), generator=g, device="cuda", dtype=torch.uint8)
    sx = torch.rand(M, generator=g, device="cuda").half() * 0.01
    sw = torch.rand(N, generator=g, device="cuda").half() * 0.01
    wb, xb = ops.int4_to_bf6(w, weights=True), ops.int4_to_bf6(x)
    t6 = timeit(lambda: ops.bf6_linear(xb, sx, wb, sw, None, M, N, K), steps=50)
    tc = timeit(lambda: ops.bf6_matmul(xb, wb, M, N, K), steps=50)
    print(f"K={K:5d}: linear {t6:7.1f} us ({t6 / 4:6.2f} per tile round, {K // 128} stages) | int32 out {tc:7.1f} us")
