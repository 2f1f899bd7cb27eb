#!/usr/bin/env python3
"""Linear4bit FP6-path GEMM against the token count (prefill of 1 ... 8 sequences of 2048): tools/scratch/gemm_m_sweep.py [N K]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flatquant_amd import ops
from tools.bench_gemm import timeit

N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4096)
g = torch.Generator(device="cuda").manual_seed(0)
for M in (512, 1024, 2048, 4096, 8192, 16384):
    x = torch.randint(0, 256, (M, K // 2), generator=g, device="cuda", dtype=torch.uint8)
    w = torch.randint(0, 256, (N, K // 2), generator=g, device="cuda", dtype=torch.uint8)
    sx = torch.rand(M, generator=g, device="cuda").half() * 0.01
    sw = torch.rand(N, generator=g, device="cuda").half() * 0.01
    wb, xb = ops.int4_to_bf6(w, weights=True), ops.int4_to_bf6(x)
    t6 = timeit(lambda: ops.bf6_linear(xb, sx, wb, sw, None, M, N, K), steps=50)
    t8 = timeit(lambda: ops.int4_linear(x, sx, w, sw, None), steps=50)
    ok = torch.equal(ops.bf6_linear(xb, sx, wb, sw, None, M, N, K), ops.int4_linear(x, sx, w, sw, None))
    print(f"M={M:6d} N={N} K={K}: FP6 path {t6:7.1f} us ({2.0 * M * N * K / t6 / 1e9:5.2f} Pop/s) | int8 path {t8:7.1f} us  identical={ok}")
