#!/usr/bin/env python3
"""Where the transient FP6 route of Linear4bit starts to pay: whole-call time (weights + activations converted for the call + GEMM)
against the int8 path, 129 ... 512 tokens."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flatquant_amd import ops
from tools.bench_gemm import timeit

g = torch.Generator(device="cuda").manual_seed(0)
for N, K in ((4096, 4096), (11008, 4096), (4096, 11008)):
    w = torch.randint(0, 256, (N, K // 2), generator=g, device="cuda", dtype=torch.uint8)
    sw = torch.rand(N, generator=g, device="cuda").half() * 0.01
    for M in (129, 192, 256, 384, 512):
        x = torch.randint(0, 256, (M, K // 2), generator=g, device="cuda", dtype=torch.uint8)
        sx = torch.rand(M, generator=g, device="cuda").half() * 0.01
        t6 = timeit(lambda: ops.bf6_linear(ops.int4_to_bf6(x), sx, ops.int4_to_bf6(w, weights=True), sw, None, M, N, K), steps=50)
        t8 = timeit(lambda: ops.int4_linear(x, sx, w, sw, None), steps=50)
        print(f"M={M:4d} N={N:5d} K={K:5d}: FP6 route, both conversions inside {t6:6.1f} us | int8 path {t8:6.1f} us")
