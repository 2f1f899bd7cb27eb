#!/usr/bin/env python3
"""Decode-sized Linear4bit launches (fq_int4_skinny_linear_f16) for rocprofv3 --kernel-trace: the four GEMM shapes of a Llama-3-8B layer
at M = 1, 16, 64, 128 rows, 200 launches each over rotating weight images (the seven linears of a layer never re-read a weight within a step)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flatquant_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
for N, K in ((4096, 4096), (1024, 4096), (14336, 4096), (4096, 14336)):
    ws = [torch.randint(0, 256, (N, K // 2), generator=g, device="cuda", dtype=torch.uint8) for _ in range(4)]
    imgs = [ops.int4_to_frag(w) for w in ws]
    wsc = torch.full((N,), 0.01, device="cuda", dtype=torch.float16)
    for M in (1, 16, 64, 128):
        x = torch.randint(0, 256, (M, K // 2), generator=g, device="cuda", dtype=torch.uint8)
        xs = torch.full((M,), 0.02, device="cuda", dtype=torch.float16)
        for i in range(200):
            ops.int4_skinny_linear(x, xs, imgs[i % 4], wsc, None, N)
        torch.cuda.synchronize()
print("done")
