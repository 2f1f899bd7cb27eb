#!/usr/bin/env python3
"""Fused FQ_GROUP128 epilogue against the two-launch composition it replaces (transform launch + row quantiser over the
(-1, 128) view), through flatquant_amd.ops (allocation included in both): 16384 tokens, fake-quant output."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flatquant_amd import ops  # noqa: E402

F, T, R16 = 0x02, 0x04, 0x08
g = torch.Generator(device="cuda").manual_seed(0)


def timeit(fn, n=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, dt) in ((64, 112, torch.bfloat16), (64, 112, torch.float16), (32, 64, torch.bfloat16), (64, 128, torch.float16), (112, 128, torch.float16)):
    rows = 16384
    x = torch.randn(rows, M * N, generator=g, device="cuda").to(dt)
    L = (torch.randn(M, M, generator=g, device="cuda") / M ** 0.5).to(dt)
    R = (torch.randn(N, N, generator=g, device="cuda") / N ** 0.5).to(dt)
    sig = [(0.98, 0.98)]
    fused = timeit(lambda: ops.kron_quant(x, L, R, sig, F | R16, groupsize=128))
    two = timeit(lambda: ops._quant_groups_of(ops.kron_quant(x, L, R, flags=T).y, sig, F | R16, 128, ops.FusedOutputs()))
    tok = timeit(lambda: ops.kron_quant(x, L, R, sig, F | R16))
    d = M * N
    print(f"{M}x{N} {str(dt)[6:]:8s}: fused group-128 {fused:7.1f} us ({rows * 4 * d / fused / 1e3:5.0f} GB/s)   two launches {two:7.1f} us   "
          f"per-token scales {tok:7.1f} us")
