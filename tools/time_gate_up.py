#!/usr/bin/env python3
"""The gate / up pair of a gated MLP on the FP6 GEMM path with kept weight images, timed through the library
entry points (activation conversion included, as Linear4bit.forward runs it):
  two launches   fq_int4_linear_fp6_f16 per projection (the reference's structure)
  multi, own x   fq_int4_linear_fp6_multi_f16, each projection its own packed input (FlatQuant: two clip pairs)
  multi, one x   the same with one packed input for both (the fuseLN branch)
  gate_up        fq_int4_linear_fp6_gate_up_f16 (SiLU.mul in the epilogue), own x / one x
  silu_mul       fq_silu_mul_f16 over the two [M, N] results (what the separate path adds unless the down transform absorbs it)
usage: tools/time_gate_up.py [M N K] ...   (default: 16384 14336 4096 and 2048 11008 4096)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flatquant_amd import ops  # noqa: E402


def timeit(fn, steps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


args = [int(v) for v in sys.argv[1:]] or [16384, 14336, 4096, 2048, 11008, 4096]
g = torch.Generator(device="cuda").manual_seed(0)
for M, N, K in zip(args[0::3], args[1::3], args[2::3]):
    pr = []
    for p in range(2):
        x = torch.randint(0, 256, (M, K // 2), generator=g, device="cuda", dtype=torch.uint8)
        w = torch.randint(0, 256, (N, K // 2), generator=g, device="cuda", dtype=torch.uint8)
        sx = (torch.rand(M, generator=g, device="cuda") * 0.01).half()
        sw = (torch.rand(N, generator=g, device="cuda") * 0.01).half()
        pr.append((x, sx, w, ops.int4_to_bf6(w, weights=True), sw, None))
    shared = [pr[0], (pr[0][0], pr[0][1]) + pr[1][2:]]
    ys = ops.int4_linear_fp6_multi(pr)
    two = timeit(lambda: [ops.int4_linear_fp6(*q) for q in pr])
    print(f"M={M} N={N} K={K}:  two launches {two:8.1f} us   multi own x {timeit(lambda: ops.int4_linear_fp6_multi(pr)):8.1f} us   multi one x {timeit(lambda: ops.int4_linear_fp6_multi(shared)):8.1f} us   "
          f"gate_up own x {timeit(lambda: ops.int4_linear_fp6_gate_up(pr[0], pr[1])):8.1f} us   gate_up one x {timeit(lambda: ops.int4_linear_fp6_gate_up(shared[0], shared[1])):8.1f} us   "
          f"silu_mul {timeit(lambda: ops.silu_mul(ys[0], ys[1])):7.1f} us   x conversion {timeit(lambda: ops.int4_to_bf6(pr[0][0])):6.1f} us", flush=True)
