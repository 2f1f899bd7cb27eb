#!/usr/bin/env python3
"""Linear4bit groups (q / k / v; up / gate) on the FP6 path: ONE multi-problem launch (fq_int4_linear_fp6_multi_f16) against one launch per
projection (fq_int4_linear_fp6_f16), kept weight images, activation conversions included, at M = 512 .. 16384 rows. us per group."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flatquant_amd import ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(ts)[1]


for name, Ns, K in (("q/k/v", [4096, 1024, 1024], 4096), ("up/gate", [14336, 14336], 4096), ("q/k/v MHA", [4096, 4096, 4096], 4096)):
    ws = [torch.randint(0, 256, (N, K // 2), generator=g, device="cuda", dtype=torch.uint8) for N in Ns]
    fimg = [ops.int4_to_bf6(w, weights=True) for w in ws]
    wsc = [torch.full((N,), 0.01, device="cuda", dtype=torch.float16) for N in Ns]
    row = []
    for M in (512, 1024, 2048, 4096, 8192, 16384):
        xs = [torch.randint(0, 256, (M, K // 2), generator=g, device="cuda", dtype=torch.uint8) for _ in Ns]
        sx = torch.full((M,), 0.02, device="cuda", dtype=torch.float16)
        multi = lambda: ops.int4_linear_fp6_multi([(xs[p], sx, ws[p], fimg[p], wsc[p], None) for p in range(len(Ns))])
        sep = lambda: [ops.int4_linear_fp6(xs[p], sx, ws[p], fimg[p], wsc[p], None) for p in range(len(Ns))]
        reps = max(5, min(50, int(2e5 / M)))
        row.append(f"M={M}: multi {timeit(multi, reps):7.1f} | separate {timeit(sep, reps):7.1f}")
    print(f"{name:10s} " + "   ".join(row))
