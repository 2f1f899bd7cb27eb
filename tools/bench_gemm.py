#!/usr/bin/env python3
"""Linear4bit GEMM: int8-path kernel vs FP6-path kernel (+ the activation conversion launch) on one MI355X."""
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


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    M = 16384
    for N, K in ((4096, 4096), (1024, 4096), (14336, 4096), (4096, 14336), (8192, 8192)):
        x = torch.randint(0, 256, (M, K // 2), generator=g, device="cuda", dtype=torch.uint8)
        w = torch.randint(0, 256, (N, K // 2), generator=g, device="cuda", dtype=torch.uint8)
        sx = torch.rand(M, generator=g, device="cuda").half() * 0.01
        sw = torch.rand(N, generator=g, device="cuda").half() * 0.01
        t8 = timeit(lambda: ops.int4_linear(x, sx, w, sw, None))
        wb = ops.int4_to_bf6(w, weights=True)
        xb = ops.int4_to_bf6(x)
        tc = timeit(lambda: ops.int4_to_bf6(x))
        t6 = timeit(lambda: ops.bf6_linear(xb, sx, wb, sw, None, M, N, K))
        ok = torch.equal(ops.bf6_linear(xb, sx, wb, sw, None, M, N, K), ops.int4_linear(x, sx, w, sw, None))
        pops = 2.0 * M * N * K
        print(f"M={M} N={N} K={K}: int8 path {t8:8.1f} us ({pops / t8 / 1e9:5.2f} Pop/s) | FP6 path {t6:8.1f} us "
              f"({pops / t6 / 1e9:5.2f} Pop/s) + convert x {tc:6.1f} us -> {pops / (t6 + tc) / 1e9:5.2f} Pop/s   identical={ok}")


if __name__ == "__main__":
    main()
