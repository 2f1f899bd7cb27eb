#!/usr/bin/env python3
"""Timing ladder on the GPU box: the fused kernel's output variants next to HBM-only kernels moving the same
bytes (rowquant: 8 KB in / 2 KB out per token; a plain device copy)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flatquant_amd import _probe, ops  # noqa: E402
from flatquant_amd._lib import (FQ_NO_CLAMP0, FQ_OUT_FAKEQUANT, FQ_OUT_PACKED, FQ_OUT_TRANSFORM,  # noqa: E402
                                FQ_QUANT_F16, FQ_ROUND_Y_F16)

ROWS, D, NB = 16384, 4096, 4


def timeit(fn, steps=100, warm=10):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3   # us


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    xs = [torch.randn(ROWS, D, generator=g, device="cuda", dtype=torch.float16) for _ in range(NB)]
    L = (torch.randn(64, 64, generator=g, device="cuda") / 8).half()
    R = (torch.randn(64, 64, generator=g, device="cuda") / 8).half()
    sig = [(0.982, 0.982)]
    ys = [torch.empty_like(xs[0]) for _ in range(NB)]
    cases = {
        "kron packed (P|NC0)": lambda i: ops.kron_quant(xs[i % NB], L, R, sig, FQ_OUT_PACKED | FQ_NO_CLAMP0),
        "kron packed x3 clips": lambda i: ops.kron_quant(xs[i % NB], L, R, sig * 3, FQ_OUT_PACKED),
        "kron fakequant (F|R16)": lambda i: ops.kron_quant(xs[i % NB], L, R, sig, FQ_OUT_FAKEQUANT | FQ_ROUND_Y_F16),
        "kron transform only (T)": lambda i: ops.kron_quant(xs[i % NB], L, R, flags=FQ_OUT_TRANSFORM),
        "rowquant packed fp32": lambda i: ops.rowquant(xs[i % NB], sig, FQ_OUT_PACKED),
        "rowquant packed f16 (Quantizer)": lambda i: ops.rowquant(xs[i % NB], sig, FQ_OUT_PACKED | FQ_QUANT_F16),
        "rowquant fakequant": lambda i: ops.rowquant(xs[i % NB], sig, FQ_OUT_FAKEQUANT),
        "torch copy 128MiB": lambda i: ys[i % NB].copy_(xs[(i + 1) % NB]),
    }
    qb = torch.empty(ROWS, D // 2, dtype=torch.uint8, device="cuda")
    sb = torch.empty(ROWS, dtype=torch.float16, device="cuda")
    for w in (2, 3, 4, 8):
        cases[f"stream probe 8K in/2K out, {w} waves/SIMD"] = (lambda i, w=w: _probe.probe_stream_4096(xs[i % NB], qb, sb, w))
    for name, fn in cases.items():
        us = timeit(fn)
        print(f"{name:34s} {us:9.1f} us   {ROWS * D / us:12.0f} Melem/s")


if __name__ == "__main__":
    main()
