#!/usr/bin/env python3
"""Timing table for the other hot-path rows / BASELINE configs (C3-C5 pieces) on one MI355X.
Each line: kernel, shape, us per launch (HIP events, 50 launches, buffers rotated), algorithmic GB/s, Melem/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flatquant_amd import ops  # noqa: E402
from flatquant_amd._lib import FQ_NO_CLAMP0, FQ_OUT_FAKEQUANT, FQ_OUT_PACKED, FQ_QUANT_F16, FQ_ROUND_Y_F16  # noqa: E402
from flatquant_amd.flatquant.function_utils import get_decompose_dim  # noqa: E402
from flatquant_amd.flatquant.hadamard_utils import get_hadK  # noqa: E402

ROWS = 16384
NB = 3


def timeit(fn, steps=50, warm=10):
    """median of per-launch event pairs (round 5: the mean of 50 launches after 5 warm-ups, with the outputs allocated inside every
    call, put two 4x outliers into round 4's tables — 144x192 at 792 us, 168x176 at 1117 — that the kernel does not have:
    profiles/r05_tiles_outliers.txt, 2050 launches each under rocprofv3: max / avg = 1.47 / 1.30). Also prints nothing else: callers
    that want the spread use tools/time_dist.py."""
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        fn(i)
        ev[i + 1].record()
    torch.cuda.synchronize()
    us = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(steps))
    return us[len(us) // 2]


def line(name, shape, us, bytes_per_row, d, rows=ROWS):
    print(f"{name:34s} {shape:22s} {us:9.1f} us  {rows * bytes_per_row / us / 1e3:8.0f} GB/s  {rows * d / us:10.0f} Melem/s")


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    sig = [(0.982, 0.982)]   # sigmoid(4.0): the reference's initial clip factors (no digit outside [-8, 7])
    if os.environ.get("SIG"):  # e.g. SIG=0.9: trained clip factors, the quantiser's clamp route
        sig = [(float(os.environ["SIG"]),) * 2]
    for d in (4096, 8192, 14336, 28672, 11008, 7168, 2048, 18432, 5120, 13824, 8960, 18944, 27648, 29568):  # (18432 = DeepSeek-V3 dense ffn; the last six: Qwen2.5 widths)
        M, N = get_decompose_dim(d)
        rows = ROWS if d <= 14336 else ROWS // 2   # (8192 tokens for the wide ones)
        xs = [torch.randn(rows, d, generator=g, device="cuda", dtype=torch.float16) for _ in range(NB)]
        L = (torch.randn(M, M, generator=g, device="cuda") / M ** 0.5).half()
        R = (torch.randn(N, N, generator=g, device="cuda") / N ** 0.5).half()
        us = timeit(lambda i: ops.kron_quant(xs[i % NB], L, R, sig, FQ_OUT_PACKED | FQ_NO_CLAMP0))
        line("kron+quant packed", f"d={d} ({M}x{N}) rows={rows}", us, 2.5 * d + 2, d, rows)
        if os.environ.get("KRON_ONLY"):
            del xs
            continue
        if d in (4096, 14336):
            us = timeit(lambda i: ops.kron_quant(xs[i % NB], L, R, sig, FQ_OUT_FAKEQUANT | FQ_ROUND_Y_F16))
            line("kron+fake-quant fp16", f"d={d} ({M}x{N}) rows={rows}", us, 4.0 * d, d, rows)
            us = timeit(lambda i: ops.kron_quant(xs[i % NB], L, R, sig * 3, FQ_OUT_PACKED | FQ_NO_CLAMP0))
            line("kron+quant packed, 3 clip sets", f"d={d} ({M}x{N}) rows={rows}", us, 2.0 * d + 3 * (0.5 * d + 2), d, rows)
        if d in (4096, 14336, 28672, 11008):
            hk, K = get_hadK(d)
            hk = None if hk is None else hk.half().cuda()
            us = timeit(lambda i: ops.hadamard(xs[i % NB], K, hk))
            line("hadamard fp16 out", f"n={d} (K={K}) rows={rows}", us, 4.0 * d, d, rows)
            us = timeit(lambda i: ops.rowquant(xs[i % NB], sig, FQ_OUT_PACKED | FQ_QUANT_F16))
            line("rowquant packed (Quantizer)", f"cols={d} rows={rows}", us, 2.5 * d + 2, d, rows)
        del xs
    if os.environ.get("KRON_ONLY"):
        return
    if not os.environ.get("NO_GEMM"):
        # the consumer of the packed activations: INT4 x INT4 GEMM with the dequant epilogue (Linear4bit), 16384 tokens
        for N_, K_ in ((4096, 4096), (14336, 4096), (4096, 14336), (8192, 8192)):
            xq = torch.randint(0, 256, (ROWS, K_ // 2), generator=g, device="cuda", dtype=torch.uint8)
            wq = torch.randint(0, 256, (N_, K_ // 2), generator=g, device="cuda", dtype=torch.uint8)
            sx = torch.rand(ROWS, generator=g, device="cuda").half() * 0.01
            sw = torch.rand(N_, generator=g, device="cuda").half() * 0.01
            us = timeit(lambda i: ops.int4_linear(xq, sx, wq, sw), steps=20)
            tops = 2.0 * ROWS * N_ * K_ / us / 1e6
            print(f"{'int4 linear (gemm + dequant)':34s} {'M=%d N=%d K=%d' % (ROWS, N_, K_):22s} {us:9.1f} us  {tops:8.0f} TOP/s")
            us = timeit(lambda i: ops.int4_matmul(xq, wq), steps=20)
            print(f"{'int4 gemm -> int32':34s} {'M=%d N=%d K=%d' % (ROWS, N_, K_):22s} {us:9.1f} us  {2.0 * ROWS * N_ * K_ / us / 1e6:8.0f} TOP/s")
            wb, xb = ops.int4_to_bf6(wq, weights=True), ops.int4_to_bf6(xq)
            us = timeit(lambda i: ops.bf6_linear(xb, sx, wb, sw, None, ROWS, N_, K_), steps=20)
            uc = timeit(lambda i: ops.int4_to_bf6(xq), steps=20)
            print(f"{'int4 linear, FP6 matrix path':34s} {'M=%d N=%d K=%d' % (ROWS, N_, K_):22s} {us:9.1f} us  "
                  f"{2.0 * ROWS * N_ * K_ / us / 1e6:8.0f} TOP/s  (+ {uc:.1f} us to convert the activations)")
            del wb, xb
    # fusions either side of the path
    xs = [torch.randn(ROWS, 4096, generator=g, device="cuda", dtype=torch.float16) for _ in range(NB)]
    L = (torch.randn(64, 64, generator=g, device="cuda") / 8).half()
    R = (torch.randn(64, 64, generator=g, device="cuda") / 8).half()
    us = timeit(lambda i: ops.rmsnorm(xs[i % NB], 1e-5))
    line("rmsnorm", "d=4096", us, 4.0 * 4096, 4096)
    us = timeit(lambda i: ops.rmsnorm_kron_quant(xs[i % NB], 1e-5, L, R, sig, FQ_OUT_PACKED | FQ_NO_CLAMP0))
    line("rmsnorm + kron + quant, one launch", "d=4096 (64x64)", us, 2.5 * 4096 + 2, 4096)
    us = timeit(lambda i: ops.rmsnorm_kron_quant(xs[i % NB], 1e-5, L, R, sig * 3, FQ_OUT_PACKED | FQ_NO_CLAMP0))
    line("rmsnorm + kron + 3 clip sets", "d=4096 (64x64)", us, 2.0 * 4096 + 3 * (2048 + 2), 4096)
    del xs
    gs = [torch.randn(ROWS, 14336, generator=g, device="cuda", dtype=torch.float16) for _ in range(2)]
    up = torch.randn(ROWS, 14336, generator=g, device="cuda", dtype=torch.float16)
    M_, N_ = get_decompose_dim(14336)
    L = (torch.randn(M_, M_, generator=g, device="cuda") / M_ ** 0.5).half()
    R = (torch.randn(N_, N_, generator=g, device="cuda") / N_ ** 0.5).half()
    us = timeit(lambda i: ops.silu_mul(gs[i % 2], up))
    line("silu.mul", "d=14336", us, 6.0 * 14336, 14336)
    us = timeit(lambda i: ops.silu_mul_kron_quant(gs[i % 2], up, L, R, sig, FQ_OUT_PACKED | FQ_NO_CLAMP0))
    line("silu.mul + kron + quant, one launch", f"d=14336 ({M_}x{N_})", us, 4.5 * 14336 + 2, 14336)
    hk, K = get_hadK(14336)
    hk = hk.half().cuda()   # (once: the Kronecker factors of this rotation are cached per hadK tensor)
    us = timeit(lambda i: ops.hadamard_quant(gs[i % 2], K, hk, sig[0], up=up))
    line("silu.mul + hadamard + quantizer", "n=14336 (K=28)", us, 4.5 * 14336 + 2, 14336)
    us = timeit(lambda i: ops.hadamard_quant(gs[i % 2], K, hk, sig[0]))
    line("hadamard + quantizer, one launch", "n=14336 (K=28)", us, 2.5 * 14336 + 2, 14336)
    del gs, up
    for n in (11008, 8960):   # Llama-2-7B / Qwen2.5-1.5B ffn: K = 172 / 140, the rotation as a 172 x 64 / 140 x 64 Kronecker launch
        gs = [torch.randn(ROWS, n, generator=g, device="cuda", dtype=torch.float16) for _ in range(2)]
        hk, K = get_hadK(n)
        hk = hk.half().cuda()
        us = timeit(lambda i: ops.hadamard_quant(gs[i % 2], K, hk, sig[0]))
        line("hadamard + quantizer, one launch", f"n={n} (K={K})", us, 2.5 * n + 2, n)
        us = timeit(lambda i: ops.hadamard_quant(gs[i % 2], K, hk, sig[0], fwht_route=True))
        line("hadamard + quantizer, FWHT route", f"n={n} (K={K})", us, 2.5 * n + 2, n)
        del gs
    ks = [torch.randn(ROWS * 8, 128, generator=g, device="cuda", dtype=torch.float16) for _ in range(NB)]
    T = (torch.randn(128, 128, generator=g, device="cuda") / 128 ** 0.5).half()
    us = timeit(lambda i: ops.kv_quant(ks[i % NB], T))
    line("K transform + asym INT4 pack", "16384 x 8 heads x 128", us, 2.5 * 128 + 4, 128, ROWS * 8)
    us = timeit(lambda i: ops.kv_quant(ks[i % NB]))
    line("V asym INT4 pack", "16384 x 8 heads x 128", us, 2.5 * 128 + 4, 128, ROWS * 8)
    del ks
    # paged INT4 cache: decode attention, one query token per request (kv_cache.py batch_decode_i4)
    for bsz, seq in ((16, 2048), (64, 2048), (8, 8192)):
        heads, hd, page = 32, 128, 2048
        n_pg = (seq + page - 1) // page
        data = torch.randint(0, 256, (bsz * n_pg, 1, 2, heads, page, hd // 2), generator=g, device="cuda", dtype=torch.uint8)
        par = (torch.rand(bsz * n_pg, 1, 2, heads, page, 2, generator=g, device="cuda") * 0.2 + 0.05).half()
        indptr = (torch.arange(bsz + 1, device="cuda", dtype=torch.int32) * n_pg)
        indices = torch.arange(bsz * n_pg, device="cuda", dtype=torch.int32)
        last = torch.full((bsz,), (seq - 1) % page + 1, device="cuda", dtype=torch.int32)
        q = torch.randn(bsz, heads, hd, generator=g, device="cuda").half()
        us = timeit(lambda i: ops.kv_batch_decode(q, data, par, indptr, indices, last, 0))
        byt = bsz * heads * seq * 2 * (hd // 2 + 4)
        print(f"{'INT4 paged decode attention':34s} {'bsz=%d seq=%d heads=32' % (bsz, seq):22s} {us:9.1f} us  {byt / us / 1e3:8.0f} GB/s")
        del data
        if (bsz, seq) == (16, 2048):   # the fp16 configuration of the same cache (disable_quant=True)
            data16 = torch.randn(bsz * n_pg, 1, 2, heads, page, hd, generator=g, device="cuda").half()
            us = timeit(lambda i: ops.kv_batch_decode(q, data16, par, indptr, indices, last, 0))
            print(f"{'fp16 paged decode attention':34s} {'bsz=%d seq=%d heads=32' % (bsz, seq):22s} {us:9.1f} us  {bsz * heads * seq * 2 * hd * 2 / us / 1e3:8.0f} GB/s")
            del data16
        del par
    for hd, H in ((128, 32), (128, 64)):
        xs = [torch.randn(ROWS, hd, H, generator=g, device="cuda", dtype=torch.float16) for _ in range(NB)]
        P = (torch.randn(H, H, generator=g, device="cuda") / H ** 0.5).half()
        us = timeit(lambda i: ops.block_quant(xs[i % NB], P, sig, FQ_OUT_PACKED | FQ_NO_CLAMP0, True))
        line("block (o_proj) + quant packed", f"hd={hd} H={H}", us, 2.5 * hd * H + 2, hd * H)
        del xs


if __name__ == "__main__":
    main()
