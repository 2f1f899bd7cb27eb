#!/usr/bin/env python3
"""Kernel time of fq_kron_quant_{f16,bf16} through the C ABI with pre-allocated, rotating buffers (no torch allocation,
no Python module path inside the timed loop), one line per case:
    tools/time_kron.py [M N rows mode dtype] ...      mode = packed | packedr | fq | y | fqy | packed3 | h16   dtype = f16 | bf16
    (h16: fq_kron_quant_ex_f16 with the deploy Quantizer's fp16 arithmetic and a post-scale — a Hadamard rotation as one launch)
Without arguments: the table of the fake-quant contract (FlatQuantizedLinear._eval_forward) next to the packed one.
Prints us per launch (HIP events over 100 launches, median of 5 rounds), algorithmic GB/s and the fraction of 8 TB/s."""
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flatquant_amd import _lib  # noqa: E402

lib = _lib.lib
P4 = ctypes.c_void_p * 4
F4 = ctypes.c_float * 4
FLAGS = {"packed": 0x01 | 0x10, "fq": 0x02 | 0x08, "y": 0x04, "fqy": 0x02 | 0x04 | 0x08, "packed3": 0x01 | 0x10,
         "packedr": 0x01 | 0x08,   # the deploy flags of a pair with M > 64 (deploy_kron_flags): Y through fp16, extrema clamped
         "h16": 0x01 | 0x08 | 0x20 | 0x400}
BYTES = {"packed": lambda d: 2.5 * d + 2, "fq": lambda d: 4.0 * d, "y": lambda d: 4.0 * d, "fqy": lambda d: 6.0 * d,
         "packed3": lambda d: 2.0 * d + 3 * (0.5 * d + 2), "packedr": lambda d: 2.5 * d + 2, "h16": lambda d: 2.5 * d + 2}


def time_case(M, N, rows, mode, dtype="f16", rounds=5, steps=100, sig=float(os.environ.get("SIG", "0.9820137619972229"))):   # SIG=0.7: the clamp path
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    fn = lib.fq_kron_quant_bf16 if dtype == "bf16" else lib.fq_kron_quant_f16
    d = M * N
    g = torch.Generator(device="cuda").manual_seed(0)
    nb = max(2, min(4, int((600 << 20) // (rows * d * 2))))          # rotate over > 256 MB where memory allows
    xs = [torch.randn(rows, d, generator=g, device="cuda", dtype=torch.float32).to(td) for _ in range(nb)]
    L = (torch.randn(M, M, generator=g, device="cuda") / M ** 0.5).to(td)
    R = (torch.randn(N, N, generator=g, device="cuda") / N ** 0.5).to(td)
    nclip = 3 if mode == "packed3" else 1
    qs = [[torch.empty(rows, d // 2, dtype=torch.uint8, device="cuda") for _ in range(nclip)] for _ in range(nb)] if mode in ("packed", "packedr", "packed3", "h16") else None
    ss = [torch.empty(rows, dtype=td, device="cuda") for _ in range(nclip)]
    fqs = [torch.empty(rows, d, dtype=td, device="cuda") for _ in range(nb)] if "fq" in mode else None
    ys = [torch.empty(rows, d, dtype=td, device="cuda") for _ in range(nb)] if "y" in mode else None
    wsb = lib.fq_kron_workspace_bytes(M, N)
    ws = torch.empty(max(int(wsb), 16), dtype=torch.uint8, device="cuda")
    if wsb > 0:
        _lib.check((lib.fq_kron_prepare_bf16 if dtype == "bf16" else lib.fq_kron_prepare_f16)(
            L.data_ptr(), R.data_ptr(), M, N, ws.data_ptr(), int(wsb), None))
    smax, smin = F4(*[sig] * 4), F4(*[sig] * 4)
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    flags = FLAGS[mode] | (0x40 if wsb > 0 else 0)

    def launch(i):
        qa, sa, fa = P4(), P4(), P4()
        if qs:
            for c in range(nclip):
                qa[c], sa[c] = qs[i % nb][c].data_ptr(), ss[c].data_ptr()
        if fqs:
            fa[0] = fqs[i % nb].data_ptr()
        if mode == "h16":
            rc = lib.fq_kron_quant_ex_f16(xs[i % nb].data_ptr(), None, L.data_ptr(), R.data_ptr(), rows, M, N, ctypes.c_float(0.0762),
                                          smax, smin, nclip, flags, qa, sa, fa, None, ws.data_ptr() if wsb > 0 else None,
                                          int(max(wsb, 0)), sp)
        else:
            rc = fn(xs[i % nb].data_ptr(), L.data_ptr(), R.data_ptr(), None, rows, M, N, smax, smin, nclip, flags, qa, sa, fa,
                    ys[i % nb].data_ptr() if ys else None, ws.data_ptr() if wsb > 0 else None, int(max(wsb, 0)), sp)
        _lib.check(rc)

    for i in range(30):
        launch(i)
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            launch(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / steps * 1e3)
    us = statistics.median(ts)
    gbs = rows * BYTES[mode](d) / us / 1e3
    print(f"{mode:8s} {dtype:4s} {M:3d}x{N:<3d} rows={rows:5d}  {us:8.1f} us  {gbs:7.0f} GB/s  {gbs / 8000:5.3f} of 8 TB/s  (min {min(ts):.1f})", flush=True)
    return us


if __name__ == "__main__":
    a = sys.argv[1:]
    if a:
        while a:
            M, N, rows, mode, dtype = int(a[0]), int(a[1]), int(a[2]), a[3], a[4]
            time_case(M, N, rows, mode, dtype)
            a = a[5:]
    else:
        for (M, N, rows) in ((64, 64, 16384), (112, 128, 16384), (64, 128, 16384), (64, 112, 16384), (32, 64, 16384), (128, 224, 8192)):
            for mode in ("packed", "fq", "y", "fqy"):
                for dtype in ("f16", "bf16"):
                    time_case(M, N, rows, mode, dtype)
