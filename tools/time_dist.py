#!/usr/bin/env python3
"""Per-launch time DISTRIBUTION of fq_kron_quant_f16 through the C ABI with pre-allocated, rotating buffers (nothing is allocated
inside the timed loop): one HIP-event pair per launch, N launches per case.
    tools/time_dist.py [MxN:rows[:launches]] ...         default: the two pairs of VERDICT r04 weak #3 (144x192:8192:2000 168x176:8192:2000)
Prints min / p1 / median / p99 / max, the number of launches beyond 2x the median and where in the sequence they sit — a kernel with
a slow mode shows them spread over the run, a cold start shows them at the front. Run it under
`rocprofv3 --kernel-trace --stats` for the kernel's own durations (the event pairs include the gap in front of a launch)."""
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


def prepare(M, N, rows, flags=0x01 | 0x10, dtype=torch.float16, nclip=1):
    """-> launch(i): one pre-planned call of fq_kron_quant_f16 (buffers rotate over > 256 MB where memory allows)"""
    d = M * N
    g = torch.Generator(device="cuda").manual_seed(0)
    nb = max(2, min(4, int((600 << 20) // (rows * d * 2))))
    xs = [torch.randn(rows, d, generator=g, device="cuda", dtype=torch.float32).to(dtype) for _ in range(nb)]
    L = (torch.randn(M, M, generator=g, device="cuda") / M ** 0.5).to(dtype)
    R = (torch.randn(N, N, generator=g, device="cuda") / N ** 0.5).to(dtype)
    qs = [[torch.empty(rows, d // 2, dtype=torch.uint8, device="cuda") for _ in range(nclip)] for _ in range(nb)]
    ss = [torch.empty(rows, dtype=dtype, device="cuda") for _ in range(nclip)]
    wsb = lib.fq_kron_workspace_bytes(M, N)
    ws = torch.empty(max(int(wsb), 16), dtype=torch.uint8, device="cuda")
    if wsb > 0:
        _lib.check(lib.fq_kron_prepare_f16(L.data_ptr(), R.data_ptr(), M, N, ws.data_ptr(), int(wsb), None))
    sig = 0.9820137619972229
    smax, smin = F4(*[sig] * 4), F4(*[sig] * 4)
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    fl = flags | (0x40 if wsb > 0 else 0)
    args = []
    for b in range(nb):
        qa, sa = P4(), P4()
        for c in range(nclip):
            qa[c], sa[c] = qs[b][c].data_ptr(), ss[c].data_ptr()
        args.append((xs[b].data_ptr(), qa, sa))
    keep = (xs, L, R, qs, ss, ws)

    def launch(i):
        x, qa, sa = args[i % nb]
        _lib.check(lib.fq_kron_quant_f16(x, L.data_ptr(), R.data_ptr(), None, rows, M, N, smax, smin, nclip, fl, qa, sa, P4(), None,
                                         ws.data_ptr() if wsb > 0 else None, int(max(wsb, 0)), sp))
    launch.keep = keep
    return launch


def distribution(launch, n, warm=50):
    for i in range(warm):
        launch(i)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        launch(i)
        ev[i + 1].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]


def main():
    cases = []
    for spec in sys.argv[1:]:
        f = spec.split(":")
        M, N = (int(v) for v in f[0].split("x"))
        cases.append((M, N, int(f[1]), int(f[2]) if len(f) > 2 else 2000))
    if not cases:
        cases = [(144, 192, 8192, 2000), (168, 176, 8192, 2000)]
    for M, N, rows, n in cases:
        us = distribution(prepare(M, N, rows), n)
        s = sorted(us)
        med = statistics.median(us)
        slow = [i for i, u in enumerate(us) if u > 2 * med]
        q = lambda f: s[min(len(s) - 1, int(f * len(s)))]
        print(f"{M}x{N} rows={rows} launches={n}: min {s[0]:.1f}  p1 {q(0.01):.1f}  median {med:.1f}  p99 {q(0.99):.1f}  max {s[-1]:.1f} us;  "
              f"sigma {statistics.pstdev(us):.2f};  > 2x median: {len(slow)}" + (f" at {slow[:12]}" if slow else ""), flush=True)


if __name__ == "__main__":
    main()
