#!/usr/bin/env python3
"""Launch time of the d=4096 packed kernel vs token count: T(rows) = fixed + slope * rows (GPU box only).

Separates the per-launch overhead (dispatch, fragment-image build, ramp and tail) from the streaming rate.
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flatquant_amd._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED, check, lib  # noqa: E402

MAXR = 131072
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(MAXR, 4096, generator=g, device="cuda", dtype=torch.float16)
L = (torch.randn(64, 64, generator=g, device="cuda") / 8).half()
R = (torch.randn(64, 64, generator=g, device="cuda") / 8).half()
q = torch.empty(MAXR, 2048, dtype=torch.uint8, device="cuda")
s = torch.empty(MAXR, dtype=torch.float16, device="cuda")
sig = 0.9820137619972229
smax, smin = (ctypes.c_float * 4)(sig), (ctypes.c_float * 4)(sig)
qa, sa, none4 = (ctypes.c_void_p * 4)(), (ctypes.c_void_p * 4)(), (ctypes.c_void_p * 4)()
qa[0], sa[0] = q.data_ptr(), s.data_ptr()
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lp, rp = (ctypes.c_void_p(t.data_ptr()) for t in (L, R))


def launch(rows, i=0):
    off = (i * rows) % (MAXR - rows + 1)          # walk through the 1 GB buffer: no launch re-reads a cached range
    xp = ctypes.c_void_p(x.data_ptr() + off * 8192)
    qa[0], sa[0] = q.data_ptr() + off * 2048, s.data_ptr() + off * 2
    check(lib.fq_kron_quant_f16(xp, lp, rp, None, rows, 64, 64, smax, smin, 1, FQ_OUT_PACKED | FQ_NO_CLAMP0,
                                qa, sa, none4, None, None, 0, sp))


pts = []
for rows in (256, 1024, 4096, 8192, 16384, 32768, 65536, 131072):
    n = 200 if rows <= 16384 else 50
    for _ in range(10):
        launch(rows)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        launch(rows, i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    pts.append((rows, us))
    print(f"rows {rows:7d}: {us:8.2f} us   {rows * 10242 / us / 1e6:6.2f} TB/s algorithmic")
(r0, t0), (r1, t1) = pts[-3], pts[-1]
slope = (t1 - t0) / (r1 - r0)
print(f"slope {slope * 1e3:.3f} ns/token = {10242 / slope / 1e6:.2f} TB/s streaming; "
      f"fixed part at 16384 rows: {dict(pts)[16384] - slope * 16384:.1f} us")

from flatquant_amd import _probe, ops  # noqa: E402

print("no-arithmetic streaming kernel (same bytes):")
for rows in (1024, 4096, 16384, 65536, 131072):
    n = 100 if rows <= 16384 else 30
    views = [(x[o:o + rows], q[o:o + rows], s[o:o + rows]) for o in
             [(i * rows) % (MAXR - rows + 1) for i in range(n)]]
    for v in views[:5]:
        _probe.probe_stream_4096(*v)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for v in views:
        _probe.probe_stream_4096(*v)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"rows {rows:7d}: {us:8.2f} us   {rows * 10242 / us / 1e6:6.2f} TB/s")
