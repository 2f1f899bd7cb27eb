#!/usr/bin/env python3
"""Per-phase cycle accounting of the d=4096 packed kernel (debug entry fq_debug_kron64_trace; GPU box only)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flatquant_amd._lib import check, lib  # noqa: E402

ROWS = 16384
fn = lib.fq_debug_kron64_trace
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64] + [ctypes.c_void_p] * 4
g = torch.Generator(device="cuda").manual_seed(0)
xs = [torch.randn(ROWS, 4096, generator=g, device="cuda", dtype=torch.float16) for _ in range(4)]
L = (torch.randn(64, 64, generator=g, device="cuda") / 8).half()
R = (torch.randn(64, 64, generator=g, device="cuda") / 8).half()
q = torch.empty(ROWS, 2048, dtype=torch.uint8, device="cuda")
s = torch.empty(ROWS, dtype=torch.float16, device="cuda")
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
n_waves = 16 * min((ROWS + 15) // 16, n_cu)
trace = torch.zeros(n_waves * 6, dtype=torch.int64, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for i in range(8):
    check(fn(xs[i % 4].data_ptr(), L.data_ptr(), R.data_ptr(), ROWS, q.data_ptr(), s.data_ptr(), trace.data_ptr(), st))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
check(fn(xs[1].data_ptr(), L.data_ptr(), R.data_ptr(), ROWS, q.data_ptr(), s.data_ptr(), trace.data_ptr(), st))
e1.record()
torch.cuda.synchronize()
stamps = trace[n_waves * 4:].cpu().reshape(n_waves, 2).double()
t = trace[:n_waves * 4].cpu().reshape(n_waves, 4).double()
tok_per_wave = ROWS / n_waves
names = ["wait X (vmcnt)", "GEMM1 + prefetch issue + cvt", "GEMM2 + stats reduce", "quant/pack/store"]
print(f"kernel {e0.elapsed_time(e1) * 1e3:.1f} us, {n_waves} waves, {tok_per_wave:.2f} tokens/wave; s_memtime ticks:")
tot = t.sum(dim=1)
for k, nm in enumerate(names):
    print(f"  {nm:32s} mean/token {t[:, k].mean() / tok_per_wave:9.0f}   share {100 * t[:, k].sum() / tot.sum():5.1f}%")
print(f"  total per wave: mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f} ticks")
span = stamps[:, 1].max() - stamps[:, 0].min()
print(f"  first wave start -> last wave end: {span:.0f} ticks = {span / (e0.elapsed_time(e1) * 1e3):.1f} ticks/us (kernel time incl. launch)")
life = stamps[:, 1] - stamps[:, 0]
print(f"  wave lifetime mean {life.mean():.0f} min {life.min():.0f} max {life.max():.0f}; prologue+tail = lifetime - loop: mean {(life - tot).mean():.0f}")
