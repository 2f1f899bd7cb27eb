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
trace = torch.zeros(n_waves * 14, dtype=torch.int64, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for i in range(8):
    check(fn(xs[i % 4].data_ptr(), L.data_ptr(), R.data_ptr(), ROWS, q.data_ptr(), s.data_ptr(), trace.data_ptr(), st))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
check(fn(xs[1].data_ptr(), L.data_ptr(), R.data_ptr(), ROWS, q.data_ptr(), s.data_ptr(), trace.data_ptr(), st))
e1.record()
torch.cuda.synchronize()
stamps = trace[n_waves * 4:].cpu().reshape(n_waves, 10).double()
t = trace[:n_waves * 4].cpu().reshape(n_waves, 4).double()
tok_per_wave = ROWS / n_waves
names = ["wait X (vmcnt)", "GEMM1 + prefetch issue + cvt", "GEMM2 + stats reduce", "quant/pack/store"]
print(f"kernel {e0.elapsed_time(e1) * 1e3:.1f} us, {n_waves} waves, {tok_per_wave:.2f} tokens/wave; s_memtime ticks:")
tot = t.sum(dim=1)
for k, nm in enumerate(names):
    print(f"  {nm:32s} mean/token {t[:, k].mean() / tok_per_wave:9.0f}   share {100 * t[:, k].sum() / tot.sum():5.1f}%")
print(f"  total per wave: mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f} ticks")
life = stamps[:, 1] - stamps[:, 0]
life_ns = (stamps[:, 5] - stamps[:, 4]) * 10.0
ghz = (life / life_ns).mean()
blk = stamps.reshape(-1, 16, 10)
print(f"  shader clock from wave lifetimes: {ghz:.3f} GHz (s_memtime ticks per ns of s_memrealtime)")
print(f"  wave lifetime mean {life_ns.mean() / 1e3:.2f} us min {life_ns.min() / 1e3:.2f} max {life_ns.max() / 1e3:.2f}; "
      f"token loop {tot.mean() / ghz / 1e3:.2f} us")
print(f"  prologue per wave: start -> kernargs {(stamps[:, 6] - stamps[:, 0]).mean() / ghz / 1e3:.2f} us -> loads issued "
      f"{(stamps[:, 7] - stamps[:, 6]).mean() / ghz / 1e3:.2f} us -> gather landed {(stamps[:, 2] - stamps[:, 7]).mean() / ghz / 1e3:.2f} us "
      f"(slowest wave of a workgroup {((blk[:, :, 2] - blk[:, :, 7]).max(dim=1).values).mean() / ghz / 1e3:.2f} us)")
print(f"  prologue per wave: start -> gather done {(stamps[:, 2] - stamps[:, 0]).mean() / ghz / 1e3:.2f} us, "
      f"gather done -> barrier passed {(stamps[:, 3] - stamps[:, 2]).mean() / ghz / 1e3:.2f} us")
t0 = stamps[:, 4].min()
first = blk[:, :, 4].min(dim=1).values
print(f"  workgroup's first wave starts {10 * (first - t0).mean() / 1e3:.2f} us (max {10 * (first - t0).max() / 1e3:.2f}) after the "
      f"earliest; its last wave {10 * (blk[:, :, 4].max(dim=1).values - first).mean() / 1e3:.2f} us after its first")
print(f"  wave ends: mean {10 * (stamps[:, 5] - t0).mean() / 1e3:.2f} us, last {10 * (stamps[:, 5] - t0).max() / 1e3:.2f} us after "
      f"the earliest start (event time of the launch {e0.elapsed_time(e1) * 1e3:.1f} us)")
bend = 10 * (blk[:, :, 5].max(dim=1).values - t0) / 1e3          # per-workgroup end time (us)
bq = torch.quantile(bend, torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64))
print("  workgroup end times (us): min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % tuple(bq.tolist()))
wend = 10 * (blk[:, :, 5] - t0) / 1e3
print(f"  spread of wave ends inside a workgroup: mean {(wend.max(dim=1).values - wend.min(dim=1).values).mean():.1f} us")
nb = bend.numel()
print("  mean workgroup end by (workgroup id % 8):", " ".join(f"{bend[i::8].mean():.1f}" for i in range(8)))
loop_per_blk = t.reshape(-1, 16, 4).sum(dim=(1, 2)) / ghz / 1e3 / 16
print(f"  mean in-loop time per wave, by workgroup: min {loop_per_blk.min():.1f} median {loop_per_blk.median():.1f} max {loop_per_blk.max():.1f} us")
print("  prologue stamps relative to the workgroup's first wave start: mean over waves / mean over workgroups of the slowest wave (us)")
base = blk[:, :, 0].min(dim=1, keepdim=True).values
for nm, k in (("kernargs loaded", 6), ("prologue loads issued", 7), ("gather landed + frag written", 2), ("at the barrier", 8), ("barrier passed", 3)):
    rel = (blk[:, :, k] - base) / ghz / 1e3
    print(f"    {nm:30s} {rel.mean():6.2f} / {rel.max(dim=1).values.mean():6.2f}")
