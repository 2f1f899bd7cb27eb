"""deploy Quantizer / fp32 rowquant timings (16384 rows), FQHIP_LIB selects the library."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from flatquant_amd import ops
P, F, Q16, S16 = 1, 2, 0x20, 0x400
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(fn, steps=60, warm=10):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3
out = []
for d in (4096, 8192, 14336, 28672):
    rows = 16384 if d <= 14336 else 8192
    xs = [torch.randn(rows, d, generator=g, device="cuda", dtype=torch.float16) for _ in range(3)]
    a = timeit(lambda i: ops.rowquant(xs[i % 3], [(0.98, 0.98)], P))
    b = timeit(lambda i: ops.rowquant(xs[i % 3], [(0.98, 0.98)], P | Q16 | S16))
    c = timeit(lambda i: ops.rowquant(xs[i % 3], [(0.98, 0.98)], F))
    out.append(f"d={d}: fp32 packed {a:6.1f}  fp16 Quantizer {b:6.1f}  fake-quant {c:6.1f}")
    del xs
print(" | ".join(out))
