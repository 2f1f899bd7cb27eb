"""Row quantisers on SHORT rows (cols = 128: head-sized rows of the cache quantisers, 128-element groups), FQHIP_LIB selects the library."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from flatquant_amd import ops
P, F, Q16, ASYM = 1, 2, 0x20, 0x800
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(fn, steps=40, warm=5):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3
for rows, cols in ((16384 * 56, 128), (16384 * 32, 128), (16384 * 8, 128), (16384 * 16, 64)):
    xs = [torch.randn(rows, cols, generator=g, device="cuda", dtype=torch.float16) for _ in range(2)]
    out = []
    for name, fl in (("packed fp32", P), ("fake fp32", F), ("fake fp16", F | Q16), ("asym fake fp32", F | ASYM), ("asym fake fp16", F | ASYM | Q16)):
        us = timeit(lambda i: ops.rowquant(xs[i % 2], [(0.98, 0.98)], fl))
        bpe = 2.5 if fl & P else 4.0
        out.append(f"{name} {us:7.1f} us ({rows * cols * bpe / us / 1e6:4.2f} TB/s)")
    print(f"rows {rows} x {cols}: " + " | ".join(out))
    del xs
