"""K / V quantise+pack timing vs row count (rows of 128), FQHIP_LIB selects the library."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from flatquant_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
NB = 3
T = (torch.randn(128, 128, generator=g, device="cuda") / 128 ** 0.5).half()
def timeit(fn, steps=100, warm=10):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3
for rows in (16384, 131072, 524288, 2097152):
    ks = [torch.randn(rows, 128, generator=g, device="cuda", dtype=torch.float16) for _ in range(NB)]
    k, v = timeit(lambda i: ops.kv_quant(ks[i % NB], T)), timeit(lambda i: ops.kv_quant(ks[i % NB]))
    print(f"rows {rows:8d}: K {k:7.1f} us ({rows * 324 / k / 1e6:5.2f} TB/s)   V {v:7.1f} us ({rows * 324 / v / 1e6:5.2f} TB/s)")
    del ks
