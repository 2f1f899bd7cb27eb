import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flatquant_amd import ops
def timeit(fn, steps=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3
g = torch.Generator(device="cuda").manual_seed(0)
for M in (1, 8, 32, 64, 128):
    for N, K in ((4096, 4096), (14336, 4096), (4096, 14336)):
        x = torch.randint(0, 256, (M, K // 2), generator=g, device="cuda", dtype=torch.uint8)
        w = torch.randint(0, 256, (N, K // 2), generator=g, device="cuda", dtype=torch.uint8)
        sx = torch.rand(M, generator=g, device="cuda").half() * 0.01
        sw = torch.rand(N, generator=g, device="cuda").half() * 0.01
        t = timeit(lambda: ops.int4_linear(x, sx, w, sw, None))
        img = ops.int4_to_frag(w)
        t2 = timeit(lambda: ops.int4_skinny_linear(x, sx, img, sw, None, N))
        print(f"M={M:4d} N={N:6d} K={K:6d}: tile kernel {t:8.1f} us | skinny {t2:8.1f} us   weights {N*K/2/t2/1e3:7.0f} GB/s")
