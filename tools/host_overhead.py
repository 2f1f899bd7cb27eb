#!/usr/bin/env python3
"""Host-side cost of a library call (round 4): launches on 4-row inputs, so that the kernel is ~3 us and the
wall time per call is the Python / ctypes / allocator path. Prints us per call for the ops and the deploy modules of a decoder
layer, then a cProfile of ops.kron_quant.   python tools/host_overhead.py [--profile]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flatquant_amd import deploy, ops  # noqa: E402
from flatquant_amd._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(4, 4096, generator=g, device=dev, dtype=torch.float16)
xf = torch.randn(4, 14336, generator=g, device=dev, dtype=torch.float16)
L = (torch.randn(64, 64, generator=g, device=dev) / 8).half()
R = (torch.randn(64, 64, generator=g, device=dev) / 8).half()
L2 = (torch.randn(112, 112, generator=g, device=dev) / 10).half()
R2 = (torch.randn(128, 128, generator=g, device=dev) / 11).half()
SIG = [(0.98, 0.98)]
P = FQ_OUT_PACKED | FQ_NO_CLAMP0


def timeit(name, fn, n=20000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()      # host time to ISSUE n calls (the queue never fills: the kernels are shorter than the calls)
    torch.cuda.synchronize()
    print(f"{name:46s} {(t1 - t0) / n * 1e6:7.2f} us per call (host)", flush=True)


ot = deploy.nn.OnlineTrans(4096, trans="matmul", decompose=True, lac=True).to(dev)
ot.left_matrix.copy_(L), ot.right_matrix.copy_(R)
qz = deploy.nn.Quantizer(lac=True).to(dev)
lin = deploy.nn.Linear4bit(4096, 4096).to(dev)
x3 = x.reshape(1, 4, 4096)
p = ot(x3)
timeit("ops.kron_quant 64x64 packed", lambda: ops.kron_quant(x, L, R, SIG, P))
timeit("ops.kron_quant 112x128 packed", lambda: ops.kron_quant(xf, L2, R2, SIG, P))
timeit("ops.rowquant (deploy Quantizer arithmetic)", lambda: ops.rowquant(x, SIG, FQ_OUT_PACKED | 0x20 | 0x400))
# (round 5) the modules' DEFAULT forward: a C-side prepared call with fresh outputs (ops.FreshPlan, fq_plan_*)
timeit("deploy.nn.OnlineTrans(matmul).forward (default)", lambda: ot(x3))
timeit("deploy.nn.Quantizer(lac).forward (default)", lambda: qz(x))
timeit("deploy.nn.Linear4bit.forward, decode (default)", lambda: lin(p))
ot.fast_path = qz.fast_path = lin.fast_path = False   # the general entry points (round 4's default)
timeit("OnlineTrans(matmul).forward, fast_path = False", lambda: ot(x3))
timeit("Quantizer(lac).forward, fast_path = False", lambda: qz(x))
timeit("Linear4bit.forward, decode, fast_path = False", lambda: lin(p))
ot.fast_path = qz.fast_path = lin.fast_path = True
# (round 4) prepared launches with static outputs (ops.LaunchPlan; the modules' opt-in static_outputs attribute)
kp = ops.kron_plan(x, L, R, SIG, P)
timeit("ops.kron_plan(...).run 64x64 packed", lambda: kp.run(x))
rp = ops.rowquant_plan(x, SIG, FQ_OUT_PACKED | 0x20 | 0x400)
timeit("ops.rowquant_plan(...).run", lambda: rp.run(x))
ot.static_outputs = True
qz.static_outputs = True
timeit("OnlineTrans(matmul).forward, static_outputs", lambda: ot(x3))
timeit("Quantizer(lac).forward, static_outputs", lambda: qz(x))
lin.static_outputs = True
timeit("Linear4bit.forward (decode kernel), static_outputs", lambda: lin(p))
lin.static_outputs = False
ot.static_outputs = False
qz.static_outputs = False
timeit("torch.empty x2 (the output allocation alone)", lambda: (torch.empty((4, 2048), dtype=torch.uint8, device=dev), torch.empty((4,), dtype=torch.float16, device=dev)))
if "--profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20000):
        ops.kron_quant(x, L, R, SIG, P)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20000):
        ot(x3)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
