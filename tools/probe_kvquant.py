#!/usr/bin/env python3
"""Small-batch latency of fq_kv_quant_f16 (run under rocprofv3 --kernel-trace --stats): rows in {128, 4096}, with / without the K transform.
Each variant is launched 200 times in its own stretch, separated by a marker kernel count so the stats rows can be told apart by Calls."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flatquant_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
T = (torch.randn(128, 128, generator=g, device="cuda") / 128 ** 0.5).half()
for rows, n in ((128, 200), (4096, 300), (131072, 400)):
    x = torch.randn(rows, 128, generator=g, device="cuda").half()
    for _ in range(n):
        ops.kv_quant(x, T)
    torch.cuda.synchronize()
    for _ in range(n + 50):
        ops.kv_quant(x)
    torch.cuda.synchronize()
print("done")
