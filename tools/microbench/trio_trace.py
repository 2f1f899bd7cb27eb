#!/usr/bin/env python3
"""Phase stamps of fq_kron_trio_kernel (measurement build -DTRIO_TRACE=<workgroup>): FQHIP_LIB=variants/libfqhip_trace.so
python tools/microbench/trio_trace.py.  Prints, per wave and iteration, the s_memtime deltas between the stamps."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flatquant_amd import ops  # noqa: E402
from flatquant_amd._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED, LIB_PATH  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(16384, 14336, generator=g, device=dev, dtype=torch.float16)
L = (torch.randn(112, 112, generator=g, device=dev) / 112 ** 0.5).half()
R = (torch.randn(128, 128, generator=g, device=dev) / 128 ** 0.5).half()
for _ in range(20):
    ops.kron_quant(x, L, R, [(0.98, 0.98)], FQ_OUT_PACKED | FQ_NO_CLAMP0)
torch.cuda.synchronize()
lib = ctypes.CDLL(LIB_PATH)
buf = np.zeros(12 * 32 * 12, dtype=np.uint64)
rc = lib.fq_trio_trace_read(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
t = buf.reshape(12, 32, 12).astype(np.int64)
names = ["bar1", "A:gemm1", "bar2", "dma", "B:gemm2", "stats", "bar3", "C:quant", "wait", "stores"]
t0 = t[:, :, 0].min()
print("stamp units: s_memtime ticks; columns = time spent between consecutive stamps")
print("wave it  start " + " ".join(f"{n:>8s}" for n in names) + "    total")
for w in (0, 1, 4, 8):
    for it in range(4, 12):
        r = t[w, it]
        d = np.diff(r[:11])
        nxt = t[w, it + 1, 0] - r[0]
        print(f"{w:4d} {it:2d} {r[0] - t0:6d} " + " ".join(f"{v:8d}" for v in d) + f" {nxt:8d}")
rt = (t[0, 20, 11] - t[0, 4, 11]) / 100.0  # s_memrealtime: 100 MHz
print(f"s_memtime ticks per microsecond over wave 0's iterations 4..20: {(t[0, 20, 0] - t[0, 4, 0]) / rt:.0f} "
      f"({rt:.1f} us for 16 iterations = {rt / 16:.2f} us per token of a group)")
per = (t[0, 20, 0] - t[0, 4, 0]) / 16
print(f"ticks per iteration (wave 0, it 4..20): {per:.0f} = {per / 3:.0f} per interval")
