#!/usr/bin/env python3
"""Phase stamps of fq_kron_duo_kernel (measurement build -DDUO_TRACE=<workgroup>): FQHIP_LIB=variants/libfqhip_dtrace.so
python tools/microbench/duo_trace.py.  Prints, per wave and iteration, the s_memtime deltas between the stamps."""
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
x = torch.randn(8192, 28672, generator=g, device=dev, dtype=torch.float16)
L = (torch.randn(128, 128, generator=g, device=dev) / 128 ** 0.5).half()
R = (torch.randn(224, 224, generator=g, device=dev) / 224 ** 0.5).half()
for _ in range(20):
    ops.kron_quant(x, L, R, [(0.98, 0.98)], FQ_OUT_PACKED | FQ_NO_CLAMP0)
torch.cuda.synchronize()
lib = ctypes.CDLL(LIB_PATH)
buf = np.zeros(8 * 32 * 12, dtype=np.uint64)
rc = lib.fq_duo_trace_read(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
t = buf.reshape(8, 32, 12).astype(np.int64)
names = ["meetA", "gemm1", "meetX", "dma", "gemm2", "extrema", "meetB", "quant", "wait"]
t0 = t[:, :, 0].min()
print("stamp units: s_memtime ticks (100 MHz); columns = time between consecutive stamps")
print("wave it  start " + " ".join(f"{n:>8s}" for n in names) + "    total   (stores: ticks inside `quant` spent issuing the token's four 16-byte stores)")
for w in (0, 1, 3, 4, 7):
    for it in range(4, 12):
        r = t[w, it]
        d = np.diff(r[:10])
        nxt = t[w, it + 1, 0] - r[0]
        print(f"{w:4d} {it:2d} {r[0] - t0:6d} " + " ".join(f"{v:8d}" for v in d) + f" {nxt:8d}   stores {r[10]:6d}")
rt = (t[0, 14, 11] - t[0, 4, 11]) / 100.0  # s_memrealtime: 100 MHz
print(f"s_memtime ticks per microsecond over wave 0's iterations 4..14: {(t[0, 14, 0] - t[0, 4, 0]) / rt:.0f} "
      f"({rt:.1f} us for 10 iterations = {rt / 10:.2f} us per token of a group)")
