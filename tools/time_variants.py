#!/usr/bin/env python3
"""A/B timing of the C2 launch: libfqhip.so and every variants/libfqhip_*.so (tools/variants.sh) loaded into ONE
process and timed round-robin (ROUNDS x 100 launches each), so clock / thermal drift hits every variant alike.
Prints median and min over rounds."""
import ctypes
import glob
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flatquant_amd._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED, LIB_PATH  # noqa: E402

ROUNDS = int(os.environ.get("ROUNDS", "20"))
ROWS = 16384
g = torch.Generator(device="cuda").manual_seed(0)
xs = [torch.randn(ROWS, 4096, generator=g, device="cuda", dtype=torch.float16) for _ in range(4)]
L = (torch.randn(64, 64, generator=g, device="cuda") / 8).half()
R = (torch.randn(64, 64, generator=g, device="cuda") / 8).half()
qs = [torch.empty(ROWS, 2048, dtype=torch.uint8, device="cuda") for _ in range(4)]
s = torch.empty(ROWS, dtype=torch.float16, device="cuda")
sig = 0.9820137619972229
smax, smin = (ctypes.c_float * 4)(sig), (ctypes.c_float * 4)(sig)
none4 = (ctypes.c_void_p * 4)()
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P4, F4 = ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_float)

libs = {"default": LIB_PATH, "default (2nd handle)": LIB_PATH + ".copy"}
import shutil
shutil.copyfile(LIB_PATH, LIB_PATH + ".copy")   # a second, independently loaded copy = the noise floor of this A/B
for path in sorted(glob.glob(os.path.join(ROOT, "variants", "libfqhip_*.so"))):
    libs[os.path.basename(path)[len("libfqhip_"):-3]] = path
fns = {}
for name, path in libs.items():
    fn = ctypes.CDLL(path).fq_kron_quant_f16
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, F4, F4, ctypes.c_int, ctypes.c_int,
                                          P4, P4, P4, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    fns[name] = fn


# MODE=packed (the C2 launch, default) | fq (fake-quant output, FlatQuantizedLinear's contract: 16 KB per token) |
#      fqy (transform + fake-quant outputs) | y (transform only)
MODE = os.environ.get("MODE", "packed")
fqs = [torch.empty(ROWS, 4096, dtype=torch.float16, device="cuda") for _ in range(4)] if MODE != "packed" else None
ys = [torch.empty(ROWS, 4096, dtype=torch.float16, device="cuda") for _ in range(4)] if MODE in ("fqy", "y") else None
if os.environ.get("SIGV"):
    sig = float(os.environ["SIGV"])
    smax, smin = (ctypes.c_float * 4)(sig), (ctypes.c_float * 4)(sig)


def launch(fn, i):
    qa, sa, fa = (ctypes.c_void_p * 4)(), (ctypes.c_void_p * 4)(), (ctypes.c_void_p * 4)()
    if MODE == "packed":
        qa[0], sa[0] = qs[i % 4].data_ptr(), s.data_ptr()
        rc = fn(xs[i % 4].data_ptr(), L.data_ptr(), R.data_ptr(), None, ROWS, 64, 64, smax, smin, 1,
                FQ_OUT_PACKED | FQ_NO_CLAMP0, qa, sa, none4, None, None, 0, sp)
    else:
        fa[0] = fqs[i % 4].data_ptr()
        flags = {"fq": 0x02 | 0x08, "fqy": 0x02 | 0x04 | 0x08, "y": 0x04}[MODE]
        rc = fn(xs[i % 4].data_ptr(), L.data_ptr(), R.data_ptr(), None, ROWS, 64, 64, smax, smin, 1,
                flags, none4, none4, fa, ys[i % 4].data_ptr() if ys else None, None, 0, sp)
    assert rc == 0, rc


for fn in fns.values():                     # warm-up: every variant, and the clocks
    for i in range(100):
        launch(fn, i)
torch.cuda.synchronize()
times = {name: [] for name in fns}
import random
random.seed(1)
order = list(fns.items())
for rnd in range(ROUNDS):
    random.shuffle(order)                   # no variant always runs right after the same neighbour
    for name, fn in order:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(100):
            launch(fn, i)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 100 * 1e3)
for name, t in times.items():
    print(f"{name:24s} median {statistics.median(t):6.2f}  min {min(t):6.2f}  max {max(t):6.2f} us")
