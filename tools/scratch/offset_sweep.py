#!/usr/bin/env python3
"""Does the distance between the input stream and the output stream of the 64x64 launch matter (HBM channel / bank aliasing)?
One pool; x buffers at fixed places, output buffers displaced by DELTA bytes; MODE=fq|packed; LIBS = extra library paths."""
import ctypes, os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flatquant_amd._lib import LIB_PATH
MODE = os.environ.get("MODE", "fq")
ROWS = 16384
XB = ROWS * 8192
OB = XB if MODE == "fq" else ROWS * 2048
NB = 4
pool = torch.empty(NB * XB + NB * (OB + (64 << 20)) + (256 << 20), dtype=torch.uint8, device="cuda")
base = (pool.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20)
g = torch.Generator(device="cuda").manual_seed(0)
src = torch.randn(ROWS, 4096, generator=g, device="cuda", dtype=torch.float16)
xs = [base + i * XB for i in range(NB)]
for p in xs:
    ctypes.cdll.LoadLibrary  # noqa
for i in range(NB):
    off = xs[i] - pool.data_ptr()
    pool[off:off + XB].view(torch.float16).view(ROWS, 4096).copy_(src)
L = (torch.randn(64, 64, generator=g, device="cuda") / 8).half()
R = (torch.randn(64, 64, generator=g, device="cuda") / 8).half()
s = torch.empty(ROWS, dtype=torch.float16, device="cuda")
sig = 0.9820137619972229
smax, smin = (ctypes.c_float * 4)(sig), (ctypes.c_float * 4)(sig)
none4 = (ctypes.c_void_p * 4)()
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P4, F4 = ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_float)
libs = {"default": LIB_PATH}
for path in os.environ.get("LIBS", "").split(":"):
    if path:
        libs[os.path.basename(path)] = path
out_base = base + NB * XB
for name, path in libs.items():
    fn = ctypes.CDLL(path).fq_kron_quant_f16
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, F4, F4, ctypes.c_int, ctypes.c_int,
                                          P4, P4, P4, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    for delta in [0, 256, 1024, 4096, 8192, 16384, 65536, 1 << 20, (1 << 20) + 4096, (2 << 20) + 8192 + 256, 3 << 20, (16 << 20) + 12288, (32 << 20) + 4096 * 5]:
        stride = OB + (64 << 20) if delta else OB
        outs = [out_base + i * OB + delta for i in range(NB)] if delta == 0 else [out_base + i * (OB + (8 << 20)) + delta for i in range(NB)]

        def launch(i):
            a = (ctypes.c_void_p * 4)()
            a[0] = outs[i % NB]
            if MODE == "fq":
                rc = fn(xs[i % NB], L.data_ptr(), R.data_ptr(), None, ROWS, 64, 64, smax, smin, 1, 0x02 | 0x08, none4, none4, a, None, None, 0, sp)
            else:
                sa = (ctypes.c_void_p * 4)()
                sa[0] = s.data_ptr()
                rc = fn(xs[i % NB], L.data_ptr(), R.data_ptr(), None, ROWS, 64, 64, smax, smin, 1, 0x01 | 0x10, a, sa, none4, None, None, 0, sp)
            assert rc == 0
        for i in range(50):
            launch(i)
        torch.cuda.synchronize()
        ts = []
        for rnd in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(100):
                launch(i)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 10)
        print(f"{name:28s} {MODE} delta {delta:>10d} (x0 {xs[0] % (1 << 30):>10d}, out0 {outs[0] % (1 << 30):>10d}): median {statistics.median(ts):6.2f} us  min {min(ts):6.2f}")
