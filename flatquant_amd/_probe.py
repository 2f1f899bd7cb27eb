"""ctypes binding of libfqprobe.so (include/fqprobe.h): measurement / test infrastructure — the streaming floor bench.py
quotes and the one-instruction MFMA probe of tools/mfma_probe*.py. NOT part of the product: nothing under flatquant_amd/
imports this module except on request, and libfqhip.so does not contain these kernels."""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfqprobe.so")
_vp, _i64, _i = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
SYMBOLS = {
    "fq_probe_mfma_32x32x16_f16": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "fq_probe_stream_4096": (_i, [_vp, _i64, _vp, _vp, _i, _vp]),
}
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: run `make -C flatquant_amd/csrc`")
        _lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args
    return _lib


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def probe_mfma(A: torch.Tensor, B: torch.Tensor, C: torch.Tensor) -> torch.Tensor:
    """D = A[32,16] @ B[16,32] + C[32,32] by one v_mfma_f32_32x32x16_f16 (oracle-calibration helper)."""
    assert A.is_cuda and A.dtype == B.dtype == torch.float16 and C.dtype == torch.float32
    D = torch.empty((32, 32), dtype=torch.float32, device=A.device)
    with torch.cuda.device(A.device):
        rc = lib().fq_probe_mfma_32x32x16_f16(A.data_ptr(), B.data_ptr(), C.data_ptr(), D.data_ptr(), _stream(A))
    if rc != 0:
        raise RuntimeError(f"fq_probe_mfma_32x32x16_f16 failed ({rc})")
    return D


def probe_stream_4096(x: torch.Tensor, q: torch.Tensor, s: torch.Tensor, waves_per_simd: int = 4) -> None:
    """HBM-floor probe: move the bytes of the d = 4096 fused kernel with no arithmetic (measurement aid)."""
    assert x.is_cuda and x.dtype == torch.float16 and q.dtype == torch.uint8 and s.dtype == torch.float16
    rows = x.numel() // 4096
    with torch.cuda.device(x.device):
        rc = lib().fq_probe_stream_4096(x.data_ptr(), rows, q.data_ptr(), s.data_ptr(), waves_per_simd, _stream(x))
    if rc != 0:
        raise RuntimeError(f"fq_probe_stream_4096 failed ({rc})")
