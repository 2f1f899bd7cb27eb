"""ctypes binding of libfqhip.so (C ABI declared in include/fqhip.h).

There is NO fallback: if the shared library is missing or fails to load, importing this module raises, and
every op in the package is unusable.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C flatquant_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os

# torch FIRST: the PyTorch-ROCm wheel bundles its own libamdhip64.so; loading libfqhip.so before it would bind
# /opt/rocm's copy under the same soname and the two runtimes would disagree about devices and streams.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FQHIP_LIB", os.path.join(_HERE, "lib", "libfqhip.so"))  # FQHIP_LIB: A/B builds
# FQHIP_OVERLAY (measurement only, tools/variants.sh): ':'-separated small shared objects, each ONE kernel file rebuilt with other -D
# switches. Loaded RTLD_GLOBAL in front of the library, their fq_launch_* definitions take the place of the library's own (its calls
# between translation units go through the PLT) — a 100 KB object per A/B variant instead of a 13 MB copy of the whole library.
OVERLAYS = [p for p in os.environ.get("FQHIP_OVERLAY", "").split(":") if p]

# flags (include/fqhip.h)
FQ_OUT_PACKED = 0x01
FQ_OUT_FAKEQUANT = 0x02
FQ_OUT_TRANSFORM = 0x04
FQ_ROUND_Y_F16 = 0x08
FQ_NO_CLAMP0 = 0x10
FQ_QUANT_F16 = 0x20
FQ_WS_PREPARED = 0x40
FQ_IN_RMSNORM = 0x80
FQ_IN_SILU_MUL = 0x100
FQ_GROUP128 = 0x200
FQ_SIG_F16 = 0x400
FQ_ASYM = 0x800
FQ_RATIO_POST = 0x10000
FQ_KV_LAC = 0x1
FQ_MAX_CLIPS = 4

FQ_OK, FQ_EINVAL, FQ_EUNSUPPORTED, FQ_ELAUNCH = 0, -1, -2, -3


class FqKronJob(ctypes.Structure):
    """include/fqhip.h: one job of fq_kron_quant_multi_{f16,bf16}"""
    _fields_ = [("x", ctypes.c_void_p), ("workspace", ctypes.c_void_p), ("q", ctypes.c_void_p), ("scale", ctypes.c_void_p),
                ("rows", ctypes.c_int64)]

# every symbol include/fqhip.h declares: (name, restype, argtypes)
_vp, _i64, _i, _f = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
_fp = ctypes.POINTER(ctypes.c_float)
_vpp = ctypes.POINTER(ctypes.c_void_p)
_ip = ctypes.POINTER(ctypes.c_int)
SYMBOLS = {
    "fq_kron_quant_f16": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _fp, _fp, _i, _i, _vpp, _vpp, _vpp, _vp, _vp, _i64,
                               _vp]),
    "fq_kron_quant_bf16": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _fp, _fp, _i, _i, _vpp, _vpp, _vpp, _vp, _vp, _i64,
                                _vp]),
    "fq_kron_quant_grouped_bf16": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "fq_kron_prepare_bf16": (_i, [_vp, _vp, _i, _i, _vp, _i64, _vp]),
    "fq_block_quant_bf16": (_i, [_vp, _vp, _i64, _i, _i, _i, _fp, _fp, _i, _i, _vpp, _vpp, _vpp, _vp, _vp]),
    "fq_rowquant_bf16": (_i, [_vp, _i64, _i, _fp, _fp, _i, _i, _vpp, _vpp, _vpp, _vp]),
    "fq_fakequant_bits_f16": (_i, [_vp, _i64, _i, _f, _f, _i, _i, _vp, _vp]),
    "fq_fakequant_bits_bf16": (_i, [_vp, _i64, _i, _f, _f, _i, _i, _vp, _vp]),
    "fq_kron_quant_grouped_mats_f16": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "fq_kron_quant_grouped_mats_bf16": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "fq_kron_quant_grouped_f16": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "fq_kron_quant_ex_f16": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _f, _fp, _fp, _i, _i, _vpp, _vpp, _vpp, _vp, _vp, _i64, _vp]),
    "fq_kron_workspace_bytes": (_i64, [_i, _i]),
    "fq_single_trans_f16": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "fq_single_trans_bf16": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "fq_kron_multi_table_bytes": (_i64, [_i]),
    "fq_kron_multi_prepare": (_i, [_vp, _i, _vp, _i64, _vp]),
    "fq_kron_quant_multi_f16": (_i, [_vp, _i, _i, _f, _f, _i, _vp]),
    "fq_kron_quant_multi_bf16": (_i, [_vp, _i, _i, _f, _f, _i, _vp]),
    "fq_kron_prepare_f16": (_i, [_vp, _vp, _i, _i, _vp, _i64, _vp]),
    "fq_rmsnorm_kron_quant_f16": (_i, [_vp, _f, _vp, _vp, _i64, _i, _i, _fp, _fp, _i, _i, _vpp, _vpp, _vpp, _vp, _vp]),
    "fq_rmsnorm_kron_quant_ws_f16": (_i, [_vp, _f, _vp, _vp, _i64, _i, _i, _fp, _fp, _i, _i, _vpp, _vpp, _vpp, _vp, _vp, _i64,
                                          _vp]),
    "fq_rmsnorm_f16": (_i, [_vp, _vp, _i64, _i, _f, _vp]),
    "fq_silu_mul_kron_quant_f16": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _fp, _fp, _i, _i, _vpp, _vpp, _vpp, _vp, _vp,
                                        _i64, _vp]),
    "fq_silu_mul_f16": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "fq_int4_frag_bytes": (_i64, [_i, _i]),
    "fq_int4_to_frag": (_i, [_vp, _i, _i, _vp, _vp]),
    "fq_int4_skinny_gemm_i32": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp]),
    "fq_int4_skinny_linear_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp]),
    "fq_int4_skinny_split_workspace_bytes": (_i64, [_i64, _i, _i]),
    "fq_int4_skinny_linear_split_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _i64, _vp]),
    "fq_int4_skinny_linear_multi_f16": (_i, [_i, _vpp, _vpp, _vpp, _vpp, _vpp, _i64, _ip, _i, _vpp, _vp]),
    "fq_kron64_linear_multi_f16": (_i, [_vp, _i, _f, _vp, _vp, _i64, _i, _fp, _fp, _i, _vpp, _vpp, _vpp, _ip, _vpp, _vp, _i64, _vp]),
    "fq_bf6_blob_bytes": (_i64, [_i64, _i]),
    "fq_int4_to_bf6": (_i, [_vp, _i64, _i, _i, _vp, _vp]),
    "fq_bf6_gemm_i32": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp]),
    "fq_bf6_linear_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp]),
    "fq_int4_linear_fp6_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _i64, _vp]),
    "fq_int4_linear_fp6_multi_f16": (_i, [_i, _vpp, _vpp, _vpp, _vpp, _vpp, _vpp, _i64, _ip, _i, _vpp, _vp, _i64, _vp]),
    "fq_kv_quant_f16": (_i, [_vp, _vp, _i64, _i, _f, _f, _i, _vp, _vp, _vp, _vp]),
    "fq_kv_dequant_f16": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp]),
    "fq_kv_append_i4": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "fq_kv_quant_append_i4": (_i, [_vp, _vp, _vp, _i64, _i, _i, _fp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "fq_kv_batch_decode_i4_ex": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "fq_kv_batch_decode_i4": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "fq_kv_append_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "fq_kv_batch_decode_f16_ex": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "fq_kv_batch_decode_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "fq_kv_decode_workspace_bytes": (_i64, [_i, _i, _i]),
    "fq_kv_batch_decode_split": (_i, [_i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp]),
    "fq_kv_decode_workspace_bytes_gqa": (_i64, [_i, _i, _i, _i]),
    "fq_kv_batch_decode_gqa": (_i, [_i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp]),
    "fq_kv_transform_image_bytes": (_i64, [_i]),
    "fq_kv_transform_image_f16": (_i, [_vp, _i, _vp, _vp]),
    "fq_kv_decode_append_i4": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp]),
    "fq_kv_batch_decode_copies": (_i, [_i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp]),
    "fq_silu_mul_hadamard_quant_f16": (_i, [_vp, _vp, _i64, _i, _i, _vp, _f, _f, _f, _vp, _vp, _vp]),
    "fq_block_quant_f16": (_i, [_vp, _vp, _i64, _i, _i, _i, _fp, _fp, _i, _i, _vpp, _vpp, _vpp, _vp, _vp]),
    "fq_int4_gemm_i32": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp]),
    "fq_int4_linear_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp]),
    "fq_hadamard_f16": (_i, [_vp, _vp, _i64, _i, _i, _vp, _f, _vp]),
    "fq_fwht_f32_f16": (_i, [_vp, _vp, _i64, _i, _f, _vp]),
    "fq_plan_kron": (_vp, [_i, _vp, _vp, _i64, _i, _i, _f, _f, _i, _vp, _i64]),
    "fq_plan_rowquant": (_vp, [_i, _i64, _i, _f, _f, _i]),
    "fq_plan_skinny_linear": (_vp, [_vp, _vp, _vp, _i64, _i, _i]),
    "fq_plan_run": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "fq_plan_free": (None, [_vp]),
    "fq_hadamard_quant_f16": (_i, [_vp, _i64, _i, _i, _vp, _f, _f, _f, _vp, _vp, _vp]),
    "fq_hadamard_quant_mfma_f16": (_i, [_vp, _i64, _i, _i, _vp, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "fq_silu_mul_hadamard_quant_mfma_f16": (_i, [_vp, _vp, _i64, _i, _i, _vp, _f, _f, _f, _vp, _vp, _vp]),
    "fq_hadamard_quantizer_mfma_f16": (_i, [_vp, _vp, _i64, _i, _i, _vp, _f, _f, _vp, _vp, _vp, _vp]),
    "fq_rowquant_f16": (_i, [_vp, _i64, _i, _fp, _fp, _i, _i, _vpp, _vpp, _vpp, _vp]),
    "fq_sym_quant_f16": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "fq_sym_dequant_i32_f16": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp]),
    "fq_last_error": (ctypes.c_char_p, []),
    "fq_version": (_i, []),
}


class FqError(RuntimeError):
    """A libfqhip entry point returned a negative code (mirrors torch::check* -> RuntimeError in the
    reference's bindings.cpp:11-16,29-34)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"libfqhip error {code}: {msg}")
        self.code = code


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `make -C flatquant_amd/csrc` "
            "(needs hipcc, --offload-arch=gfx950). flatquant_amd has no CPU or PyTorch fallback.")
    for ov in OVERLAYS:
        ctypes.CDLL(os.path.abspath(ov), mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(code: int) -> None:
    if code != FQ_OK:
        raise FqError(code, lib.fq_last_error().decode("utf-8", "replace"))
