"""Mirror of the reference's ``deploy`` package façade (deploy/__init__.py:37-71) over libfqhip."""
import torch

from .. import ops


class PackedQuantizedTensor:
    """Packed INT4 activation + per-token fp16 scales.  Reference: deploy/__init__.py:55-71."""

    def __init__(self, quantized_x: torch.Tensor, scales_x: torch.Tensor):
        self.quantized_x = quantized_x
        self.scales_x = scales_x

    def size(self):
        return self.quantized_x.size()

    @property
    def device(self):
        return self.quantized_x.device

    @property
    def dtype(self):
        return self.quantized_x.dtype


class LazyPackedQuantizedTensor(PackedQuantizedTensor):
    """(round 6) What a fused OnlineTrans group hands out when the projections behind it can run the transform as their launch's
    PROLOGUE (deploy/fuse.py, decode-sized calls under capture): the packed bytes are only produced if somebody other than those
    projections asks for them (``.quantized_x`` / ``.scales_x`` then run the group's transform launch, once for all members)."""

    def __init__(self, group, index, shape):
        self._group, self._index, self._shape = group, index, shape

    def _real(self):
        return self._group.materialise()[self._index]

    @property
    def quantized_x(self):
        return self._real().quantized_x

    @quantized_x.setter
    def quantized_x(self, v):      # (a caller that replaces the bytes owns them from then on)
        r = self._real()
        self.__class__ = PackedQuantizedTensor
        self.__dict__.clear()
        self.quantized_x, self.scales_x = v, r.scales_x

    @property
    def scales_x(self):
        return self._real().scales_x

    def size(self):
        return torch.Size(self._shape)


def flatten_last_dim_and_return_shape(x: torch.Tensor):
    shape_excl_last = x.shape[:-1]
    return x.view(-1, x.shape[-1]), shape_excl_last


def sym_quant(x, scale):
    """deploy/__init__.py:43-46 -> _CUDA.sym_quant (quant.cu:13-63): fp16 in, packed uint8 out."""
    assert x.dtype == scale.dtype == torch.float16
    x, x_shape_excl_last = flatten_last_dim_and_return_shape(x)
    return ops.sym_quant(x, scale.view(-1)).view(*x_shape_excl_last, -1)


def sym_dequant(q, scale_row, scale_col, bits=32):
    """deploy/__init__.py:48-52 -> _CUDA.sym_dequant (quant.cu:66-101)."""
    assert q.dtype == torch.int32
    assert scale_row.dtype == scale_col.dtype == torch.float16
    if bits != 32:
        raise RuntimeError("Unsupported data type")          # bindings.cpp:82
    q, q_shape_excl_last = flatten_last_dim_and_return_shape(q)
    return ops.sym_dequant(q, scale_row.view(-1), scale_col.reshape(-1)).view(*q_shape_excl_last, -1)


def matmul(A, B):
    """INT4 x INT4 -> INT32 GEMM on packed nibbles (deploy/__init__.py:37-41 -> gemm.cu, CUTLASS)."""
    assert A.shape[-1] % 32 == 0, "A.shape[-1]: {} must be multiplication of 32".format(A.shape[-1])
    A, A_shape_excl_last = flatten_last_dim_and_return_shape(A)
    B, B_shape_excl_last = flatten_last_dim_and_return_shape(B)
    return ops.int4_matmul(A.contiguous(), B.contiguous()).view(*A_shape_excl_last, *B_shape_excl_last)


from . import functional, nn  # noqa: E402,F401
from .fuse import fuse, unfuse  # noqa: E402,F401
from .graphed import GraphedDecode  # noqa: E402,F401

__all__ = ["matmul", "sym_quant", "sym_dequant", "PackedQuantizedTensor", "nn", "functional", "fuse", "unfuse", "GraphedDecode"]
