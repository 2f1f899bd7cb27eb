"""Host-side INT4 helpers (torch, any device) — counterpart of deploy/functional/quantization.py.
Used for weights and by tests; the activation path packs inside the HIP kernels."""
import torch


def two_compl(x, bits: int):
    return torch.where(x < 0, 2 ** bits + x, x)


def get_minq_maxq(bits: int, sym: bool):
    if sym:
        maxq = torch.tensor(2 ** (bits - 1) - 1)
        minq = -maxq - 1
    else:
        maxq = torch.tensor(2 ** bits - 1)
        minq = torch.tensor(0)
    return minq, maxq


def sym_quant(x, scale, maxq):
    scale = scale.to(x.device)
    return torch.clamp(torch.round(x / scale), -(maxq + 1), maxq), scale


def sym_dequant(q, scale):
    return scale * q


def sym_quant_dequant(x, scale, maxq):
    return sym_dequant(*sym_quant(x, scale, maxq))


def asym_quant(x, scale, zero, maxq):
    scale, zero = scale.to(x.device), zero.to(x.device)
    return torch.clamp(torch.round(x / scale) + zero, 0, maxq), scale, zero


def asym_dequant(q, scale, zero):
    return scale * (q - zero)


def asym_quant_dequant(x, scale, zero, maxq):
    return asym_dequant(*asym_quant(x, scale, zero, maxq))


def pack_i4(q):
    """Two signed 4-bit values per byte, even column in the low nibble (quantization.py:49-56)."""
    assert torch.is_signed(q), "The tensor to be packed should be signed int"
    assert torch.all(torch.logical_and(q >= -8, q <= 7))
    u = two_compl(q.to(dtype=torch.int8), 4).to(torch.uint8)
    return u[..., 0::2] | (u[..., 1::2] << 4)


def unpack_i4(x: torch.Tensor):
    """Inverse of pack_i4 -> int32 (quantization.py:60-82)."""
    assert x.dtype == torch.uint8, "The tensor to be unpacked should be stored in uint8"
    lo = (x & 0x0F).to(torch.int32)
    hi = ((x & 0xF0) >> 4).to(torch.int32)
    lo = torch.where(lo >= 8, lo - 16, lo)
    hi = torch.where(hi >= 8, hi - 16, hi)
    return torch.stack((lo, hi), dim=-1).reshape(*x.shape[:-1], x.shape[-1] * 2)
