"""Counterpart of deploy/functional/online_trans.py: the dispatch the deploy modules call."""
import torch

from ... import ops
from ..._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED, FQ_QUANT_F16, FQ_SIG_F16
from ...flatquant.function_utils import get_decompose_dim  # noqa: F401
from ...flatquant.hadamard_utils import get_hadK, is_pow2, matmul_hadU_cuda  # noqa: F401
from .. import PackedQuantizedTensor


def _clip(v):
    return ops.host_scalar(v)  # no per-call device sync for the buffers the deploy modules keep on the GPU


def kronecker_matmul(x, invs, clip_factor_a_max=1.0, clip_factor_a_min=1.0):
    """Transform + per-token INT4 quantisation, returns a PackedQuantizedTensor.

    Reference: deploy/functional/online_trans.py:113-141.
      len(invs) == 2: x [bsz, seq, d], invs = [left [M,M], right [N,N]] -> kron_matmul (Triton, 1-2 launches)
      len(invs) == 1: x [bsz, seq, head_dim, num_heads], invs = [P [H,H]] -> block_matmul (packed transposed)
    Here each is ONE HIP launch.  ``clip_factor_a_*`` are the raw (pre-sigmoid) factors, as in the reference.
    Statistics follow the Triton kernels (no clamp of the extrema to zero: FQ_NO_CLAMP0).
    """
    init_shape = x.shape
    sig = ops.sigmoid_pair(_clip(clip_factor_a_max), _clip(clip_factor_a_min))
    if len(invs) == 2:
        bsz, seq_len, hidden_dim = init_shape
        invL, invR = invs
        assert x.is_contiguous(), "Matrix B must be contiguous"
        o = ops.kron_quant(x, invL.contiguous(), invR.contiguous(), [sig], FQ_OUT_PACKED | FQ_NO_CLAMP0)
        return PackedQuantizedTensor(o.q[0].reshape(bsz, seq_len, -1), o.scale[0].reshape(bsz, 1, seq_len))
    if len(invs) == 1:
        bsz, seq_len, head_dim, num_heads = init_shape
        assert x.is_contiguous(), "Matrix B must be contiguous"
        o = ops.block_quant(x, invs[0].contiguous(), [sig], FQ_OUT_PACKED | FQ_NO_CLAMP0, transpose_out=True)
        return PackedQuantizedTensor(o.q[0].reshape(bsz, seq_len, -1, num_heads),
                                     o.scale[0].reshape(bsz, 1, seq_len))
    raise NotImplementedError


def quant(x, clip_factor_a_max=1.0, clip_factor_a_min=1.0, input_clip_ratio=1.0):
    """deploy/functional/online_trans.py:90-110: per-token fp16 scales then sym_quant, fused in one launch."""
    cmax, cmin = _clip(clip_factor_a_max), _clip(clip_factor_a_min)
    x2 = x.reshape(-1, x.shape[-1])
    if cmax != 1.0:
        # :91-104: fp16 extrema x a 0-dim fp32 sigmoid tensor is an fp16 product under torch's promotion -> FQ_SIG_F16
        o = ops.rowquant(x2.contiguous(), [ops.sigmoid_pair_f16(cmax, cmin)], FQ_OUT_PACKED | FQ_QUANT_F16 | FQ_SIG_F16)
    elif input_clip_ratio != 1.0:
        # :106: (max|x| / 7).to(fp16) * ratio keeps the reference's op order in torch (x itself, not the flattened view:
        # the scales keep x's leading shape); the pack is the kernel behind deploy.sym_quant
        from .. import sym_quant
        scales = (torch.max(torch.abs(x), dim=-1)[0].unsqueeze(1) / 7).to(torch.float16) * input_clip_ratio
        return PackedQuantizedTensor(sym_quant(x, scales), scales)
    else:
        o = ops.rowquant(x2.contiguous(), [(1.0, 1.0)], FQ_OUT_PACKED | FQ_QUANT_F16)
    return PackedQuantizedTensor(o.q[0], o.scale[0].reshape(-1, 1))
