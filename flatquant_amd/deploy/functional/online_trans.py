"""Counterpart of deploy/functional/online_trans.py: the dispatch the deploy modules call."""
import torch

from ... import ops
from ..._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED, FQ_QUANT_F16, FQ_RATIO_POST, FQ_ROUND_Y_F16, FQ_SIG_F16
from ...flatquant.function_utils import get_decompose_dim  # noqa: F401
from ...flatquant.hadamard_utils import get_hadK, is_pow2, matmul_hadU_cuda  # noqa: F401
from .. import PackedQuantizedTensor


def _clip(v):
    return ops.host_scalar(v)  # no per-call device sync for the buffers the deploy modules keep on the GPU


def deploy_kron_flags(M: int, N: int) -> int:
    """Flags that reproduce the reference's Triton launch structure for a factor pair (deploy/kernels/kron_matmul.py:192-266):
    M <= 64: one kernel quantises the fp32 accumulator, extrema NOT clamped through zero (:91-107) -> FQ_NO_CLAMP0.
    M > 64 (`is_split`): the product is first stored as fp16 (:213-232) and `quant_kernel` (:133-189) quantises that tensor,
    loading it into a next_power_of_2(M) x next_power_of_2(N) tile whose padding is ZERO (:156-157) — so zero joins the extrema
    exactly when either size is not a power of two -> FQ_ROUND_Y_F16, and FQ_NO_CLAMP0 only for power-of-two pairs."""
    flags = FQ_OUT_PACKED
    if M <= 64:
        return flags | FQ_NO_CLAMP0
    flags |= FQ_ROUND_Y_F16
    if is_pow2(M) and is_pow2(N):
        flags |= FQ_NO_CLAMP0
    return flags


def kronecker_matmul(x, invs, clip_factor_a_max=1.0, clip_factor_a_min=1.0):
    """Transform + per-token INT4 quantisation, returns a PackedQuantizedTensor.

    Reference: deploy/functional/online_trans.py:113-141.
      len(invs) == 2: x [bsz, seq, d], invs = [left [M,M], right [N,N]] -> kron_matmul (Triton, 1-2 launches)
      len(invs) == 1: x [bsz, seq, head_dim, num_heads], invs = [P [H,H]] -> block_matmul (packed transposed)
    Here each is ONE HIP launch.  ``clip_factor_a_*`` are the raw (pre-sigmoid) factors, as in the reference.
    Statistics follow the Triton kernels pair by pair (deploy_kron_flags: fp32 accumulator and no clamp through zero for
    M <= 64; the fp16 round trip and the zero-padded tile of the split path for M > 64).
    """
    init_shape = x.shape
    sig = ops.sigmoid_pair(_clip(clip_factor_a_max), _clip(clip_factor_a_min))
    if len(invs) == 2:
        bsz, seq_len, hidden_dim = init_shape
        invL, invR = invs
        assert x.is_contiguous(), "Matrix B must be contiguous"
        o = ops.kron_quant(x, invL.contiguous(), invR.contiguous(), [sig], deploy_kron_flags(invL.shape[0], invR.shape[0]))
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
    else:
        # :106: (max|x| / 7).to(fp16) * ratio in the same launch (FQ_RATIO_POST); the scales keep x's leading shape as the
        # reference's `torch.max(..., dim=-1)[0].unsqueeze(1)` gives it
        o = ops.rowquant(x2.contiguous(), [(float(input_clip_ratio), 1.0)], FQ_OUT_PACKED | FQ_QUANT_F16 | FQ_RATIO_POST)
        return PackedQuantizedTensor(o.q[0].reshape(x.shape[:-1] + (x.shape[-1] // 2,)), o.scale[0].reshape(x.shape[:-1]).unsqueeze(1))
    return PackedQuantizedTensor(o.q[0].reshape(x.shape[:-1] + (x.shape[-1] // 2,)), o.scale[0].reshape(-1, 1))
