"""Counterpart of the quantisation half of deploy/transformers/kv_cache.py.

What MultiLayerPagedKVCache4Bit.update does to the new keys / values before it hands them to the paged cache
(kv_cache.py:262-297): the K transform, asymmetric INT4 quantisation + packing, and the (scale, zero) parameter
tensors — here one launch per tensor (fq_kv_quant_f16) instead of a GEMM plus ~15 element-wise launches. The paged
cache itself and the INT4 decode attention (init_kv_i4 / append_kv_i4 / batch_decode_i4, kernels/flashinfer.cu) are
outside this round's scope; these functions produce exactly the tensors those entry points take.
"""
import torch

from ... import ops


def _clip(v) -> float:
    return float(torch.as_tensor(ops.host_scalar(v), dtype=torch.float16))   # the fp16 value torch multiplies with


def asym_quantize_and_pack_i4(x: torch.Tensor, clip_factor_a_max=1.0, clip_factor_a_min=1.0, lac: bool = False,
                              quantize: bool = True):
    """kv_cache.py:11-51. x fp16 [..., head_dim]; the clip factors are the already sigmoid-ed ones (:278-279) and
    only matter with ``lac``. -> (q uint8 [..., head_dim/2], scale fp16 [..., 1], zero fp16 [..., 1]); with
    ``quantize=False`` the first element is ``scale * (q - zero)`` instead (:37-38, :45-46)."""
    q, param = ops.kv_quant(x.contiguous(), None, (_clip(clip_factor_a_max), _clip(clip_factor_a_min)), lac)
    scale, zero = param[..., 0:1], param[..., 1:2]
    if not quantize:
        return ops.kv_dequant(q, param, lac=True), scale, zero   # scale * (q - zero) is the lac de-quantiser's formula
    return q, scale, zero


def unpack_i4_and_asym_dequantize(q: torch.Tensor, scale: torch.Tensor, zero: torch.Tensor, lac: bool = False):
    """kv_cache.py:54-61."""
    param = torch.cat([scale.reshape(*q.shape[:-1], 1), zero.reshape(*q.shape[:-1], 1)], dim=-1).to(torch.float16)
    return ops.kv_dequant(q.contiguous(), param.contiguous(), lac)


def transform_quantize_kv(key_states: torch.Tensor, value_states: torch.Tensor, trans_matrix_k=None,
                          kclip=(1.0, 1.0), vclip=(1.0, 1.0), lac: bool = False):
    """kv_cache.py:262-297 for [bsz, added_length, num_kv_heads, head_dim] fp16 keys / values:
    keys @ trans_matrix_k (when given) -> asym INT4 + pack; values -> asym INT4 + pack.
    -> (key_q, k_param, value_q, v_param): uint8 [bsz, len, heads, head_dim/2] and fp16 [bsz*len, heads, 2] = (scale, zero),
    the ``k`` / ``k_param`` / ``v`` / ``v_param`` arguments of init_kv_i4 / append_kv_i4 (before any GQA repeat)."""
    b, n, heads, hd = key_states.shape
    t = None if trans_matrix_k is None else trans_matrix_k.to(device=key_states.device, dtype=torch.float16).contiguous()
    kq, kp = ops.kv_quant(key_states.contiguous(), t, (_clip(kclip[0]), _clip(kclip[1])), lac)
    vq, vp = ops.kv_quant(value_states.contiguous(), None, (_clip(vclip[0]), _clip(vclip[1])), lac)
    return kq, kp.reshape(b * n, heads, 2), vq, vp.reshape(b * n, heads, 2)
