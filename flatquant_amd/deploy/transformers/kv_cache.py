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


def init_kv_i4(kv_data, kv_param, kv_indptr, kv_indices, last_page_offset, k, v, k_param, v_param, seqlen_indptr, layer_idx,
               group_size=1):
    """kv_cache.py:69-80 (-> _CUDA.init_kv_i4): append each request's tokens seqlen_indptr[b] .. seqlen_indptr[b+1] - 1.
    ``group_size`` (extension): see ops.kv_append."""
    ops.kv_append(kv_data, kv_param, kv_indptr, kv_indices, last_page_offset, k.contiguous(), v.contiguous(),
                  k_param.contiguous(), v_param.contiguous(), layer_idx, seqlen_indptr, group_size)


def append_kv_i4(kv_data, kv_param, kv_indptr, kv_indices, last_page_offset, k, v, k_param, v_param, layer_idx, group_size=1):
    """kv_cache.py:83-95 (-> _CUDA.append_kv_i4): one new token per request."""
    ops.kv_append(kv_data, kv_param, kv_indptr, kv_indices, last_page_offset, k.contiguous(), v.contiguous(),
                  k_param.contiguous(), v_param.contiguous(), layer_idx, None, group_size)


def batch_decode_i4(o, q, kv_data, kv_param, kv_indptr, kv_indices, last_page_offset, layer_idx):
    """kv_cache.py:98-105 (-> _CUDA.batch_decode_i4): writes the attention output into ``o``."""
    o.copy_(ops.kv_batch_decode(q.contiguous(), kv_data, kv_param, kv_indptr, kv_indices, last_page_offset, layer_idx))
    return o


class MultiLayerPagedKVCache4Bit:
    """The INT4 configuration of kv_cache.py:166-359 (``disable_quant=False``, ``trans`` "matmul" or "none") with the
    same constructor arguments, page / scale tensors and ``update`` contract: the first call per layer stores the
    (transformed, quantised) prompt keys / values and returns the fp16 key / value states for the prefill attention; every
    later call appends one token per request and returns a callable that runs the INT4 decode attention for the
    layer's query ([bsz, 1, heads, head_dim] -> the same shape). Requests have equal lengths (no attention mask), as
    the reference's own restriction to one page count per batch implies (:371-372). trans="had" (QuaRot: a normalised
    Hadamard rotation of keys and queries over head_dim, kv_cache.py:64-67) runs the register FWHT (fq_hadamard_f16)
    in front of the quantiser / of the decode attention."""

    def __init__(self, batch_size, page_size, max_seq_len, device, n_layers, num_heads, head_dim, disable_quant=False,
                 trans_dtype=torch.float16, trans="had", group_size=1):
        if disable_quant:
            raise NotImplementedError("flatquant_amd: the fp16 configuration of the paged cache is not built")
        self.page_size, self.batch_size, self.max_seq_len = page_size, batch_size, max_seq_len
        self.device, self.n_layers, self.trans, self.group_size = device, n_layers, trans, group_size
        self.org_head_dim = head_dim
        n_pages = self.page_cnt_from_length(max_seq_len) * batch_size
        self.pages = torch.empty((n_pages, n_layers, 2, num_heads, page_size, head_dim // 2), dtype=torch.uint8, device=device)
        self.scales = torch.empty((n_pages, n_layers, 2, num_heads, page_size, 2), dtype=torch.float16, device=device)
        self._needs_init = [True] * n_layers
        self.length = 0

    def page_cnt_from_length(self, length):
        return (length + self.page_size - 1) // self.page_size

    def _ensure_page_cnt_per_batch(self, expected):
        need = expected * self.batch_size
        have = self.pages.shape[0]
        if need <= have:
            return
        grow = max(need, have * 2) - have
        self.pages = torch.cat([self.pages, torch.empty((grow, *self.pages.shape[1:]), dtype=self.pages.dtype, device=self.device)])
        self.scales = torch.cat([self.scales, torch.empty((grow, *self.scales.shape[1:]), dtype=self.scales.dtype, device=self.device)])

    @property
    def seen_tokens(self):
        return self.length

    def get_seq_length(self, layer_idx=0):
        return self.length

    def get_cache_specs_for_flash_infer(self):
        """kv_cache.py:362-385 without an attention mask: page p of request b is page index p * batch_size + b."""
        page_cnt = self.page_cnt_from_length(self.length)
        ptr = self.length % self.page_size
        if self.length != 0 and ptr == 0:
            ptr = self.page_size
        dev = self.device
        return {
            "kv_data": self.pages,
            "kv_param": self.scales,
            "kv_indptr": torch.arange(0, self.batch_size + 1, device=dev, dtype=torch.int32) * page_cnt,
            "kv_indices": ((torch.arange(page_cnt, device=dev, dtype=torch.int32) * self.batch_size).unsqueeze(0)
                           + torch.arange(self.batch_size, device=dev, dtype=torch.int32).unsqueeze(1)).reshape(-1).contiguous(),
            "last_page_offset": torch.full((self.batch_size,), ptr, device=dev, dtype=torch.int32),
        }

    def update(self, key_states, value_states, layer_idx, cache_kwargs=None):
        cache_kwargs = cache_kwargs or {}
        if cache_kwargs.get("attention_mask") is not None:
            raise NotImplementedError("flatquant_amd: ragged batches (attention_mask) are not built")
        b, added, heads, hd = key_states.shape
        assert b == self.batch_size
        tk = cache_kwargs.get("trans_matrix_k") if self.trans.startswith("matmul") else None
        tk_inv_t = cache_kwargs.get("trans_matrix_k_inv_t") if self.trans.startswith("matmul") else None
        # :286-296 grouped-query attention: every query head gets its copy — made by the scatter (group_size below).
        # kv_cache.py:283-284: the cache's own calls leave lac off, so the clip factors play no part.
        if layer_idx == 0:
            self._ensure_page_cnt_per_batch(self.page_cnt_from_length(self.length + added))
            self.length += added
        if getattr(self, "_specs_len", None) != (self.length, self.pages.data_ptr()):   # index tensors: once per step, not per layer
            self._specs, self._specs_len = self.get_cache_specs_for_flash_infer(), (self.length, self.pages.data_ptr())
        specs = self._specs
        args = (specs["kv_data"], specs["kv_param"], specs["kv_indptr"], specs["kv_indices"], specs["last_page_offset"])
        tk16 = None if tk is None else tk.to(device=key_states.device, dtype=torch.float16).contiguous()
        had = self.trans == "had"
        if had:                                                         # :265-266 matmul_had_cuda on the keys
            key_states = ops.hadamard(key_states.to(torch.float16).contiguous())
        # K transform + K / V quantisation + append: one launch (fq_kv_quant_append_i4)
        ops.kv_quant_append(key_states.contiguous(), value_states.contiguous(), tk16, *args, layer_idx, self.group_size)
        if self._needs_init[layer_idx]:
            self._needs_init[layer_idx] = False
            keys = key_states if tk16 is None else torch.matmul(key_states.to(torch.float16), tk16)
            return keys, value_states                                   # :341-344: the un-quantised states for prefill
        assert added == 1

        def attend(q, transposed=False):
            """q [bsz, 1, heads, head_dim] -> attention output of the same shape; ``transposed`` (extension): as
            [bsz, 1, head_dim, heads], what o_proj_trans takes (saves the transpose + copy of modeling_llama.py:147-149)."""
            bq, q_len, n_q, d = q.shape
            assert q_len == 1
            q2 = q.reshape(bq, n_q, d).to(torch.float16)
            qt = None
            if had:                                                         # :134-138: the query side of the rotation
                q2 = ops.hadamard(q2.contiguous())
            elif tk_inv_t is not None:                                      # :139-140: ... of the learned K transform, in the launch
                qt = tk_inv_t.to(q.device, torch.float16).contiguous()
            return ops.kv_batch_decode(q2.contiguous(), *args, layer_idx, qt, transposed).unsqueeze(1)
        return attend
