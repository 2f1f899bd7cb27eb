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


# The fp16 configuration's entry points (kv_cache.py:107-137 -> _CUDA.init_kv_f16 / append_kv_f16 / batch_decode_f16): the same
# calls on fp16 pages [pages, layers, 2, heads, page_size, head_dim]; ops.kv_append / ops.kv_batch_decode pick the kernel by the
# page dtype.
def init_kv_f16(kv_data, kv_param, kv_indptr, kv_indices, last_page_offset, k, v, k_param, v_param, seqlen_indptr, layer_idx,
                group_size=1):
    init_kv_i4(kv_data, kv_param, kv_indptr, kv_indices, last_page_offset, k, v, k_param, v_param, seqlen_indptr, layer_idx, group_size)


def append_kv_f16(kv_data, kv_param, kv_indptr, kv_indices, last_page_offset, k, v, k_param, v_param, layer_idx, group_size=1):
    append_kv_i4(kv_data, kv_param, kv_indptr, kv_indices, last_page_offset, k, v, k_param, v_param, layer_idx, group_size)


def batch_decode_f16(o, q, kv_data, kv_param, kv_indptr, kv_indices, last_page_offset, layer_idx):
    return batch_decode_i4(o, q, kv_data, kv_param, kv_indptr, kv_indices, last_page_offset, layer_idx)


def n_q_is_cache_heads(kv_data, src_heads, group):
    """the pages hold src_heads x group heads (the replicated layout: one cache head per query head)"""
    return kv_data.shape[3] == src_heads * group


# fuse_append: up to this many workgroups — (request, cache head) pairs, or (request, KV head) pairs where one workgroup serves a group — the decode launch quantises and appends the step's own row (every workgroup of a
# pair's last row runs the quantiser prologue: ~1.5 us each). Measured on a Llama-3-8B layer's captured step (profiles/r06_fused_append.txt):
# 66.5 -> 61.6 us at one request, 105 -> 99.5 at 16 (512 pairs), 162.6 -> 157.7 at 64 requests on the shared cache (512 pairs); level at 1024 and
# 2048 pairs; 368 -> 373-379 at 4096 pairs (the prologue in front of every one of eight rounds of workgroups costs more than the launch it saves).
FUSE_APPEND_MAX_PAIRS = 1024


class MultiLayerPagedKVCache4Bit:
    """kv_cache.py:166-392 with the same constructor arguments, page / scale tensors and ``update`` contract: the first call
    per layer stores the (transformed, quantised) prompt keys / values and returns the fp16 key / value states for the prefill
    attention; every later call appends one token per request and returns a callable that runs the decode attention for the
    layer's query ([bsz, 1, heads, head_dim] -> the same shape).

    * ``disable_quant=False``: the INT4 cache (uint8 pages of head_dim / 2 bytes per row + (scale, zero) per row);
      ``disable_quant=True``: the fp16 configuration (:177-190, init_kv_f16 / append_kv_f16 / batch_decode_f16): fp16 pages,
      the scale tensor is kept and filled with (1, 0) like the reference's, the prefill gets the UN-transformed states back
      (:262-263 are assigned before the transform and only the INT4 branch re-assigns them, :280-281).
    * ``trans`` "matmul*" (the learned K transform, with its inverse-transpose on the query side), "had" (QuaRot: a normalised
      Hadamard rotation of keys and queries over head_dim, :64-67, here the register FWHT) or anything else (none).
    * ``cache_kwargs["attention_mask"]`` [bsz, seq] of 0 / 1 (ragged prompts, :315-326,362-372): the prompt tokens whose mask
      is 1 are compacted request by request into the cache; request lengths are the mask's row sums from then on (the mask
      grows by one column per decode step). As in the reference, all requests must need the same number of pages
      (NotImplementedError otherwise, :371-372)."""

    def __init__(self, batch_size, page_size, max_seq_len, device, n_layers, num_heads, head_dim, disable_quant=False,
                 trans_dtype=torch.float16, trans="had", group_size=1, share_kv_heads=False, fuse_append=True,
                 read_one_copy=True):
        """``share_kv_heads`` (extension, round 6): with grouped-query attention (``group_size`` > 1) the reference's cache holds one copy of
        every KV head per QUERY head (:286-296). With this flag the pages hold the ``num_heads // group_size`` KV heads once and the decode
        launch maps query head h to cache head h // group_size (fq_kv_batch_decode_gqa): the same attention output bit for bit, 1 / group_size
        of the cache memory and of the bytes a decode step reads (a Llama-3-8B layer's step at 64 requests x 2048 tokens:
        profiles/r06_gqa_cache.txt). The page layout then differs from the reference's in its head count, nothing else.
        ``read_one_copy`` (round 6, third session): on the reference's replicated layout (``group_size`` > 1, not shared) the decode launch reads
        ONE of a KV head's ``group_size`` identical copies for the whole group of query heads (ops.kv_batch_decode(kv_copies=...)): every copy
        is still written as the reference writes it, the values read are the same (the output bit-identical below 32 (request, KV head) pairs, up to
        the order of fp32 additions above), a decode step reads 1 / group_size of the rows. Set it False
        if something other than this class's ``update`` fills the pages with copies that differ.
        ``fuse_append`` (round 6, third session): a decode step's K transform + K / V quantisation + append run INSIDE the decode-attention launch
        of the closure ``update`` returns (ops.kv_decode_append, fq_kv_decode_append_i4: same cache bytes, same output, one launch instead of
        two) where the geometry allows (INT4 pages, head_dim 128, page_size % 16 == 0). The rows reach the cache when the closure is called;
        if it is not called before the cache is touched again, they are appended by the usual launch then (``_flush_pending``)."""
        self.page_size, self.batch_size, self.max_seq_len = page_size, batch_size, max_seq_len
        self.device, self.n_layers, self.trans, self.group_size = device, n_layers, trans, group_size
        self.share_kv_heads = bool(share_kv_heads) and group_size > 1
        if self.share_kv_heads:
            assert num_heads % group_size == 0
            num_heads = num_heads // group_size          # heads the pages hold
            self.group_size = 1                          # ... and the append scatters every source head to exactly one cache head
        self.disable_quant = disable_quant
        self.org_head_dim = head_dim
        n_pages = self.page_cnt_from_length(max_seq_len) * batch_size
        self._pages = torch.empty((n_pages, n_layers, 2, num_heads, page_size, head_dim if disable_quant else head_dim // 2),
                                 dtype=torch.float16 if disable_quant else torch.uint8, device=device)
        self._scales = torch.empty((n_pages, n_layers, 2, num_heads, page_size, 2), dtype=torch.float16, device=device)
        self._needs_init = [True] * n_layers
        self.fuse_append = bool(fuse_append)
        self.read_one_copy = bool(read_one_copy)
        self._pending = None    # a decode step's (k, v, transform, index tensors, layer) not yet appended: the closure's launch will (fuse_append)
        self.length = 0
        self.generation = 0     # bumped when the page / index storage moves: graphs captured over the old pointers are stale (deploy.graphed)
        self._log = None        # deploy.graphed: the (layer_idx, added) host steps of the calls made while it is a list
        self._skip_host = False  # deploy.graphed: update() leaves the host step to replay_host() (capture pass)

    # The page / scale tensors as the reference names them. Reading them from outside first appends a decode step's rows that were left to
    # the closure's launch (fuse_append): whoever looks at the cache sees every token update() was given, as in the reference.
    @property
    def pages(self):
        self._flush_pending()
        return self._pages

    @pages.setter
    def pages(self, t):
        self._pages = t

    @property
    def scales(self):
        self._flush_pending()
        return self._scales

    @scales.setter
    def scales(self, t):
        self._scales = t

    def page_cnt_from_length(self, length):
        return (length + self.page_size - 1) // self.page_size

    def _ensure_page_cnt_per_batch(self, expected):
        need = expected * self.batch_size
        have = self._pages.shape[0]
        if need <= have:
            return
        grow = max(need, have * 2) - have
        self._pages = torch.cat([self._pages, torch.empty((grow, *self._pages.shape[1:]), dtype=self._pages.dtype, device=self.device)])
        self._scales = torch.cat([self._scales, torch.empty((grow, *self._scales.shape[1:]), dtype=self._scales.dtype, device=self.device)])
        self.generation += 1

    def would_grow(self, added: int) -> bool:
        """Would appending ``added`` tokens per request re-allocate the pages (and so move every pointer a captured graph holds)?"""
        return self.page_cnt_from_length(self.length + added) * self.batch_size > self._pages.shape[0]

    @property
    def seen_tokens(self):
        return self.length

    def get_seq_length(self, layer_idx=0):
        return self.length

    def get_cache_specs_for_flash_infer(self, attention_mask=None):
        """kv_cache.py:362-385: page p of request b is page index p * batch_size + b; without a mask every request has
        ``self.length`` tokens, with one its row sum."""
        dev = self.device
        if attention_mask is None:
            page_cnt = self.page_cnt_from_length(self.length)
            ptr = self.length % self.page_size
            if self.length != 0 and ptr == 0:
                ptr = self.page_size
            last = torch.full((self.batch_size,), ptr, device=dev, dtype=torch.int32)
        else:
            seqlens = attention_mask.to(dev).sum(dim=-1, dtype=torch.int32)
            cnt = self.page_cnt_from_length(seqlens)
            if bool((cnt[0] != cnt).any()):                                            # (:371-372, a host read as there)
                raise NotImplementedError("Current implementation does not support the case where batches have different number of pages")
            page_cnt = int(cnt[0])
            last = seqlens % self.page_size
            last = torch.where((seqlens != 0) & (last == 0), torch.full_like(last, self.page_size), last).to(torch.int32).contiguous()
        return {
            "kv_data": self._pages,
            "kv_param": self._scales,
            "kv_indptr": torch.arange(0, self.batch_size + 1, device=dev, dtype=torch.int32) * page_cnt,
            "kv_indices": ((torch.arange(page_cnt, device=dev, dtype=torch.int32) * self.batch_size).unsqueeze(0)
                           + torch.arange(self.batch_size, device=dev, dtype=torch.int32).unsqueeze(1)).reshape(-1).contiguous(),
            "last_page_offset": last,
        }

    def _static_specs(self):
        """The index tensors of the no-mask case (every request ``self.length`` tokens long) as STATIC device buffers rewritten in place:
        kv_indptr / kv_indices only when the page count changes, last_page_offset by one fill per step — same values as
        get_cache_specs_for_flash_infer(None) (tests/test_gpu_kvcache.py), no arange / full / reshape launches per step, and addresses
        a captured decode step can keep (deploy.graphed)."""
        bsz, dev = self.batch_size, self.device
        st = self.__dict__.get("_st")
        if st is None or st["pages_ptr"] != self._pages.data_ptr():
            st = {"pages_ptr": self._pages.data_ptr(), "page_cnt": -1, "ptr": -1,
                  "indptr": torch.zeros(bsz + 1, dtype=torch.int32, device=dev),
                  "indices": torch.zeros(max(1, self._pages.shape[0]), dtype=torch.int32, device=dev),
                  "last": torch.zeros(bsz, dtype=torch.int32, device=dev)}
            self._st = st
            self.generation += 1
        page_cnt = self.page_cnt_from_length(self.length)
        ptr = self.length % self.page_size
        if self.length != 0 and ptr == 0:
            ptr = self.page_size
        if st["page_cnt"] != page_cnt:
            st["indptr"].copy_(torch.arange(0, bsz + 1, device=dev, dtype=torch.int32) * page_cnt)
            st["indices"][:bsz * page_cnt].copy_(((torch.arange(page_cnt, device=dev, dtype=torch.int32) * bsz).unsqueeze(0)
                                                  + torch.arange(bsz, device=dev, dtype=torch.int32).unsqueeze(1)).reshape(-1))
            st["page_cnt"] = page_cnt
        if st["ptr"] != ptr:
            st["last"].fill_(ptr)
            st["ptr"] = ptr
        return {"kv_data": self._pages, "kv_param": self._scales, "kv_indptr": st["indptr"],
                "kv_indices": st["indices"][:bsz * page_cnt], "last_page_offset": st["last"]}

    def _host_step(self, layer_idx, added, mask=None):
        """The host side of one ``update``: layer 0 makes room and advances the length; the index tensors follow the length."""
        if layer_idx == 0:
            self._ensure_page_cnt_per_batch(self.page_cnt_from_length(self.length + added))
            self.length += added
        key = (self.length, self._pages.data_ptr(), None if mask is None else (mask.data_ptr(), ops.ver(mask), tuple(mask.shape)))
        if getattr(self, "_specs_key", None) != key:   # index tensors: once per step, not per layer
            self._specs = self._static_specs() if mask is None else self.get_cache_specs_for_flash_infer(mask)
            self._specs_key = key

    def replay_host(self, log):
        """deploy.graphed: apply the host steps ``log`` recorded ([(layer_idx, added), ...]) — what the update() calls inside a captured
        decode step did on the host — before the graph that holds their launches is replayed."""
        for layer_idx, added in log:
            self._host_step(layer_idx, added)

    def _as_f16(self, t):
        """``t`` as a contiguous fp16 tensor on the cache's device — itself when it already is one; otherwise ONE converted copy per source tensor
        (keyed by its storage and version): a model that keeps trans_matrix_k in fp32 (modeling_llama.py:185-186 registers fp32 buffers) would
        otherwise pay a conversion launch per layer and step, and the fragment image of the fused append (ops.kv_transform_image, keyed by the
        tensor it is given) would be rebuilt every step."""
        if t is None:
            return None
        dev = torch.device(self.device)
        if t.dtype == torch.float16 and t.is_contiguous() and t.device.type == dev.type and (dev.index is None or t.device.index == dev.index):
            return t
        memo = self.__dict__.setdefault("_f16_memo", {})
        key = (t.data_ptr(), ops.ver(t), t.dtype, tuple(t.shape), ops.cache_epoch())     # (ops.invalidate_caches() after a write through .data)
        ent = memo.get(key)
        if ent is None:
            if len(memo) >= 4 * max(1, self.n_layers):
                memo.clear()
            ent = memo[key] = (t.to(device=self.device, dtype=torch.float16).contiguous(), t)    # (the source is kept alive: its address cannot be recycled under the key)
        return ent[0]

    def _flush_pending(self):
        """Append the rows a decode step left to its closure's launch if that closure was never called (the two-launch form's first launch)."""
        pend, self._pending = self._pending, None
        if pend is not None:
            k, v, tk16, args, layer_idx, group = pend
            ops.kv_quant_append(k, v, tk16, *args, layer_idx, group)

    def update(self, key_states, value_states, layer_idx, cache_kwargs=None):
        self._flush_pending()
        cache_kwargs = cache_kwargs or {}
        mask = cache_kwargs.get("attention_mask")
        b, added, heads, hd = key_states.shape
        assert b == self.batch_size
        tk = cache_kwargs.get("trans_matrix_k") if self.trans.startswith("matmul") else None
        tk_inv_t = cache_kwargs.get("trans_matrix_k_inv_t") if self.trans.startswith("matmul") else None
        # :286-296 grouped-query attention: every query head gets its copy — made by the scatter (group_size below).
        # kv_cache.py:283-284: the cache's own calls leave lac off, so the clip factors play no part.
        if self._log is not None:
            self._log.append((layer_idx, added) if (mask is None and not self._needs_init[layer_idx]) else None)   # None: not replayable
        if not self._skip_host:
            self._host_step(layer_idx, added, mask)
        specs = self._specs
        args = (specs["kv_data"], specs["kv_param"], specs["kv_indptr"], specs["kv_indices"], specs["last_page_offset"])
        tk16 = self._as_f16(tk)
        had = self.trans == "had"
        orig_k, orig_v = key_states, value_states
        init = self._needs_init[layer_idx]
        if had:                                                         # :265-266 matmul_had_cuda on the keys
            key_states = ops.hadamard(key_states.to(torch.float16).contiguous())
        ragged_init = init and mask is not None
        # the reference's replicated layout: group_size identical copies per KV head, one of them read (read_one_copy)
        copies = self.group_size if (self.read_one_copy and 1 < self.group_size <= 8 and n_q_is_cache_heads(specs["kv_data"], heads, self.group_size)) else 1
        wgs = b * specs["kv_data"].shape[3]                 # workgroups of the decode launch: one per (request, cache head) ...
        if copies in (2, 4, 8) and hd == 128 and wgs // copies >= 32:     # (KV_MERGE_MIN_PAIRS of fq_kvcache.hip)
            wgs //= min(copies, 4)                          # ... or per (request, KV head) where one workgroup serves the group — up to four heads (fq_kv_decode_wg_heads)
        fused = (self.fuse_append and not init and added == 1 and not self.disable_quant and key_states.dtype == torch.float16
                 and wgs <= FUSE_APPEND_MAX_PAIRS
                 and value_states.dtype == torch.float16 and ops.kv_decode_append_supported(specs["kv_data"], heads))
        if fused:
            # the closure's launch quantises and appends these rows itself (ops.kv_decode_append)
            self._pending = (key_states.contiguous(), value_states.contiguous(), tk16, args, layer_idx, self.group_size)
            keys_t = None
        elif not self.disable_quant and not ragged_init:
            # K transform + K / V quantisation + append: one launch (fq_kv_quant_append_i4); every request appends `added`
            # tokens at the end of ITS length (last_page_offset is per request)
            ops.kv_quant_append(key_states.contiguous(), value_states.contiguous(), tk16, *args, layer_idx, self.group_size)
            keys_t = None
        else:
            keys_t = key_states.to(torch.float16) if tk16 is None else torch.matmul(key_states.to(torch.float16), tk16)
            if self.disable_quant:                                      # :270-274: fp16 rows, (scale, zero) = (1, 0)
                kq, vq = keys_t.contiguous(), value_states.to(torch.float16).contiguous()
                pk = (b, added, heads, kq.device)
                if getattr(self, "_unit_param_key", None) != pk:   # (kept: a host-to-device copy per step otherwise, and no graph capture)
                    one = torch.tensor([1.0, 0.0], dtype=torch.float16, device=kq.device)
                    self._unit_param, self._unit_param_key = one.expand(b, added, heads, 2).contiguous(), pk
                kp = vp = self._unit_param
            else:
                kq, kp = ops.kv_quant(keys_t.contiguous())
                vq, vp = ops.kv_quant(value_states.to(torch.float16).contiguous())
            kq, vq = kq.reshape(b * added, heads, -1), vq.reshape(b * added, heads, -1)
            kp, vp = kp.reshape(b * added, heads, 2), vp.reshape(b * added, heads, 2)
            if init:
                if mask is not None:                                    # :315-326: keep the tokens whose mask is 1
                    m = mask.to(kq.device)
                    keep = torch.nonzero(m.flatten(), as_tuple=False).flatten()
                    kq, vq, kp, vp = (t.index_select(0, keep).contiguous() for t in (kq, vq, kp, vp))
                    indptr = torch.nn.functional.pad(torch.cumsum(m.sum(dim=-1, dtype=torch.int32), 0, dtype=torch.int32), (1, 0))
                else:
                    indptr = torch.arange(b + 1, device=kq.device, dtype=torch.int32) * added
                ops.kv_append(*args, kq.contiguous(), vq.contiguous(), kp.contiguous(), vp.contiguous(), layer_idx,
                              indptr.contiguous(), self.group_size)
            else:
                ops.kv_append(*args, kq.contiguous(), vq.contiguous(), kp.contiguous(), vp.contiguous(), layer_idx, None,
                              self.group_size)
        if init:
            self._needs_init[layer_idx] = False
            if self.disable_quant:
                return orig_k, orig_v                                   # :262-263 (never re-assigned on this branch)
            if keys_t is None:
                keys_t = key_states if tk16 is None else torch.matmul(key_states.to(torch.float16), tk16)
            return keys_t, value_states                                 # :280-281,341-344: the un-quantised states for prefill
        assert added == 1
        length_now = self.length                                          # (an upper bound of every request's rows: the split decode's hint)

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
                qt = self._as_f16(tk_inv_t)
            pend = self._pending
            if pend is not None and pend[3] is args and pend[4] == layer_idx:      # this step's rows: appended by the decode launch itself
                self._pending = None
                return ops.kv_decode_append(q2.contiguous(), pend[0].view(b, heads, hd), pend[1].view(b, heads, hd), pend[2], *args, layer_idx,
                                            qt, transposed, seq_hint=length_now, read_one_copy=copies > 1).unsqueeze(1)
            self._flush_pending()
            return ops.kv_batch_decode(q2.contiguous(), *args, layer_idx, qt, transposed, seq_hint=length_now, kv_copies=copies).unsqueeze(1)
        return attend
