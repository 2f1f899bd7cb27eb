"""Paged INT4 KV cache: append (bit-exact placement against the numpy restatement of page.cuh) and decode attention
(fp32 online softmax over de-quantised rows; tolerance 2e-3 of the output's maximum against an fp64 softmax attention on
the same cache contents). The reference's kernels are CUDA (vendored FlashInfer) and cannot run here: the oracle restates
their formulas (file:line in oracle/fq_oracle.py) — parity for this row is pinned on those, not on reference outputs."""
import numpy as np
import pytest
import torch

from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


def make_cache(pages, layers, heads, page_size, hd, seed):
    rng = np.random.default_rng(seed)
    data = rng.integers(0, 256, (pages, layers, 2, heads, page_size, hd // 2), dtype=np.uint8)
    param = np.stack([rng.uniform(0.02, 0.3, (pages, layers, 2, heads, page_size)),
                      rng.uniform(0.1, 2.0, (pages, layers, 2, heads, page_size))], axis=-1).astype(np.float16)
    return data, param


def i32(a):
    return torch.tensor(np.asarray(a), dtype=torch.int32, device="cuda")


@pytest.mark.parametrize("hd,heads,page_size,lens", [(128, 4, 16, [37, 37]), (128, 2, 8, [8, 1, 23]), (64, 3, 32, [65]),
                                                      (128, 8, 2048, [300, 300])])
def test_append_prefill_then_decode_steps(ops, hd, heads, page_size, lens):
    layers, layer = 3, 1
    batch = len(lens)
    pages_per = [(n + 1 + page_size - 1) // page_size for n in lens]          # room for one more token each
    indptr = np.concatenate([[0], np.cumsum(pages_per)]).astype(np.int32)
    rng = np.random.default_rng(hd + heads)
    indices = rng.permutation(int(indptr[-1])).astype(np.int32)                  # pages scattered on purpose
    data, param = make_cache(int(indptr[-1]), layers, heads, page_size, hd, 1)
    ref_data, ref_param = data.copy(), param.copy()
    d_data, d_param = torch.from_numpy(data).cuda(), torch.from_numpy(param).cuda()
    # ---- prefill: request b appends lens[b] tokens ----
    tot = sum(lens)
    k = rng.integers(0, 256, (tot, heads, hd // 2), dtype=np.uint8)
    v = rng.integers(0, 256, (tot, heads, hd // 2), dtype=np.uint8)
    kp = rng.uniform(0.01, 1.0, (tot, heads, 2)).astype(np.float16)
    vp = rng.uniform(0.01, 1.0, (tot, heads, 2)).astype(np.float16)
    seq_indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    last = np.array([(n - 1) % page_size + 1 for n in lens], dtype=np.int32)
    used_indptr = np.concatenate([[0], np.cumsum([(n + page_size - 1) // page_size for n in lens])]).astype(np.int32)
    used_indices = np.concatenate([indices[indptr[b]:indptr[b] + used_indptr[b + 1] - used_indptr[b]] for b in range(batch)])
    ops.kv_append(d_data, d_param, i32(used_indptr), i32(used_indices), i32(last), torch.from_numpy(k).cuda(),
                  torch.from_numpy(v).cuda(), torch.from_numpy(kp).cuda(), torch.from_numpy(vp).cuda(), layer, i32(seq_indptr))
    O.kv_cache_append(ref_data, ref_param, used_indptr, used_indices, last, layer, k, v, kp, vp, seq_indptr)
    assert np.array_equal(d_data.cpu().numpy(), ref_data)
    assert np.array_equal(d_param.cpu().numpy().view(np.uint16), ref_param.view(np.uint16))
    # ---- one decode step: every request appends one token, then attends ----
    lens2 = [n + 1 for n in lens]
    last2 = np.array([(n - 1) % page_size + 1 for n in lens2], dtype=np.int32)
    ind2 = np.concatenate([[0], np.cumsum([(n + page_size - 1) // page_size for n in lens2])]).astype(np.int32)
    idx2 = np.concatenate([indices[indptr[b]:indptr[b] + ind2[b + 1] - ind2[b]] for b in range(batch)])
    k1 = rng.integers(0, 256, (batch, heads, hd // 2), dtype=np.uint8)
    v1 = rng.integers(0, 256, (batch, heads, hd // 2), dtype=np.uint8)
    kp1 = rng.uniform(0.01, 1.0, (batch, heads, 2)).astype(np.float16)
    vp1 = rng.uniform(0.01, 1.0, (batch, heads, 2)).astype(np.float16)
    ops.kv_append(d_data, d_param, i32(ind2), i32(idx2), i32(last2), torch.from_numpy(k1).cuda(), torch.from_numpy(v1).cuda(),
                  torch.from_numpy(kp1).cuda(), torch.from_numpy(vp1).cuda(), layer)
    O.kv_cache_append(ref_data, ref_param, ind2, idx2, last2, layer, k1, v1, kp1, vp1)
    assert np.array_equal(d_data.cpu().numpy(), ref_data)
    assert np.array_equal(d_param.cpu().numpy().view(np.uint16), ref_param.view(np.uint16))
    q = (rng.standard_normal((batch, heads, hd)) * 0.5).astype(np.float16)
    o = ops.kv_batch_decode(torch.from_numpy(q).cuda(), d_data, d_param, i32(ind2), i32(idx2), i32(last2), layer).cpu().numpy()
    ref = O.kv_cache_decode(q, ref_data, ref_param, ind2, idx2, last2, layer)
    err = np.abs(o.astype(np.float64) - ref.astype(np.float64)).max(axis=-1) / np.abs(ref.astype(np.float64)).max(axis=-1)
    assert err.max() <= 2e-3, err.max()


def test_cache_class_prefill_and_decode_match_dense_attention(ops):
    """MultiLayerPagedKVCache4Bit end to end against dense fp32 attention on the de-quantised keys / values."""
    import flatquant_amd.deploy.transformers as T
    g = torch.Generator(device="cuda").manual_seed(0)
    bsz, prompt, kv_heads, group, hd, layers = 2, 21, 2, 2, 128, 2
    cache = T.MultiLayerPagedKVCache4Bit(bsz, 16, 64, "cuda", layers, kv_heads * group, hd, trans="matmul", group_size=group)
    tk = (torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5).half()
    tk_inv_t = torch.linalg.inv(tk.float()).T.contiguous().half()
    kw = {"trans_matrix_k": tk, "trans_matrix_k_inv_t": tk_inv_t}
    dense_k = [[] for _ in range(layers)]
    dense_v = [[] for _ in range(layers)]

    def deq32(q8, par):          # n * scale - zero in fp32, as the decode kernel (and quantization.cuh:58-80) evaluates it
        n = torch.stack((q8 & 15, q8 >> 4), dim=-1).reshape(*q8.shape[:-1], -1).float()
        par = par.reshape(*q8.shape[:-1], 2).float()
        return n * par[..., 0:1] - par[..., 1:2]

    def record(layer, k, v):
        kq, kp, vq, vp = T.transform_quantize_kv(k, v, tk)
        dense_k[layer].append(deq32(kq, kp))
        dense_v[layer].append(deq32(vq, vp))

    for layer in range(layers):
        k = torch.randn(bsz, prompt, kv_heads, hd, generator=g, device="cuda").half()
        v = torch.randn(bsz, prompt, kv_heads, hd, generator=g, device="cuda").half()
        out = cache.update(k, v, layer, dict(kw))
        assert isinstance(out, tuple) and out[0].shape == k.shape
        record(layer, k, v)
    assert cache.get_seq_length() == prompt
    for step in range(3):
        for layer in range(layers):
            k = torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda").half()
            v = torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda").half()
            attend = cache.update(k, v, layer, dict(kw))
            record(layer, k, v)
            q = torch.randn(bsz, 1, kv_heads * group, hd, generator=g, device="cuda").half()
            o = attend(q)
            assert o.shape == q.shape
            K = torch.cat(dense_k[layer], dim=1).float().repeat_interleave(group, dim=2)      # [b, s, heads, hd]
            V = torch.cat(dense_v[layer], dim=1).float().repeat_interleave(group, dim=2)
            qt = torch.matmul(q.reshape(bsz, -1, hd).half(), tk_inv_t).float()                  # [b, heads, hd]
            x = torch.einsum("bhd,bshd->bhs", qt, K) / hd ** 0.5
            ref = torch.einsum("bhs,bshd->bhd", torch.softmax(x, dim=-1), V)
            err = (o.reshape(bsz, -1, hd).float() - ref).abs().amax(-1) / ref.abs().amax(-1)
            assert err.max().item() <= 3e-3
    assert cache.get_seq_length() == prompt + 3


def test_errors(ops):
    import flatquant_amd.deploy.transformers as T
    c16 = T.MultiLayerPagedKVCache4Bit(1, 16, 32, "cuda", 1, 2, 128, disable_quant=True)   # (built in round 2)
    assert c16.pages.dtype == torch.float16 and c16.pages.shape[-1] == 128
    data = torch.zeros(2, 1, 2, 2, 16, 48, dtype=torch.uint8, device="cuda")                     # head_dim 96
    par = torch.zeros(2, 1, 2, 2, 16, 2, dtype=torch.float16, device="cuda")
    z = torch.zeros(2, dtype=torch.int32, device="cuda")
    with pytest.raises(Exception):
        ops.kv_batch_decode(torch.zeros(1, 2, 96, dtype=torch.float16, device="cuda"), data, par, z, z, z[:1], 0)


@pytest.mark.parametrize("hd,kv_heads,group,added,lac", [(128, 2, 2, 21, False), (128, 8, 4, 1, False), (64, 3, 1, 5, True),
                                                        (128, 1, 3, 1, True), (128, 8, 8, 1, False), (128, 2, 8, 9, True)])   # (groups of 8: Llama-2/3-70B)
def test_fused_quantise_and_append_equals_the_three_launches(ops, hd, kv_heads, group, added, lac):
    """fq_kv_quant_append_i4 == fq_kv_quant_f16 (keys, with the transform) + fq_kv_quant_f16 (values) + fq_kv_append_i4
    with the GQA repeat: identical cache bytes and parameters."""
    g = torch.Generator(device="cuda").manual_seed(hd + kv_heads + added)
    bsz, page, layers, layer = 3, 16, 2, 1
    heads = kv_heads * group
    prior = 7                                                        # tokens already in the cache
    total = prior + added
    n_pg = (total + page - 1) // page
    shape = (bsz * n_pg, layers, 2, heads, page, hd // 2)
    base_d = torch.randint(0, 256, shape, generator=g, device="cuda", dtype=torch.uint8)
    base_p = torch.rand(*shape[:-1], 2, generator=g, device="cuda").half()
    indptr = (torch.arange(bsz + 1, device="cuda", dtype=torch.int32) * n_pg)
    indices = torch.randperm(bsz * n_pg, generator=g, device="cuda").to(torch.int32)
    last = torch.full((bsz,), (total - 1) % page + 1, device="cuda", dtype=torch.int32)
    k = (torch.randn(bsz, added, kv_heads, hd, generator=g, device="cuda") * 2).half()
    v = (torch.randn(bsz, added, kv_heads, hd, generator=g, device="cuda") * 2).half()
    T = (torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5).half()
    clip = (0.98, 0.9, 0.95, 0.97)
    d1, p1 = base_d.clone(), base_p.clone()
    ops.kv_quant_append(k, v, T, d1, p1, indptr, indices, last, layer, group, clip, lac)
    d2, p2 = base_d.clone(), base_p.clone()
    kq, kp = ops.kv_quant(k, T, clip[:2], lac)
    vq, vp = ops.kv_quant(v, None, clip[2:], lac)
    seq = (torch.arange(bsz + 1, device="cuda", dtype=torch.int32) * added)
    ops.kv_append(d2, p2, indptr, indices, last, kq.reshape(-1, kv_heads, hd // 2), vq.reshape(-1, kv_heads, hd // 2),
                  kp.reshape(-1, kv_heads, 2), vp.reshape(-1, kv_heads, 2), layer, seq, group)
    assert torch.equal(d1, d2)
    assert torch.equal(p1.view(torch.int16), p2.view(torch.int16))
    assert not torch.equal(d1, base_d)


def test_cache_class_hadamard_keys(ops):
    """trans="had" (QuaRot): keys and queries rotated by the normalised Hadamard transform; q.k is invariant under it, so
    the attention output equals dense attention on (rotated, de-quantised) keys with the rotated query."""
    import flatquant_amd.deploy.transformers as T
    g = torch.Generator(device="cuda").manual_seed(11)
    bsz, prompt, heads, hd = 2, 19, 4, 128
    cache = T.MultiLayerPagedKVCache4Bit(bsz, 16, 64, "cuda", 1, heads, hd, trans="had")
    k0 = torch.randn(bsz, prompt, heads, hd, generator=g, device="cuda").half()
    v0 = torch.randn(bsz, prompt, heads, hd, generator=g, device="cuda").half()
    keys, vals = cache.update(k0, v0, 0, {})
    assert torch.equal(keys, ops.hadamard(k0)) and torch.equal(vals, v0)
    k1 = torch.randn(bsz, 1, heads, hd, generator=g, device="cuda").half()
    v1 = torch.randn(bsz, 1, heads, hd, generator=g, device="cuda").half()
    attend = cache.update(k1, v1, 0, {})
    q = torch.randn(bsz, 1, heads, hd, generator=g, device="cuda").half()
    o = attend(q)

    def deq32(x):
        q8, par = ops.kv_quant(x)
        n = torch.stack((q8 & 15, q8 >> 4), dim=-1).reshape(*q8.shape[:-1], -1).float()
        return n * par[..., 0:1].float() - par[..., 1:2].float()
    K = deq32(ops.hadamard(torch.cat([k0, k1], dim=1).contiguous()))
    V = deq32(torch.cat([v0, v1], dim=1).contiguous())
    qr = ops.hadamard(q.reshape(bsz, heads, hd).contiguous()).float()
    x = torch.einsum("bhd,bshd->bhs", qr, K) / hd ** 0.5
    ref = torch.einsum("bhs,bshd->bhd", torch.softmax(x, dim=-1), V)
    err = (o.reshape(bsz, heads, hd).float() - ref).abs().amax(-1) / ref.abs().amax(-1)
    assert err.max().item() <= 3e-3


def test_decode_with_query_transform_and_transposed_output(ops):
    """fq_kv_batch_decode_i4_ex: the query transform inside the launch (vs torch.matmul in front, tolerance: another
    accumulation order before the fp16 rounding of q') and the [batch, head_dim, heads] output layout (exact)."""
    g = torch.Generator(device="cuda").manual_seed(21)
    bsz, heads, hd, page, seq = 3, 4, 128, 16, 45
    n_pg = (seq + page - 1) // page
    data = torch.randint(0, 256, (bsz * n_pg, 1, 2, heads, page, hd // 2), generator=g, device="cuda", dtype=torch.uint8)
    par = (torch.rand(bsz * n_pg, 1, 2, heads, page, 2, generator=g, device="cuda") * 0.2 + 0.05).half()
    indptr = torch.arange(bsz + 1, device="cuda", dtype=torch.int32) * n_pg
    indices = torch.randperm(bsz * n_pg, generator=g, device="cuda").to(torch.int32)
    last = torch.full((bsz,), (seq - 1) % page + 1, device="cuda", dtype=torch.int32)
    q = torch.randn(bsz, heads, hd, generator=g, device="cuda").half()
    Tq = (torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5).half()
    plain = ops.kv_batch_decode(q, data, par, indptr, indices, last, 0)
    tr = ops.kv_batch_decode(q, data, par, indptr, indices, last, 0, transpose_out=True)
    assert tr.shape == (bsz, hd, heads) and torch.equal(tr.transpose(1, 2), plain)
    fused = ops.kv_batch_decode(q, data, par, indptr, indices, last, 0, q_trans=Tq)
    ref = ops.kv_batch_decode(torch.matmul(q, Tq).contiguous(), data, par, indptr, indices, last, 0)
    err = (fused.float() - ref.float()).abs().amax(-1) / ref.float().abs().amax(-1)
    assert err.max().item() <= 3e-3


# ---------------------------------------------------------------------------------------------------------------------
# Round 2: the fp16 configuration of the cache (disable_quant=True: init_kv_f16 / append_kv_f16 / batch_decode_f16,
# kv_cache.py:107-137,177-190) and ragged batches (attention_mask, kv_cache.py:315-326,362-372).
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hd,heads,group,page_size,lens", [(128, 4, 1, 16, [37, 35]), (128, 4, 2, 8, [8, 1, 7]), (64, 6, 3, 32, [65])])
def test_fp16_cache_append_and_decode(ops, hd, heads, group, page_size, lens):
    """fp16 pages: placement bit-exact (rows scattered like the INT4 rows, GQA repeat in the scatter), decode against an fp64
    softmax attention over the same cache contents."""
    layers, layer, batch = 2, 1, len(lens)
    rng = np.random.default_rng(hd + heads + group)
    used_indptr = np.concatenate([[0], np.cumsum([(n + page_size - 1) // page_size for n in lens])]).astype(np.int32)
    n_pages = int(used_indptr[-1]) + 2
    indices = rng.permutation(n_pages).astype(np.int32)[:used_indptr[-1]]
    data = (rng.standard_normal((n_pages, layers, 2, heads, page_size, hd)) * 3).astype(np.float16)
    param = rng.uniform(0.1, 1.0, (n_pages, layers, 2, heads, page_size, 2)).astype(np.float16)
    ref = data.copy()
    d_data, d_param = torch.from_numpy(data).cuda(), torch.from_numpy(param).cuda()
    tot, src = sum(lens), heads // group
    k = rng.standard_normal((tot, src, hd)).astype(np.float16)
    v = rng.standard_normal((tot, src, hd)).astype(np.float16)
    one = np.tile(np.array([1.0, 0.0], np.float16), (tot, src, 1))
    seq_indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    last = np.array([(n - 1) % page_size + 1 for n in lens], dtype=np.int32)
    ops.kv_append(d_data, d_param, i32(used_indptr), i32(indices), i32(last), torch.from_numpy(k).cuda(), torch.from_numpy(v).cuda(),
                  torch.from_numpy(one).cuda(), torch.from_numpy(one).cuda(), layer, i32(seq_indptr), group)
    for b in range(batch):
        for j in range(lens[b]):
            page, entry = indices[used_indptr[b] + j // page_size], j % page_size
            ref[page, layer, 0, :, entry] = np.repeat(k[seq_indptr[b] + j], group, axis=0)
            ref[page, layer, 1, :, entry] = np.repeat(v[seq_indptr[b] + j], group, axis=0)
    got = d_data.cpu().numpy()
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16))
    p = d_param.cpu().numpy()
    for b in range(batch):                                      # the (1, 0) parameters are scattered like the reference's
        page = indices[used_indptr[b]]
        assert np.all(p[page, layer, :, :, 0, 0] == 1.0) and np.all(p[page, layer, :, :, 0, 1] == 0.0)
    q = (rng.standard_normal((batch, heads, hd)) * 0.5).astype(np.float16)
    o = ops.kv_batch_decode(torch.from_numpy(q).cuda(), d_data, d_param, i32(used_indptr), i32(indices), i32(last), layer)
    o = o.cpu().numpy().astype(np.float64)
    for b in range(batch):
        rows = [(indices[used_indptr[b] + j // page_size], j % page_size) for j in range(lens[b])]
        for h in range(heads):
            K = np.array([ref[pg, layer, 0, h, e] for pg, e in rows], dtype=np.float64)
            V = np.array([ref[pg, layer, 1, h, e] for pg, e in rows], dtype=np.float64)
            x = K @ q[b, h].astype(np.float64) / np.sqrt(hd)
            w = np.exp(x - x.max())
            want = (w / w.sum()) @ V
            assert np.abs(o[b, h] - want).max() / np.abs(want).max() <= 2e-3
    # transposed output + query transform, as the INT4 decode takes them
    qt = (rng.standard_normal((hd, hd)) / hd ** 0.5).astype(np.float16)
    o2 = ops.kv_batch_decode(torch.from_numpy(q).cuda(), d_data, d_param, i32(used_indptr), i32(indices), i32(last), layer,
                             torch.from_numpy(qt).cuda(), True)
    q2 = torch.matmul(torch.from_numpy(q).cuda(), torch.from_numpy(qt).cuda())
    o3 = ops.kv_batch_decode(q2.contiguous(), d_data, d_param, i32(used_indptr), i32(indices), i32(last), layer)
    assert (o2.transpose(1, 2).float() - o3.float()).abs().max().item() <= 2e-3 * o3.float().abs().max().item()


def _dense_attention(q, K, V, lens, group, qfun):
    """q [b, 1, heads, hd]; K, V [b, s, kv_heads, hd] fp32 with request b's valid rows first (lens[b] of them)."""
    b, _, heads, hd = q.shape
    out = torch.zeros(b, heads, hd, device=q.device)
    for i in range(b):
        Ki = K[i, :lens[i]].repeat_interleave(group, dim=1)
        Vi = V[i, :lens[i]].repeat_interleave(group, dim=1)
        qi = qfun(q[i, 0])
        x = torch.einsum("hd,shd->hs", qi.float(), Ki) / hd ** 0.5
        out[i] = torch.einsum("hs,shd->hd", torch.softmax(x, dim=-1), Vi)
    return out


@pytest.mark.parametrize("disable_quant,trans", [(False, "matmul"), (True, "matmul"), (True, "none"), (False, "had")])
def test_cache_class_ragged_prompts_and_fp16_configuration(ops, disable_quant, trans):
    """MultiLayerPagedKVCache4Bit with an attention mask (left-padded prompts of different lengths, the mask growing by one column
    per decode step) in both cache configurations, against dense attention over each request's own tokens."""
    import flatquant_amd.deploy.transformers as T
    g = torch.Generator(device="cuda").manual_seed(3)
    bsz, prompt, kv_heads, group, hd, layers, page = 3, 20, 2, 2, 128, 2, 16
    valid = [20, 17, 18]                                        # same page count (2 pages of 16) for every request
    mask = torch.zeros(bsz, prompt, dtype=torch.int64, device="cuda")
    for i, n in enumerate(valid):
        mask[i, prompt - n:] = 1                               # left padding, as HF generates it
    cache = T.MultiLayerPagedKVCache4Bit(bsz, page, 64, "cuda", layers, kv_heads * group, hd, disable_quant=disable_quant,
                                         trans=trans, group_size=group)
    tk = (torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5).half()
    tk_inv_t = torch.linalg.inv(tk.float()).T.contiguous().half()
    kfun = {"matmul": lambda t: torch.matmul(t.half(), tk), "had": lambda t: ops.hadamard(t.half().contiguous()),
            "none": lambda t: t.half()}[trans]                  # what happens to the keys on the way into the cache
    qfun = {"matmul": lambda t: torch.matmul(t.half(), tk_inv_t), "had": lambda t: ops.hadamard(t.half().contiguous()),
            "none": lambda t: t.half()}[trans]                  # ... and to the query
    kw = {"trans_matrix_k": tk, "trans_matrix_k_inv_t": tk_inv_t}

    def stored(k, v):           # the fp32 values the cache holds for these keys / values
        kt = kfun(k)
        if disable_quant:
            return kt.float(), v.float()
        kq, kp = ops.kv_quant(kt.contiguous())
        vq, vp = ops.kv_quant(v.half().contiguous())
        deq = lambda q8, par: (torch.stack((q8 & 15, q8 >> 4), dim=-1).reshape(*q8.shape[:-1], -1).float()
                               * par[..., 0:1].float() - par[..., 1:2].float())
        return deq(kq, kp), deq(vq, vp)

    Ks, Vs = [], []
    for layer in range(layers):
        k = torch.randn(bsz, prompt, kv_heads, hd, generator=g, device="cuda").half()
        v = torch.randn(bsz, prompt, kv_heads, hd, generator=g, device="cuda").half()
        out = cache.update(k, v, layer, dict(kw, attention_mask=mask))
        assert isinstance(out, tuple) and out[0].shape == k.shape
        if disable_quant:
            assert torch.equal(out[0], k) and torch.equal(out[1], v)       # the un-transformed states (kv_cache.py:262-263)
        K, V = stored(k, v)
        Kc, Vc = torch.zeros_like(K), torch.zeros_like(V)
        for i, n in enumerate(valid):                                      # compact: valid tokens first
            Kc[i, :n], Vc[i, :n] = K[i, prompt - n:], V[i, prompt - n:]
        Ks.append([Kc]), Vs.append([Vc])
    lens = list(valid)
    for step in range(3):
        mask = torch.cat([mask, torch.ones(bsz, 1, dtype=mask.dtype, device="cuda")], dim=1)
        lens = [n + 1 for n in lens]
        for layer in range(layers):
            k = torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda").half()
            v = torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda").half()
            attend = cache.update(k, v, layer, dict(kw, attention_mask=mask))
            K1, V1 = stored(k, v)
            Ks[layer].append(K1), Vs[layer].append(V1)
            # request i's rows: its valid prompt tokens, then the decode tokens
            S = prompt + step + 1
            K = torch.zeros(bsz, S, kv_heads, hd, device="cuda")
            V = torch.zeros_like(K)
            for i, n in enumerate(valid):
                K[i, :n], V[i, :n] = Ks[layer][0][i, :n], Vs[layer][0][i, :n]
                for t in range(step + 1):
                    K[i, n + t], V[i, n + t] = Ks[layer][1 + t][i, 0], Vs[layer][1 + t][i, 0]
            q = torch.randn(bsz, 1, kv_heads * group, hd, generator=g, device="cuda").half()
            o = attend(q).reshape(bsz, -1, hd).float()
            ref = _dense_attention(q, K, V, lens, group, qfun)
            err = (o - ref).abs().amax(-1) / ref.abs().amax(-1)
            assert err.max().item() <= 4e-3, (disable_quant, trans, step, layer, err.max().item())
    # requests that need different page counts are refused like the reference refuses them (kv_cache.py:371-372)
    bad = torch.ones(bsz, 40, dtype=torch.int64, device="cuda")
    bad[0, :30] = 0
    with pytest.raises(NotImplementedError):
        cache.get_cache_specs_for_flash_infer(bad)


@pytest.mark.parametrize("cfg", ["i4", "f16"])
def test_cache_class_against_the_reference_class_calls(ops, golden, cfg):
    """tests/golden/kv_class.npz: what the REFERENCE's MultiLayerPagedKVCache4Bit hands to its CUDA entry points (recorded on the
    CPU by tools/gen_golden.py: page tables, offsets, seqlen_indptr, repeated packed keys / values and parameters) for ragged
    prompts, GQA group 2 and two decode steps, in both configurations. The same inputs through this class must leave the same
    bytes in the cache: the recorded calls are replayed into a zeroed cache with the oracle's restatement of page.cuh and
    compared with this class's pages and scales after every step."""
    import flatquant_amd.deploy.transformers as T
    g = golden("kv_class")
    bsz, prompt, kv_heads, group, hd, page = (int(t) for t in g["geom"])
    valid = g["valid"].tolist()
    cache = T.MultiLayerPagedKVCache4Bit(bsz, page, 64, "cuda", 1, kv_heads * group, hd, disable_quant=(cfg == "f16"),
                                         trans="none", group_size=group)
    cache.pages.zero_(), cache.scales.zero_()
    ref_data = np.zeros(tuple(cache.pages.shape), dtype=np.uint8 if cfg == "i4" else np.float16)
    ref_param = np.zeros(tuple(cache.scales.shape), dtype=np.float16)
    mask = torch.zeros(bsz, prompt, dtype=torch.int64, device="cuda")
    for i, n in enumerate(valid):
        mask[i, prompt - n:] = 1

    def call(ci, key):
        return g[f"{cfg}_call{ci}_{key}"]

    def replay(ci):
        seq = call(ci, "seqlen_indptr") if f"{cfg}_call{ci}_seqlen_indptr" in g.files else None
        O.kv_cache_append(ref_data, ref_param, call(ci, "kv_indptr"), call(ci, "kv_indices"), call(ci, "last_page_offset"), 0,
                          call(ci, "k"), call(ci, "v"), call(ci, "k_param"), call(ci, "v_param"), seq)

    def same_tables(ci):
        s = cache.get_cache_specs_for_flash_infer(mask)
        for key in ("kv_indptr", "kv_indices", "last_page_offset"):
            assert np.array_equal(s[key].cpu().numpy(), call(ci, key)), (ci, key)

    def same_cache(ci):
        got = cache.pages.cpu().numpy()
        assert np.array_equal(got.view(np.uint8 if cfg == "i4" else np.uint16), ref_data.view(np.uint8 if cfg == "i4" else np.uint16)), ci
        assert np.array_equal(cache.scales.cpu().numpy().view(np.uint16), ref_param.view(np.uint16)), ci

    out = cache.update(torch.from_numpy(g[f"{cfg}_k0"]).cuda(), torch.from_numpy(g[f"{cfg}_v0"]).cuda(), 0, {"attention_mask": mask})
    assert str(g[f"{cfg}_call0_name"]) == ("init_kv_i4" if cfg == "i4" else "init_kv_f16")
    same_tables(0), replay(0), same_cache(0)
    assert np.array_equal(out[0].cpu().numpy(), g[f"{cfg}_ret_k"]) and np.array_equal(out[1].cpu().numpy(), g[f"{cfg}_ret_v"])
    ci = 1
    for step in (1, 2):
        mask = torch.cat([mask, torch.ones(bsz, 1, dtype=mask.dtype, device="cuda")], dim=1)
        attend = cache.update(torch.from_numpy(g[f"{cfg}_k{step}"]).cuda(), torch.from_numpy(g[f"{cfg}_v{step}"]).cuda(), 0,
                              {"attention_mask": mask})
        same_tables(ci), replay(ci), same_cache(ci)
        ci += 1
        # the decode call: same page tables; the output against the oracle's attention over the replayed cache
        q = torch.from_numpy(g[f"{cfg}_q{step}"]).cuda()
        assert np.array_equal(call(ci, "q"), g[f"{cfg}_q{step}"].reshape(bsz, kv_heads * group, hd))
        same_tables(ci)
        o = attend(q).reshape(bsz, -1, hd).cpu().numpy().astype(np.float64)
        if cfg == "i4":
            want = O.kv_cache_decode(call(ci, "q"), ref_data, ref_param, call(ci, "kv_indptr"), call(ci, "kv_indices"),
                                     call(ci, "last_page_offset"), 0).astype(np.float64)
            assert (np.abs(o - want).max(axis=-1) / np.abs(want).max(axis=-1)).max() <= 3e-3
        ci += 1
    assert ci == int(g[f"{cfg}_n_calls"])


# ---------------------------------------------------------------------------------------------------------------------
# Round 5: a request's rows split over several workgroups (fq_kv_batch_decode_split)
@pytest.mark.parametrize("fp16", [False, True])
@pytest.mark.parametrize("bsz,heads,hd,page,lens", [(1, 32, 128, 2048, [2048]), (2, 4, 128, 16, [700, 333]), (3, 8, 64, 32, [65, 1000, 1]),
                                                   (1, 2, 128, 8, [523]), (4, 64, 128, 256, [256, 255, 257, 3]), (2, 3, 128, 2, [37, 90])])
def test_split_decode_equals_the_unsplit_launch(ops, fp16, bsz, heads, hd, page, lens):
    """With at most 128 (request, head) pairs ops.kv_batch_decode takes the split launch: the partial states of a pair's workgroups meet in a
    workspace and the last one merges them — the unsplit launch's result up to the order of fp32 additions, for both cache configurations,
    ragged lengths (requests shorter than a split's share), pages a wave's rows straddle, the query transform and the transposed output;
    a second launch finds the counters at zero again."""
    g = torch.Generator(device="cuda").manual_seed(bsz * 100 + heads + page)
    n_pg = [(n + page - 1) // page for n in lens]
    tot = sum(n_pg)
    if fp16:
        data = torch.randn(tot, 2, 2, heads, page, hd, generator=g, device="cuda").half()
    else:
        data = torch.randint(0, 256, (tot, 2, 2, heads, page, hd // 2), generator=g, device="cuda", dtype=torch.uint8)
    par = (torch.rand(tot, 2, 2, heads, page, 2, generator=g, device="cuda") * 0.2 + 0.05).half()
    indptr = torch.tensor(np.concatenate([[0], np.cumsum(n_pg)]), dtype=torch.int32, device="cuda")
    indices = torch.randperm(tot, generator=g, device="cuda").to(torch.int32)
    last = torch.tensor([(n - 1) % page + 1 for n in lens], dtype=torch.int32, device="cuda")
    q = torch.randn(bsz, heads, hd, generator=g, device="cuda").half()
    Tq = (torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5).half()
    for hint in (0, max(lens)):
        for kw in ({}, {"q_trans": Tq, "transpose_out": True}):
            ref = ops.kv_batch_decode(q, data, par, indptr, indices, last, 1, split=False, **kw).float()
            for _ in range(2):
                got = ops.kv_batch_decode(q, data, par, indptr, indices, last, 1, seq_hint=hint, **kw).float()
                assert got.shape == ref.shape and torch.isfinite(got).all()
                err = (got - ref).abs().amax() / ref.abs().amax()
                assert err.item() <= 1e-3, (hint, kw.keys(), err.item())


def test_split_decode_workspace_contract(ops):
    """fq_kv_decode_workspace_bytes: 0 beyond 128 pairs (never split), else room for 16 splits; the launch refuses a workspace that is too small."""
    import ctypes
    from flatquant_amd import _lib
    lib = _lib.lib
    assert lib.fq_kv_decode_workspace_bytes(16, 32, 128) == 0 and lib.fq_kv_decode_workspace_bytes(1, 1, 96) == -1
    need = lib.fq_kv_decode_workspace_bytes(2, 32, 128)
    assert need == 64 * 4 + 64 * 16 * 130 * 4
    data = torch.zeros(2, 1, 2, 32, 16, 64, dtype=torch.uint8, device="cuda")
    par = torch.ones(2, 1, 2, 32, 16, 2, dtype=torch.float16, device="cuda")
    z = torch.tensor([0, 1, 2], dtype=torch.int32, device="cuda")
    one = torch.tensor([16, 16], dtype=torch.int32, device="cuda")
    q = torch.zeros(2, 32, 128, dtype=torch.float16, device="cuda")
    o = torch.empty_like(q)
    ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
    args = lambda w, n: (0, o.data_ptr(), q.data_ptr(), None, 0, data.data_ptr(), par.data_ptr(), z.data_ptr(), z.data_ptr(), one.data_ptr(),
                         1, 0, 32, 16, 128, 2, 0, w, n, None)
    assert lib.fq_kv_batch_decode_split(*args(ws.data_ptr(), need - 16)) == _lib.FQ_EINVAL
    assert lib.fq_kv_batch_decode_split(*args(ws.data_ptr(), need)) == 0
    assert lib.fq_kv_batch_decode_split(*args(None, 0)) == 0              # no workspace: the unsplit launch
    torch.cuda.synchronize()
    assert int(ws.view(torch.int32)[:64].abs().sum()) == 0                # the counters are zero again


def test_split_decode_is_bit_stable_over_many_launches(ops):
    """The split launch's workgroups hand their partial states over through agent-scope stores, a counter and agent-scope loads (no fence:
    fq_kvcache.hip). A state read before it has arrived would change the result: 600 launches on the same inputs — two geometries, alternating,
    so that consecutive launches of a geometry reuse workspace lines the other one's neighbours just wrote — must give the same bits every time
    (the merge order is fixed) and stay within tolerance of the unsplit launch."""
    g = torch.Generator(device="cuda").manual_seed(77)
    cases = []
    for bsz, heads, seq, page in ((1, 32, 2048, 2048), (3, 16, 1500, 16)):
        n_pg = (seq + page - 1) // page
        data = torch.randint(0, 256, (bsz * n_pg, 1, 2, heads, page, 64), generator=g, device="cuda", dtype=torch.uint8)
        par = (torch.rand(bsz * n_pg, 1, 2, heads, page, 2, generator=g, device="cuda") * 0.2 + 0.05).half()
        indptr = torch.arange(bsz + 1, device="cuda", dtype=torch.int32) * n_pg
        indices = torch.randperm(bsz * n_pg, generator=g, device="cuda").to(torch.int32)
        last = torch.full((bsz,), (seq - 1) % page + 1, device="cuda", dtype=torch.int32)
        q = torch.randn(bsz, heads, 128, generator=g, device="cuda").half()
        args = (q, data, par, indptr, indices, last, 0)
        ref = ops.kv_batch_decode(*args, split=False).float()
        first = ops.kv_batch_decode(*args, seq_hint=seq)
        assert ((first.float() - ref).abs().amax() / ref.abs().amax()).item() <= 1e-3
        cases.append((args, seq, first))
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    for _ in range(300):
        for args, seq, first in cases:
            bad += (ops.kv_batch_decode(*args, seq_hint=seq) != first).sum()     # (compared on the device: nothing waits for the host)
    assert int(bad) == 0


@pytest.mark.parametrize("disable_quant", [False, True])
@pytest.mark.parametrize("bsz,prompt,kv_heads,group,hd,page", [(3, 37, 2, 4, 128, 16), (1, 300, 8, 4, 128, 64), (5, 20, 4, 2, 64, 16), (40, 70, 2, 4, 128, 32),
                                                               # >= 32 (request, KV head) pairs at head_dim 128: ONE workgroup serves the 4 (2) query heads of a KV head
                                                               (64, 33, 4, 4, 128, 16), (40, 50, 8, 2, 128, 32), (33, 70, 8, 4, 128, 64)])
def test_shared_kv_heads_cache_equals_the_replicated_cache(ops, disable_quant, bsz, prompt, kv_heads, group, hd, page):
    """share_kv_heads=True (round 6, extension): the pages hold the KV heads once, query head h reads cache head h // group
    (fq_kv_batch_decode_gqa). Same rows, same arithmetic: the attention output is BIT-identical to the reference-shaped cache's (one copy
    per query head, kv_cache.py:286-296) — the query transform, the transposed output, both cache dtypes, a workgroup per query head and (from
    256 pairs on) one per KV head with its query heads riding in the q . k MFMA's idle rows."""
    import flatquant_amd.deploy.transformers as T
    g = torch.Generator(device="cuda").manual_seed(bsz * 100 + prompt)
    heads = kv_heads * group
    mk = lambda share: T.MultiLayerPagedKVCache4Bit(bsz, page, prompt + 8, "cuda", 2, heads, hd, trans="matmul", group_size=group,
                                                    disable_quant=disable_quant, share_kv_heads=share)
    rep, sh = mk(False), mk(True)
    rep.pages.zero_(), sh.pages.zero_()      # (torch.empty pages: the entries no token reached are compared below too)
    assert sh.pages.shape[3] == kv_heads and rep.pages.shape[3] == heads and sh.pages.numel() * group == rep.pages.numel()
    tk = (torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5).half()
    kw = {"trans_matrix_k": tk, "trans_matrix_k_inv_t": torch.linalg.inv(tk.float()).T.contiguous().half()}
    for layer in range(2):
        k = torch.randn(bsz, prompt, kv_heads, hd, generator=g, device="cuda").half()
        v = torch.randn(bsz, prompt, kv_heads, hd, generator=g, device="cuda").half()
        rep.update(k, v, layer, dict(kw)), sh.update(k, v, layer, dict(kw))
    for step in range(4):
        for layer in range(2):
            k = torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda").half()
            v = torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda").half()
            a_rep, a_sh = rep.update(k, v, layer, dict(kw)), sh.update(k, v, layer, dict(kw))
            q = torch.randn(bsz, 1, heads, hd, generator=g, device="cuda").half()
            merged = hd == 128 and group in (2, 4) and bsz * kv_heads >= 32   # one workgroup per KV head: other lanes sum other rows
            for transposed in (False, True):
                o_rep, o_sh = a_rep(q, transposed=transposed), a_sh(q, transposed=transposed)
                if not merged:
                    assert torch.equal(o_rep, o_sh), (step, layer, transposed)
                else:   # as the split launch against the unsplit one: the order of fp32 additions differs, nothing else — the fp16 outputs
                        # differ by at most two units in the last place of the largest (measured: one, 5.7e-4 .. 9.3e-4 of the maximum)
                    err = (o_rep.float() - o_sh.float()).abs().amax() / o_rep.float().abs().amax()
                    assert torch.isfinite(o_sh).all() and err.item() <= 2.5e-3, (step, layer, transposed, err.item())
    # the rows themselves: cache head j of the shared pages == every one of its `group` copies in the replicated pages
    assert torch.equal(rep.pages[:sh.pages.shape[0]].reshape(-1, 2, 2, kv_heads, group, *rep.pages.shape[4:])[:, :, :, :, 0], sh.pages.reshape(-1, 2, 2, kv_heads, *sh.pages.shape[4:]))
    # no split workspace (split=False) and the plain entry without a query transform
    specs_r, specs_s = rep._specs, sh._specs
    q2 = torch.randn(bsz, heads, hd, generator=g, device="cuda").half()
    ar = (specs_r["kv_data"], specs_r["kv_param"], specs_r["kv_indptr"], specs_r["kv_indices"], specs_r["last_page_offset"])
    as_ = (specs_s["kv_data"], specs_s["kv_param"], specs_s["kv_indptr"], specs_s["kv_indices"], specs_s["last_page_offset"])
    o_r, o_s = ops.kv_batch_decode(q2, *ar, 1, split=False), ops.kv_batch_decode(q2, *as_, 1, split=False)
    if hd == 128 and group in (2, 4) and bsz * kv_heads >= 32:
        assert ((o_r.float() - o_s.float()).abs().amax() / o_r.float().abs().amax()).item() <= 2.5e-3
    else:
        assert torch.equal(o_r, o_s)


def test_merged_query_heads_match_dense_attention(ops):
    """the workgroup-per-KV-head launch (64 requests x 4 KV heads x 4 query heads each, shared cache) against dense fp32 attention on the
    de-quantised rows — independent of the per-query-head kernel"""
    import flatquant_amd.deploy.transformers as T
    g = torch.Generator(device="cuda").manual_seed(9)
    bsz, prompt, kv_heads, group, hd = 64, 45, 4, 4, 128
    cache = T.MultiLayerPagedKVCache4Bit(bsz, 16, 64, "cuda", 1, kv_heads * group, hd, trans="matmul", group_size=group, share_kv_heads=True)
    tk = (torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5).half()
    tk_inv_t = torch.linalg.inv(tk.float()).T.contiguous().half()
    kw = {"trans_matrix_k": tk, "trans_matrix_k_inv_t": tk_inv_t}

    def deq32(q8, par):
        n = torch.stack((q8 & 15, q8 >> 4), dim=-1).reshape(*q8.shape[:-1], -1).float()
        par = par.reshape(*q8.shape[:-1], 2).float()
        return n * par[..., 0:1] - par[..., 1:2]

    ks, vs = [], []
    k = torch.randn(bsz, prompt, kv_heads, hd, generator=g, device="cuda").half()
    v = torch.randn(bsz, prompt, kv_heads, hd, generator=g, device="cuda").half()
    cache.update(k, v, 0, dict(kw))
    kq, kp, vq, vp = T.transform_quantize_kv(k, v, tk)
    ks.append(deq32(kq, kp)), vs.append(deq32(vq, vp))
    for step in range(2):
        k = torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda").half()
        v = torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda").half()
        attend = cache.update(k, v, 0, dict(kw))
        kq, kp, vq, vp = T.transform_quantize_kv(k, v, tk)
        ks.append(deq32(kq, kp)), vs.append(deq32(vq, vp))
        q = torch.randn(bsz, 1, kv_heads * group, hd, generator=g, device="cuda").half()
        o = attend(q)
        K = torch.cat(ks, dim=1).float().repeat_interleave(group, dim=2)
        V = torch.cat(vs, dim=1).float().repeat_interleave(group, dim=2)
        qt = torch.matmul(q.reshape(bsz, -1, hd).half(), tk_inv_t).float()
        x = torch.einsum("bhd,bshd->bhs", qt, K) / hd ** 0.5
        ref = torch.einsum("bhs,bshd->bhd", torch.softmax(x, dim=-1), V)
        err = (o.reshape(bsz, -1, hd).float() - ref).abs().amax(-1) / ref.abs().amax(-1)
        assert err.max().item() <= 3e-3
