"""The decode step's append inside the decode launch (fq_kv_decode_append_i4, round 6): the launch that quantises and appends the step's own
K / V row must leave the SAME cache bytes and return the SAME attention output, bit for bit, as fq_kv_quant_append_i4 followed by the decode
launch (deploy/transformers/kv_cache.py:283-359 is those two steps) — on the replicated and the shared cache, split and unsplit launches, four
and eight waves, with and without the K transform, ragged lengths, the new row at every place of a wave's 16-row step, on a fresh page, as the
only row. The two-launch sequence itself is held to the oracle by tests/test_gpu_kvcache.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


def _cache(g, bsz, lens, cache_heads, page, hd=128, layers=2):
    """a cache whose request b holds lens[b] rows (the LAST one is the slot of the new token), pages scattered"""
    n_pg = [(n + page - 1) // page for n in lens]
    tot = sum(n_pg)
    data = torch.randint(0, 256, (tot, layers, 2, cache_heads, page, hd // 2), generator=g, device="cuda", dtype=torch.uint8)
    param = torch.stack([torch.rand(tot, layers, 2, cache_heads, page, generator=g, device="cuda") * 0.3 + 0.02,
                         torch.rand(tot, layers, 2, cache_heads, page, generator=g, device="cuda") * 2.0 + 0.1], dim=-1).half()
    indptr = torch.tensor([0] + list(torch.tensor(n_pg).cumsum(0)), dtype=torch.int32, device="cuda")
    indices = torch.randperm(tot, generator=g, device="cuda").to(torch.int32)
    last = torch.tensor([(n - 1) % page + 1 for n in lens], dtype=torch.int32, device="cuda")
    return data, param, indptr, indices, last


CASES = [
    # bsz, lens, src_heads, copies per source head in the cache, query heads per cache head, page, K transform
    (1, [2048], 8, 4, 1, 2048, True),          # Llama-3-8B, one request: the split launch (16 workgroups per pair)
    (1, [2049], 8, 1, 4, 2048, True),          # ... on the shared cache, the new row on a fresh page
    (2, [700, 333], 2, 2, 1, 16, True),        # split, ragged, small pages
    (3, [1, 16, 17], 2, 1, 2, 16, True),       # the new row is the only row / the last of a step / the first of the next
    (4, [31, 32, 33, 48], 4, 1, 1, 16, False),  # no transform (QuaRot keys arrive rotated)
    (16, [2048] * 16, 8, 4, 1, 2048, True),    # 512 pairs: unsplit, four waves
    (5, [100, 260, 90, 1000, 515], 8, 4, 1, 32, True),   # 160 pairs: eight waves, unsplit
    (32, [300 + 7 * i for i in range(32)], 8, 1, 4, 64, True),   # shared cache, 256 (request, KV head) pairs: one workgroup serves 4 query heads
    (40, [70 + i for i in range(40)], 8, 1, 2, 16, False),       # ... 2 query heads
    (9, [50 + 3 * i for i in range(9)], 2, 4, 1, 16, True),      # 72 pairs, short rows: split count limited by the length hint
    (2, [600, 90], 8, 8, 1, 32, True),                           # groups of EIGHT copies (Llama-2/3-70B: 64 query heads on 8 KV heads)
    (20, [200 + i for i in range(20)], 8, 1, 8, 64, True),       # ... on the shared cache: 160 (request, KV head) pairs, two workgroups of four heads per group
]


@pytest.mark.parametrize("bsz,lens,src_heads,copies,q_group,page,trans", CASES)
@pytest.mark.parametrize("transpose_out", [False, True])
def test_decode_append_equals_quant_append_then_decode(ops, bsz, lens, src_heads, copies, q_group, page, trans, transpose_out):
    g = torch.Generator(device="cuda").manual_seed(bsz * 1000 + lens[0] + src_heads)
    hd, layer = 128, 1
    cache_heads, heads = src_heads * copies, src_heads * copies * q_group
    data, param, indptr, indices, last = _cache(g, bsz, lens, cache_heads, page)
    k = (torch.randn(bsz, 1, src_heads, hd, generator=g, device="cuda") * 2).half()
    v = (torch.randn(bsz, 1, src_heads, hd, generator=g, device="cuda") * 2).half()
    T = (torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5).half() if trans else None
    qt = (torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5).half() if trans else None
    q = (torch.randn(bsz, heads, hd, generator=g, device="cuda") * 0.5).half()
    hint = max(lens)
    d1, p1 = data.clone(), param.clone()
    ops.kv_quant_append(k, v, T, d1, p1, indptr, indices, last, layer, copies)
    o1 = ops.kv_batch_decode(q, d1, p1, indptr, indices, last, layer, qt, transpose_out, seq_hint=hint)
    d2, p2 = data.clone(), param.clone()
    o2 = ops.kv_decode_append(q, k.view(bsz, src_heads, hd), v.view(bsz, src_heads, hd), T, d2, p2, indptr, indices, last, layer, qt, transpose_out,
                              seq_hint=hint)
    assert not torch.equal(d1, data)                       # (the step did append something)
    assert torch.equal(d2, d1)
    assert torch.equal(p2.view(torch.int16), p1.view(torch.int16))
    assert torch.equal(o2, o1)
    assert torch.isfinite(o2.float()).all()
    # ... and without the split launch's workspace (what a capture that finds none runs)
    d3, p3 = data.clone(), param.clone()
    o3 = ops.kv_decode_append(q, k.view(bsz, src_heads, hd), v.view(bsz, src_heads, hd), T, d3, p3, indptr, indices, last, layer, qt, transpose_out,
                              seq_hint=hint, split=False)
    o1u = ops.kv_batch_decode(q, d1, p1, indptr, indices, last, layer, qt, transpose_out, seq_hint=hint, split=False)
    assert torch.equal(d3, d1) and torch.equal(p3.view(torch.int16), p1.view(torch.int16)) and torch.equal(o3, o1u)


def test_new_row_at_every_slot_of_a_step_and_repeated_launches(ops):
    """lengths 1 .. 40 (the new row at each of a wave's 16 slots, in the first / second / third step), the same launch five times: bit-stable"""
    g = torch.Generator(device="cuda").manual_seed(5)
    hd, layer, src_heads, copies, page = 128, 0, 2, 2, 16
    lens = list(range(1, 41))
    bsz = len(lens)
    data, param, indptr, indices, last = _cache(g, bsz, lens, src_heads * copies, page)
    k = (torch.randn(bsz, 1, src_heads, hd, generator=g, device="cuda")).half()
    v = (torch.randn(bsz, 1, src_heads, hd, generator=g, device="cuda")).half()
    T = (torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5).half()
    q = (torch.randn(bsz, src_heads * copies, hd, generator=g, device="cuda") * 0.5).half()
    d1, p1 = data.clone(), param.clone()
    ops.kv_quant_append(k, v, T, d1, p1, indptr, indices, last, layer, copies)
    o1 = ops.kv_batch_decode(q, d1, p1, indptr, indices, last, layer, T)
    for _ in range(5):
        d2, p2 = data.clone(), param.clone()
        o2 = ops.kv_decode_append(q, k.view(bsz, src_heads, hd), v.view(bsz, src_heads, hd), T, d2, p2, indptr, indices, last, layer, T)
        assert torch.equal(d2, d1) and torch.equal(p2.view(torch.int16), p1.view(torch.int16)) and torch.equal(o2, o1)


def test_zero_rows_and_extreme_values(ops):
    """an all-zero K / V row (the 1e-5 floor of the scale), a constant row, large magnitudes: same bytes as the two launches"""
    g = torch.Generator(device="cuda").manual_seed(9)
    hd, layer, src_heads, page = 128, 1, 4, 16
    lens = [20, 21, 22, 23]
    data, param, indptr, indices, last = _cache(g, 4, lens, src_heads, page)
    k = torch.randn(4, 1, src_heads, hd, generator=g, device="cuda").half()
    v = torch.randn(4, 1, src_heads, hd, generator=g, device="cuda").half()
    k[0], v[0] = 0, 0
    k[1], v[1] = 3.0, -2.5
    k[2] *= 900
    v[2] *= 3000
    T = (torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5).half()
    q = torch.randn(4, src_heads, hd, generator=g, device="cuda").half()
    for trans in (T, None):
        d1, p1 = data.clone(), param.clone()
        ops.kv_quant_append(k, v, trans, d1, p1, indptr, indices, last, layer, 1)
        o1 = ops.kv_batch_decode(q, d1, p1, indptr, indices, last, layer)
        d2, p2 = data.clone(), param.clone()
        o2 = ops.kv_decode_append(q, k.view(4, src_heads, hd), v.view(4, src_heads, hd), trans, d2, p2, indptr, indices, last, layer)
        assert torch.equal(d2, d1) and torch.equal(p2.view(torch.int16), p1.view(torch.int16))
        assert torch.equal(o2.view(torch.int16), o1.view(torch.int16))


def test_unsupported_geometries_are_refused(ops):
    from flatquant_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(1)
    data, param, indptr, indices, last = _cache(g, 1, [20], 2, 8)           # page_size 8: a wave's 16 rows straddle pages
    assert not ops.kv_decode_append_supported(data, 2)
    k = torch.randn(1, 2, 128, generator=g, device="cuda").half()
    with pytest.raises(_lib.FqError):
        ops.kv_decode_append(torch.randn(1, 2, 128, generator=g, device="cuda").half(), k, k, None, data, param, indptr, indices, last, 0)
    data64 = torch.zeros(2, 1, 2, 2, 16, 32, dtype=torch.uint8, device="cuda")   # head_dim 64
    assert not ops.kv_decode_append_supported(data64, 2)
    assert not ops.kv_decode_append_supported(torch.zeros(2, 1, 2, 2, 16, 128, dtype=torch.float16, device="cuda"), 2)   # the fp16 configuration
    ok, _, _, _, _ = _cache(g, 1, [20], 8, 16)
    assert ops.kv_decode_append_supported(ok, 2) and ops.kv_decode_append_supported(ok, 1) and not ops.kv_decode_append_supported(ok, 3)   # (8 = 1 x 8 copies: supported since groups of eight are)
    big, _, _, _, _ = _cache(g, 1, [20], 16, 16)
    assert not ops.kv_decode_append_supported(big, 1)                                                      # sixteen copies are not


@pytest.mark.parametrize("trans", ["matmul", "had", "none"])
@pytest.mark.parametrize("share", [False, True])
def test_cache_class_fused_append_equals_the_two_launches(ops, trans, share):
    """MultiLayerPagedKVCache4Bit(fuse_append=True) against fuse_append=False over a prefill and six decode steps of two layers: the same
    attention outputs and the same pages, bit for bit; a closure that is never called leaves its rows to the next update (flush)."""
    import flatquant_amd.deploy.transformers as dt
    g = torch.Generator(device="cuda").manual_seed(3)
    bsz, prompt, kv_heads, group, hd, page, layers = 3, 45, 2, 4, 128, 16, 2
    heads = kv_heads * group
    tk = (torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5).half()
    kw = {"trans_matrix_k": tk, "trans_matrix_k_inv_t": tk}
    caches = [dt.MultiLayerPagedKVCache4Bit(bsz, page, prompt + 8, torch.device("cuda"), layers, heads, hd, trans=trans, group_size=group,
                                            share_kv_heads=share, fuse_append=f) for f in (True, False)]
    for c in caches:                # (the rows beyond a request's length are whatever torch.empty found: make them comparable)
        c.pages.zero_(), c.scales.zero_()
    for li in range(layers):
        k0 = torch.randn(bsz, prompt, kv_heads, hd, generator=g, device="cuda").half()
        v0 = torch.randn(bsz, prompt, kv_heads, hd, generator=g, device="cuda").half()
        for c in caches:
            c.update(k0, v0, li, dict(kw))
    for step in range(6):
        for li in range(layers):
            k1 = torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda").half()
            v1 = torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda").half()
            q1 = torch.randn(bsz, 1, heads, hd, generator=g, device="cuda").half()
            outs = []
            for c in caches:
                attend = c.update(k1, v1, li, dict(kw))
                if step == 3 and li == 0:
                    outs.append(None)              # the closure is dropped: the rows must still reach the cache (flushed by the next update)
                else:
                    outs.append(attend(q1, transposed=(step % 2 == 1)))
            if outs[0] is not None:
                assert torch.equal(outs[0], outs[1]), (step, li)
                assert torch.isfinite(outs[0].float()).all()
    assert caches[0].length == caches[1].length == prompt + 6
    used = caches[0].page_cnt_from_length(caches[0].length) * bsz
    assert torch.equal(caches[0].pages[:used], caches[1].pages[:used])
    assert torch.equal(caches[0].scales[:used].view(torch.int16), caches[1].scales[:used].view(torch.int16))


def test_fused_append_inside_a_captured_step(ops):
    """the fused launch under stream capture (deploy.GraphedDecode's case): replays append one row per step like the eager calls"""
    import flatquant_amd.deploy.transformers as dt
    g = torch.Generator(device="cuda").manual_seed(4)
    bsz, prompt, kv_heads, group, hd, page = 2, 30, 8, 4, 128, 2048
    heads = kv_heads * group
    tk = (torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5).half()
    kw = {"trans_matrix_k": tk, "trans_matrix_k_inv_t": tk}
    caches = [dt.MultiLayerPagedKVCache4Bit(bsz, page, 64, torch.device("cuda"), 1, heads, hd, trans="matmul", group_size=group, fuse_append=f)
              for f in (True, False)]
    for c in caches:
        c.pages.zero_(), c.scales.zero_()
    k0 = torch.randn(bsz, prompt, kv_heads, hd, generator=g, device="cuda").half()
    v0 = torch.randn(bsz, prompt, kv_heads, hd, generator=g, device="cuda").half()
    for c in caches:
        c.update(k0, v0, 0, dict(kw))
    k1 = torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda").half()
    v1 = torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda").half()
    q1 = torch.randn(bsz, 1, heads, hd, generator=g, device="cuda").half()

    def step(c):
        return c.update(k1, v1, 0, dict(kw))(q1)

    ref = [step(caches[1]).clone() for _ in range(4)]          # eager, two launches: steps 1 .. 4
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        o = step(caches[0])                                     # warm-up on the capture stream (workspace, image): step 1
        assert torch.equal(o, ref[0])
        torch.cuda.synchronize()
        ops.images_ready()
        caches[0]._host_step(0, 1)                              # step 2's host side, eagerly: inside the capture the in-place fill of the index
        caches[0]._skip_host = True                             # tensors would become a node of the graph and be replayed (deploy.graphed does the same)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            out = step(caches[0])                               # captured: step 2 (not executed)
        caches[0]._skip_host = False
    torch.cuda.synchronize()
    for i in (1, 2, 3):                                         # replays: steps 2, 3, 4 — the host side advances the index tensors in place
        if i > 1:
            caches[0]._host_step(0, 1)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref[i]), i
    used = caches[0].page_cnt_from_length(caches[0].length) * bsz
    assert caches[0].length == caches[1].length
    assert torch.equal(caches[0].pages[:used], caches[1].pages[:used])


def test_fp32_transform_matrices_are_converted_once(ops):
    """modeling_llama.py:185-186 registers trans_matrix_k as an fp32 buffer: the cache converts it once per source tensor (not per step), the fused
    append builds ONE fragment image for it, and the outputs equal those of the fp16 matrices"""
    import flatquant_amd.deploy.transformers as dt
    g = torch.Generator(device="cuda").manual_seed(8)
    bsz, prompt, kv_heads, group, hd, page = 2, 20, 2, 2, 128, 16
    heads = kv_heads * group
    tk32 = torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5
    tk16 = tk32.half()
    caches = [dt.MultiLayerPagedKVCache4Bit(bsz, page, 64, torch.device("cuda"), 1, heads, hd, trans="matmul", group_size=group) for _ in range(2)]
    kws = [{"trans_matrix_k": tk32, "trans_matrix_k_inv_t": tk32}, {"trans_matrix_k": tk16, "trans_matrix_k_inv_t": tk16}]
    k0 = torch.randn(bsz, prompt, kv_heads, hd, generator=g, device="cuda").half()
    v0 = torch.randn(bsz, prompt, kv_heads, hd, generator=g, device="cuda").half()
    for c, kw in zip(caches, kws):
        c.update(k0, v0, 0, dict(kw))
    n0 = ops.cache_stats()["kv_transform_images"]["entries"]
    for _ in range(5):
        k1 = torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda").half()
        v1 = torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda").half()
        q1 = torch.randn(bsz, 1, heads, hd, generator=g, device="cuda").half()
        o = [c.update(k1, v1, 0, dict(kw))(q1) for c, kw in zip(caches, kws)]
        assert torch.equal(o[0], o[1])
    assert len(caches[0]._f16_memo) == 1 and "_f16_memo" not in caches[1].__dict__
    assert ops.cache_stats()["kv_transform_images"]["entries"] - n0 <= 2       # one image per distinct fp16 matrix, not one per step


def _replicated(g, bsz, lens, kv_heads, copies, page, f16=False, hd=128, layers=2):
    """a cache in the reference's replicated layout: cache head c is a copy of KV head c // copies (kv_cache.py:286-296)"""
    data, param, indptr, indices, last = _cache(g, bsz, lens, kv_heads, page, hd, layers)
    if f16:
        data = torch.randn(data.shape[:-1] + (hd,), generator=g, device="cuda").half()
    return data.repeat_interleave(copies, dim=3).contiguous(), param.repeat_interleave(copies, dim=3).contiguous(), indptr, indices, last


@pytest.mark.parametrize("f16", [False, True])
@pytest.mark.parametrize("bsz,lens,kv_heads,copies,page", [(1, [2048], 8, 4, 2048), (2, [700, 333], 2, 2, 16), (16, [500 + i for i in range(16)], 8, 4, 64),
                                                           (32, [300 + 7 * i for i in range(32)], 8, 4, 64), (130, [40 + i for i in range(130)], 2, 2, 16),
                                                           (3, [1, 16, 17], 4, 3, 16), (2, [900, 333], 8, 8, 64), (17, [150 + 9 * i for i in range(17)], 8, 8, 32),
                                                           (4, [2048, 2047, 1500, 2049], 8, 4, 2048), (5, [1900 + 31 * i for i in range(5)], 8, 4, 64), (7, [700] * 7, 8, 4, 32)])   # (32 - 56 pairs: the merged launch on 4 - 8 workgroups per pair)
def test_replicated_cache_read_one_copy_is_bit_identical(ops, f16, bsz, lens, kv_heads, copies, page):
    """ops.kv_batch_decode(kv_copies=g) on a cache whose g copies per KV head are identical == the launch that reads every head's own copy:
    bit for bit where the launch geometry is the same (split launches, a workgroup per query head); from 32 (request, KV head) pairs on ONE
    workgroup serves the group (other wave count, other order of the fp32 additions — as for split launches): 1e-3 of the output's maximum.
    INT4 and fp16 pages."""
    g = torch.Generator(device="cuda").manual_seed(bsz + kv_heads)
    hd, layer = 128, 1
    data, param, indptr, indices, last = _replicated(g, bsz, lens, kv_heads, copies, page, f16)
    heads = kv_heads * copies
    q = (torch.randn(bsz, heads, hd, generator=g, device="cuda") * 0.5).half()
    qt = (torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5).half()
    for tr in (False, True):
        o1 = ops.kv_batch_decode(q, data, param, indptr, indices, last, layer, qt, tr, seq_hint=max(lens))
        o2 = ops.kv_batch_decode(q, data, param, indptr, indices, last, layer, qt, tr, seq_hint=max(lens), kv_copies=copies)
        if copies in (2, 4, 8) and bsz * kv_heads >= 32:
            assert ((o1.float() - o2.float()).abs().amax() / o1.float().abs().amax()).item() <= 1e-3
        else:
            assert torch.equal(o1, o2)
    with pytest.raises(ValueError):
        ops.kv_batch_decode(q[:, :kv_heads].contiguous(), data, param, indptr, indices, last, layer, kv_copies=copies)   # q must carry one head per cache head


@pytest.mark.parametrize("bsz,lens,kv_heads,copies,page", [(1, [2048], 8, 4, 2048), (4, [31, 32, 33, 48], 2, 2, 16), (16, [900 + i for i in range(16)], 8, 4, 64),
                                                           (64, [100 + i for i in range(64)], 8, 4, 32), (3, [70, 300, 33], 8, 8, 16), (16, [260 + i for i in range(16)], 8, 8, 32),
                                                           (4, [2048, 2047, 1500, 2049], 8, 4, 2048), (6, [1000 + 17 * i for i in range(6)], 8, 4, 64)])   # (the merged launch split over 5 - 8 workgroups per pair)
def test_decode_append_read_one_copy_writes_every_copy(ops, bsz, lens, kv_heads, copies, page):
    """cache bytes and parameters of EVERY copy bit for bit; the output bit for bit unless one workgroup serves the group (see above)"""
    g = torch.Generator(device="cuda").manual_seed(bsz * 7 + kv_heads)
    hd, layer = 128, 0
    data, param, indptr, indices, last = _replicated(g, bsz, lens, kv_heads, copies, page)
    heads = kv_heads * copies
    k = (torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda") * 2).half()
    v = (torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda") * 2).half()
    T = (torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5).half()
    q = (torch.randn(bsz, heads, hd, generator=g, device="cuda") * 0.5).half()
    d1, p1 = data.clone(), param.clone()
    ops.kv_quant_append(k, v, T, d1, p1, indptr, indices, last, layer, copies)
    o1 = ops.kv_batch_decode(q, d1, p1, indptr, indices, last, layer, T, seq_hint=max(lens))
    d2, p2 = data.clone(), param.clone()
    o2 = ops.kv_decode_append(q, k.view(bsz, kv_heads, hd), v.view(bsz, kv_heads, hd), T, d2, p2, indptr, indices, last, layer, T, seq_hint=max(lens),
                              read_one_copy=True)
    assert torch.equal(d2, d1) and torch.equal(p2.view(torch.int16), p1.view(torch.int16))
    if bsz * kv_heads >= 32:
        assert ((o1.float() - o2.float()).abs().amax() / o1.float().abs().amax()).item() <= 1e-3
    else:
        assert torch.equal(o2, o1)


@pytest.mark.parametrize("disable_quant", [False, True])
def test_cache_class_read_one_copy_equals_reading_every_copy(ops, disable_quant):
    import flatquant_amd.deploy.transformers as dt
    g = torch.Generator(device="cuda").manual_seed(6)
    bsz, prompt, kv_heads, group, hd, page = 40, 37, 8, 4, 128, 16          # 320 (request, KV head) pairs: one workgroup per group
    heads = kv_heads * group
    tk = (torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5).half()
    kw = {"trans_matrix_k": tk, "trans_matrix_k_inv_t": tk}
    caches = [dt.MultiLayerPagedKVCache4Bit(bsz, page, prompt + 8, torch.device("cuda"), 1, heads, hd, trans="matmul", group_size=group,
                                            disable_quant=disable_quant, read_one_copy=r, fuse_append=r) for r in (True, False)]
    for c in caches:
        c.pages.zero_(), c.scales.zero_()
    k0 = torch.randn(bsz, prompt, kv_heads, hd, generator=g, device="cuda").half()
    v0 = torch.randn(bsz, prompt, kv_heads, hd, generator=g, device="cuda").half()
    for c in caches:
        c.update(k0, v0, 0, dict(kw))
    for step in range(4):
        k1 = torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda").half()
        v1 = torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda").half()
        q1 = torch.randn(bsz, 1, heads, hd, generator=g, device="cuda").half()
        o = [c.update(k1, v1, 0, dict(kw))(q1) for c in caches]
        assert ((o[0].float() - o[1].float()).abs().amax() / o[1].float().abs().amax()).item() <= 1e-3, step      # (320 pairs: one workgroup per group)
    used = caches[0].page_cnt_from_length(caches[0].length) * bsz
    assert torch.equal(caches[0].pages[:used], caches[1].pages[:used])
    assert torch.equal(caches[0].scales[:used].view(torch.int16), caches[1].scales[:used].view(torch.int16))


def test_cache_class_groups_of_eight(ops):
    """64 query heads on 8 KV heads (Llama-2/3-70B): the replicated layout (eight copies written, one read, fused append) against the two-launch,
    every-copy reading — same pages, outputs within the fp32-order tolerance; the shared cache agrees with both"""
    import flatquant_amd.deploy.transformers as dt
    g = torch.Generator(device="cuda").manual_seed(11)
    bsz, prompt, kv_heads, group, hd, page = 3, 50, 8, 8, 128, 16
    heads = kv_heads * group
    tk = (torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5).half()
    kw = {"trans_matrix_k": tk, "trans_matrix_k_inv_t": tk}
    mk = lambda **k: dt.MultiLayerPagedKVCache4Bit(bsz, page, prompt + 8, torch.device("cuda"), 1, heads, hd, trans="matmul", group_size=group, **k)
    caches = [mk(), mk(read_one_copy=False, fuse_append=False), mk(share_kv_heads=True)]
    for c in caches:
        c.pages.zero_(), c.scales.zero_()
    k0 = torch.randn(bsz, prompt, kv_heads, hd, generator=g, device="cuda").half()
    v0 = torch.randn(bsz, prompt, kv_heads, hd, generator=g, device="cuda").half()
    for c in caches:
        c.update(k0, v0, 0, dict(kw))
    for step in range(3):
        k1 = torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda").half()
        v1 = torch.randn(bsz, 1, kv_heads, hd, generator=g, device="cuda").half()
        q1 = torch.randn(bsz, 1, heads, hd, generator=g, device="cuda").half()
        o = [c.update(k1, v1, 0, dict(kw))(q1) for c in caches]
        assert torch.equal(o[0], o[1]) and torch.equal(o[0], o[2]), step          # (24 (request, KV head) pairs: a workgroup per query head everywhere)
    used = caches[0].page_cnt_from_length(caches[0].length) * bsz
    assert torch.equal(caches[0].pages[:used], caches[1].pages[:used])
    assert torch.equal(caches[0].scales[:used].view(torch.int16), caches[1].scales[:used].view(torch.int16))


def test_matrix_pipe_pv_against_fp64_dense_attention(ops):
    """the merged launch with p . v on the matrix pipe (fp16 weights p_i s_i) against an fp64 softmax attention over the de-quantised cache: within
    1e-3 of each output row's largest magnitude (measured 5.2 - 5.6e-4, of which 4.9e-4 is the fp16 output's half ulp: profiles/r06_pv_mfma.txt) —
    flat and peaked score distributions, ragged lengths, 4 waves and 8 waves per workgroup"""
    for bsz, qscale in ((40, 0.05), (40, 0.6), (70, 0.2)):
        g = torch.Generator(device="cuda").manual_seed(bsz)
        kv, copies, hd, page = 8, 4, 128, 64
        lens = [200 + 37 * (b % 7) for b in range(bsz)]
        data, param, indptr, indices, last = _replicated(g, bsz, lens, kv, copies, page)
        q = (torch.randn(bsz, kv * copies, hd, generator=g, device="cuda") * qscale).half()
        o = ops.kv_batch_decode(q, data, param, indptr, indices, last, 1, kv_copies=copies, seq_hint=max(lens)).double()
        # dense reference for a few requests
        n_pg = [(n + page - 1) // page for n in lens]
        starts = [0]
        for c in n_pg:
            starts.append(starts[-1] + c)
        worst = 0.0
        for b in (0, 3, bsz - 1):
            pages = indices[starts[b]:starts[b + 1]].long()
            n = lens[b]
            for h in (0, 5, 17, 31):
                ch = (h // copies) * copies
                kq = data[pages, 1, 0, ch].reshape(-1, hd // 2)[:n].to(torch.int32)
                vq = data[pages, 1, 1, ch].reshape(-1, hd // 2)[:n].to(torch.int32)
                kn = torch.stack([kq & 15, kq >> 4], -1).reshape(n, hd).double()
                vn = torch.stack([vq & 15, vq >> 4], -1).reshape(n, hd).double()
                kp = param[pages, 1, 0, ch].reshape(-1, 2)[:n].double()
                vp = param[pages, 1, 1, ch].reshape(-1, 2)[:n].double()
                K = kn * kp[:, :1] - kp[:, 1:]
                V = vn * vp[:, :1] - vp[:, 1:]
                ref = torch.softmax((K @ q[b, h].double()) / hd ** 0.5, 0) @ V
                worst = max(worst, ((o[b, h] - ref).abs().max() / ref.abs().max()).item())
        assert worst <= 1e-3, (bsz, qscale, worst)
