"""deploy.fuse(model, capture=True) / deploy.GraphedDecode: a decoder layer's decode step served from captured HIP graphs with no graph code
on the caller's side (the reference's eager decode loop: deploy/transformers/modeling_llama.py:45-153,
benchmarks/layer_benchmark.py:131-143). The replayed step must be the eager step bit for bit, token after token, with the paged INT4 cache
really growing: two identical caches, one driven eagerly, one through the wrapper."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def build(bsz, page, prompt, max_len, seed=0):
    import flatquant_amd.deploy as deploy
    import flatquant_amd.deploy.transformers as dt
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(seed)
    hidden, heads, kv_heads, hd = 4096, 32, 8, 128

    def trans(dim, decompose=True):
        t = deploy.nn.OnlineTrans(dim, trans="matmul", decompose=decompose, lac=True).to(dev)
        for name in ("left_matrix", "right_matrix"):
            if name in t._buffers:
                b = t._buffers[name]
                b.copy_(torch.randn(b.shape, generator=g, device=dev) / b.shape[0] ** 0.5)
        t.clip_factor_a_max.fill_(4.0), t.clip_factor_a_min.fill_(3.5)
        return t

    def lin(k_in, n_out):
        m = deploy.nn.Linear4bit(k_in, n_out).to(dev)
        m.weight.copy_(torch.randint(0, 256, m.weight.shape, generator=g, device=dev, dtype=torch.uint8))
        m.weight_scales.copy_(torch.rand(m.weight_scales.shape, generator=g, device=dev) * 0.02 + 0.005)
        return m

    class Attn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.inp_trans_q = trans(hidden)
            self.q_proj, self.k_proj, self.v_proj, self.o_proj = lin(hidden, hidden), lin(hidden, kv_heads * hd), lin(hidden, kv_heads * hd), lin(hidden, hidden)
            self.o_proj_trans = trans(heads, decompose=False)
            self.register_buffer("tk", (torch.randn(hd, hd, generator=g, device=dev) / hd ** 0.5).half())

        def forward(self, h, cache):
            b = h.shape[0]
            p = self.inp_trans_q(h)
            q, k, v = self.q_proj(p), self.k_proj(p), self.v_proj(p)
            kw = {"trans_matrix_k": self.tk, "trans_matrix_k_inv_t": self.tk}
            attend = cache.update(k.view(b, 1, kv_heads, hd), v.view(b, 1, kv_heads, hd), 0, kw)
            po = self.o_proj_trans(attend(q.view(b, 1, heads, hd), transposed=True))
            po.quantized_x = po.quantized_x.contiguous().reshape(b, 1, -1)
            return self.o_proj(po)

    class Mlp(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.inp_trans_u = trans(hidden)
            self.up_proj = lin(hidden, 1024)

        def forward(self, h):
            return self.up_proj(self.inp_trans_u(h))

    class Layer(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.self_attn, self.mlp = Attn(), Mlp()

        def forward(self, h, cache, scale=1.0):
            a = self.self_attn(h, cache)
            return h + a * scale, self.mlp(a)

    def new_cache():
        c = dt.MultiLayerPagedKVCache4Bit(bsz, page, max_len, dev, 1, heads, hd, trans="matmul", group_size=heads // kv_heads)
        gk = torch.Generator(device=dev).manual_seed(99)
        c.update(torch.randn(bsz, prompt, kv_heads, hd, generator=gk, device=dev).half(),
                 torch.randn(bsz, prompt, kv_heads, hd, generator=gk, device=dev).half(), 0,
                 {"trans_matrix_k": layer.self_attn.tk, "trans_matrix_k_inv_t": layer.self_attn.tk})
        return c

    layer = Layer()
    return layer, new_cache, g


@pytest.mark.parametrize("bsz", [1, 3])
def test_graphed_layer_replays_the_eager_step_token_after_token(bsz):
    import flatquant_amd.deploy as deploy
    with torch.no_grad():
        layer, new_cache, g = build(bsz, page=16, prompt=21, max_len=64)
        xs = [torch.randn(bsz, 1, 4096, generator=g, device="cuda", dtype=torch.float16) for _ in range(14)]
        ref_cache = new_cache()
        want = [tuple(t.clone() for t in layer(x, ref_cache)) for x in xs]           # eager, cache growing 21 -> 35 (crosses a page at 32)
        rep = deploy.fuse(layer, capture=True)
        assert rep["graphed_layers"] == 1
        gd = layer.__dict__["forward"]
        cache = new_cache()
        for i, x in enumerate(xs):
            got = layer(x, cache)
            assert cache.length == 21 + i + 1
            for a, b in zip(got, want[i]):
                assert torch.equal(a, b), i
        assert gd.eager_calls == 2 and gd.captures == 1 and gd.replays == 12        # two warm-up calls, then every step from the graph
        for pg in range(cache.page_cnt_from_length(cache.length)):                   # the rows the steps wrote (the pages are torch.empty beyond them)
            rows = min(cache.page_size, cache.length - pg * cache.page_size)
            for b in range(bsz):
                i = pg * bsz + b
                assert torch.equal(cache.pages[i, :, :, :, :rows], ref_cache.pages[i, :, :, :, :rows])
                assert torch.equal(cache.scales[i, :, :, :, :rows], ref_cache.scales[i, :, :, :, :rows])
        # another signature (a different python scalar argument) gets its own warm-up + graph; prefill-sized calls stay eager
        c2, c3 = new_cache(), new_cache()
        for x in xs[:4]:
            a = layer(x, c2, scale=0.5)
        deploy.unfuse(layer)
        for x in xs[:4]:
            b = layer(x, c3, scale=0.5)
        assert all(torch.equal(u, v) for u, v in zip(a, b))
        assert "forward" not in layer.__dict__


def test_page_growth_makes_the_graph_stale_and_it_is_recaptured():
    """max_seq_len 32 -> two pages per request allocated; the 33rd token re-allocates the pages (every pointer moves): the wrapper sees it
    coming (would_grow), runs that step eagerly and captures again over the new storage."""
    import flatquant_amd.deploy as deploy
    with torch.no_grad():
        layer, new_cache, g = build(2, page=16, prompt=26, max_len=32, seed=3)
        xs = [torch.randn(2, 1, 4096, generator=g, device="cuda", dtype=torch.float16) for _ in range(12)]
        ref_cache = new_cache()
        want = [tuple(t.clone() for t in layer(x, ref_cache)) for x in xs]           # 26 -> 38: grows at 33
        gd = deploy.GraphedDecode(layer.forward)
        cache = new_cache()
        gen0 = cache.generation
        for i, x in enumerate(xs):
            got = gd(x, cache)
            for a, b in zip(got, want[i]):
                assert torch.equal(a, b), i
        assert cache.generation > gen0 and gd.captures == 2 and cache.length == 38
        assert gd.eager_calls == 3                                                   # two warm-ups + the step that grew the pages


def test_calls_that_are_not_decode_steps_stay_eager():
    import flatquant_amd.deploy as deploy
    layer, new_cache, g = build(1, page=16, prompt=5, max_len=64, seed=5)
    gd = deploy.GraphedDecode(layer.mlp.forward)
    with torch.no_grad():
        big = torch.randn(1, 200, 4096, generator=g, device="cuda", dtype=torch.float16)
        for _ in range(4):
            gd(big)                                                                  # prefill-sized
        assert gd.captures == 0 and gd.replays == 0
        x = torch.randn(1, 1, 4096, generator=g, device="cuda", dtype=torch.float16)
        outs = [gd(x).clone() for _ in range(5)]
        assert gd.captures == 1 and gd.replays == 3 and all(torch.equal(o, outs[0]) for o in outs)
    xg = torch.randn(1, 1, 4096, device="cuda", dtype=torch.float16)
    n = gd.replays
    with torch.enable_grad():
        gd(xg)                                                                       # autograd on: eager
    assert gd.replays == n


def test_tensors_inside_containers_are_graph_inputs_and_changing_signatures_stay_bounded():
    """HF passes ``position_embeddings=(cos, sin)``: tensors inside tuples / dicts are copied into static inputs like top-level ones; a caller
    whose signature changes on every call (a growing mask) never captures, stays correct and does not grow the wrapper's tables."""
    import flatquant_amd.deploy as deploy

    def fn(h, pe=None, extra=None):
        cos, sin = pe
        return h * cos + sin + (0 if extra is None else extra["b"])

    gd = deploy.GraphedDecode(fn)
    g = torch.Generator(device="cuda").manual_seed(1)
    with torch.no_grad():
        for i in range(6):
            h, c, s_, b = (torch.randn(2, 1, 64, generator=g, device="cuda", dtype=torch.float16) for _ in range(4))
            got = gd(h, pe=(c, s_), extra={"b": b})
            assert torch.equal(got, fn(h, pe=(c, s_), extra={"b": b})), i
        assert gd.captures == 1 and gd.replays == 4
        gd.MAX_SIGNATURES = 8
        for i in range(40):                                   # a new shape every call: warm-up forever, tables bounded
            m = torch.ones(2, 1, 3 + i, device="cuda", dtype=torch.float16)
            assert gd(m, pe=(m, m)).shape == m.shape
        assert gd.captures == 1 and len(gd._seen) <= 8 and len(gd._logs) <= 8 and len(gd._entries) <= 8


def test_a_callable_that_cannot_be_captured_falls_back_to_eager_for_good():
    """An op that synchronises inside the callable makes stream capture fail: the wrapper runs that call eagerly, remembers the signature and never
    tries again — results stay right, nothing raises."""
    import flatquant_amd.deploy as deploy

    def fn(h):
        return h * float(h.abs().max().item())          # .item(): a device synchronisation, illegal under capture

    gd = deploy.GraphedDecode(fn)
    g = torch.Generator(device="cuda").manual_seed(2)
    with torch.no_grad():
        for i in range(6):
            h = torch.randn(1, 1, 64, generator=g, device="cuda", dtype=torch.float16)
            assert torch.equal(gd(h), fn(h)), i
    assert gd.captures == 0 and gd.replays == 0 and len(gd._blocked) == 1
    # the wrapper still captures other callables' signatures afterwards (the failed capture left no capture state behind)
    gd2 = deploy.GraphedDecode(lambda h: h + 1)
    with torch.no_grad():
        outs = [gd2(torch.ones(1, 1, 8, device="cuda", dtype=torch.float16)).clone() for _ in range(4)]
    assert gd2.captures == 1 and all(torch.equal(o, outs[0]) for o in outs)


def test_hf_shaped_layer_call_with_keyword_arguments():
    """The call shape of the reference's decoder layer (deploy/transformers/modeling_llama.py:45-153 behind HF's LlamaDecoderLayer.forward):
    everything by keyword — position ids and cache position as tensors that change every step, (cos, sin) as a tuple, the cache as
    ``past_key_value``, flags as Python bools, a tuple coming back — through deploy.fuse(model, capture=True), against the eager layer."""
    import flatquant_amd.deploy as deploy
    with torch.no_grad():
        inner, new_cache, g = build(2, page=16, prompt=9, max_len=64, seed=8)

        class HFLayer(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.self_attn, self.mlp = inner.self_attn, inner.mlp

            def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None, output_attentions=False,
                        use_cache=False, cache_position=None, position_embeddings=None):
                cos, sin = position_embeddings
                h = hidden_states * cos + sin + position_ids.to(hidden_states.dtype).unsqueeze(-1) * 0.001
                a = self.self_attn(h, past_key_value)
                out = (hidden_states + a + cache_position.to(a.dtype).reshape(1, 1, 1) * 0.01,)
                return out + ((self.mlp(a),) if use_cache else ())

        layer = HFLayer()
        steps = []
        for t in range(8):
            steps.append(dict(hidden_states=torch.randn(2, 1, 4096, generator=g, device="cuda", dtype=torch.float16), attention_mask=None,
                              position_ids=torch.full((2, 1), 9 + t, device="cuda"), output_attentions=False, use_cache=True,
                              cache_position=torch.tensor([9 + t], device="cuda"),
                              position_embeddings=(torch.rand(2, 1, 4096, generator=g, device="cuda", dtype=torch.float16),
                                                   torch.rand(2, 1, 4096, generator=g, device="cuda", dtype=torch.float16))))
        ref_cache = new_cache()
        want = [tuple(o.clone() for o in layer(past_key_value=ref_cache, **kw)) for kw in steps]
        assert deploy.fuse(layer, capture=True)["graphed_layers"] == 1
        cache = new_cache()
        for t, kw in enumerate(steps):
            got = layer(past_key_value=cache, **kw)
            assert isinstance(got, tuple) and len(got) == 2
            for a, b in zip(got, want[t]):
                assert torch.equal(a, b), t
        gd = layer.__dict__["forward"]
        assert gd.captures == 1 and gd.replays == 6 and cache.length == ref_cache.length == 17
