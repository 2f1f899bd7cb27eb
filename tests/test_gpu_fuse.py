"""deploy.fuse(model): the fused launches installed from the OUTSIDE on a layer with the reference's module structure and call order
(tools/ref_layer.py restates deploy/transformers/modeling_llama.py:66-78, 143-150, 236-280, 518-529). Everything fuse() installs is
bit-identical to the unfused modules, so the whole layer's outputs must be bit for bit the same."""
import os
import sys

import pytest
import torch

from flatquant_amd import ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def _groups(layer):
    a, m = layer.self_attn, layer.mlp
    return (a.inp_trans_q.__dict__["_group"], a.q_proj.__dict__.get("_group"), m.inp_trans_u.__dict__["_group"], m.up_proj.__dict__.get("_group"))


@pytest.mark.parametrize("down", ["matmul", "had"])
@pytest.mark.parametrize("bsz,seq", [(2, 160), (1, 4)])       # prefill-sized (the multi-problem GEMM route) and decode-sized
def test_fused_layer_is_bit_identical_to_the_reference_structure(down, bsz, seq):
    import flatquant_amd.deploy as deploy
    from ref_layer import RefLayer
    with torch.no_grad():
        layer = RefLayer("tiny", seed=3, down=down)
        g = torch.Generator(device="cuda").manual_seed(11)
        xs = [torch.randn(bsz, seq, 4096, generator=g, device="cuda", dtype=torch.float16) for _ in range(2)]
        want = [layer(x) for x in xs]
        rep = deploy.fuse(layer)
        assert rep == {"transform_groups": 2, "linear_groups": 2, "down_proj": 1}
        assert isinstance(layer.mlp.down_proj, deploy.nn.FusedSequential)
        assert list(layer.mlp.down_proj.state_dict().keys()) == list(torch.nn.Sequential(*layer.mlp.down_proj).state_dict().keys())
        assert deploy.fuse(layer) == {"transform_groups": 0, "linear_groups": 0, "down_proj": 0}      # idempotent
        tga, lga, tgm, lgm = _groups(layer)
        for i, x in enumerate(xs):
            got = layer(x)
            for a, b in zip(got, want[i]):
                assert torch.equal(a, b)
            n = i + 1 if bsz * seq > 128 else 0      # (decode-sized calls: the groups step aside, every module runs its own prepared call)
            assert tga.launches == n and tga.served == 2 * n                     # one transform launch for q / k / v
            assert tgm.launches == n and tgm.served == n                         # ... and one for up / gate
            assert lga.launches == n and lga.served == 2 * n and lgm.launches == n and lgm.served == n
            assert tga._outs is None and tgm._outs is None and lga._ys is None and lgm._ys is None   # nothing activation-sized is kept
        # a member called on its own with some other tensor simply runs (and does not disturb the group)
        alone = layer.self_attn.inp_trans_k(xs[0])
        deploy.unfuse(layer)
        ref = layer.self_attn.inp_trans_k(xs[0])
        assert torch.equal(alone.quantized_x, ref.quantized_x) and torch.equal(alone.scales_x, ref.scales_x)
        for a, b in zip(layer(xs[1]), want[1]):                                  # unfused again (down_proj stays fused: same results)
            assert torch.equal(a, b)


def test_groups_serve_decode_sized_calls_while_a_graph_is_captured():
    """Eager decode-sized calls leave the groups aside (Python-bound: the members' prepared calls are cheaper); under stream capture the
    groups serve them — the replayed graph holds one transform launch and one weight-streaming GEMM launch per group — and the replay is
    bit for bit the eager unfused layer."""
    import flatquant_amd.deploy as deploy
    from ref_layer import RefLayer
    with torch.no_grad():
        layer = RefLayer("tiny", seed=5)
        g = torch.Generator(device="cuda").manual_seed(13)
        x = torch.randn(3, 1, 4096, generator=g, device="cuda", dtype=torch.float16)
        want = layer(x)
        deploy.fuse(layer)
        tga, lga, tgm, lgm = _groups(layer)
        for a, b in zip(layer(x), want):                     # eager: the groups step aside
            assert torch.equal(a, b)
        assert tga.launches == 0 and lga.launches == 0
        xs = x.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            layer(xs)                                        # (warm-up on the side stream: images, workspaces)
            n0 = (tga.launches + tga.lazy, lga.launches, tgm.launches + tgm.lazy, lgm.launches)
            f0 = (tga.launches, lga.fused, tgm.launches, lgm.fused)
            with torch.cuda.graph(graph, stream=s):
                got = layer(xs)
        torch.cuda.synchronize()
        assert (tga.launches + tga.lazy, lga.launches, tgm.launches + tgm.lazy, lgm.launches) == tuple(v + 1 for v in n0)   # one fused launch per group, captured
        # (round 6) three tokens of d = 4096: the transform ran as the projections' PROLOGUE — no transform launch at all in the graph
        assert (tga.launches, lga.fused, tgm.launches, lgm.fused) == (f0[0], f0[1] + 1, f0[2], f0[3] + 1)
        assert tga.served >= 2 and lga.served >= 2
        graph.replay()
        torch.cuda.synchronize()
        for a, b in zip(got, want):
            assert torch.equal(a, b)
        x2 = torch.randn(3, 1, 4096, generator=g, device="cuda", dtype=torch.float16)
        want2 = [t.clone() for t in RefLayerEager(layer, x2)]
        xs.copy_(x2)
        graph.replay()
        torch.cuda.synchronize()
        for a, b in zip(got, want2):
            assert torch.equal(a, b)


def RefLayerEager(layer, x):
    """the fused layer called eagerly at a decode size (the groups step aside): the reference result for a replay with new inputs"""
    return layer(x)


def test_fuse_without_linear_groups_and_with_static_outputs():
    import flatquant_amd.deploy as deploy
    from ref_layer import RefLayer
    with torch.no_grad():
        layer = RefLayer("tiny", seed=5)
        x = torch.randn(2, 130, 4096, device="cuda", dtype=torch.float16)
        want = layer(x)
        assert deploy.fuse(layer, linears=False) == {"transform_groups": 2, "linear_groups": 0, "down_proj": 1}
        for a, b in zip(layer(x), want):
            assert torch.equal(a, b)
        assert layer.self_attn.inp_trans_q.__dict__["_group"]._outs is None       # dropped once q, k and v have theirs
        deploy.fuse(layer, static_outputs=True)
        for _ in range(2):
            for a, b in zip(layer(x), want):
                assert torch.equal(a, b)


def test_default_module_path_returns_fresh_outputs_and_follows_updates():
    """(round 5) the default forward of OnlineTrans / Quantizer / Linear4bit is a C-side prepared call (ops.FreshPlan): outputs are
    fresh tensors every call, results are those of the general entry points, and an in-place update of a matrix / clip factor, a new
    shape or Python-float clip factors (the reference's loader: modeling_llama.py:532-538) are followed."""
    import flatquant_amd.deploy as deploy
    g = torch.Generator(device="cuda").manual_seed(2)
    t = deploy.nn.OnlineTrans(4096, trans="matmul", lac=True).cuda()
    for n in ("left_matrix", "right_matrix"):
        t._buffers[n].copy_(torch.randn(64, 64, generator=g, device="cuda") / 8)
    t.clip_factor_a_max.fill_(3.0)
    x = torch.randn(2, 7, 4096, generator=g, device="cuda", dtype=torch.float16)
    a, b = t(x), t(x)
    assert a.quantized_x.data_ptr() != b.quantized_x.data_ptr() and torch.equal(a.quantized_x, b.quantized_x)
    assert a.quantized_x.shape == (2, 7, 2048) and a.scales_x.shape == (2, 1, 7)
    t.fast_path = False
    ref = t(x)
    t.fast_path = True
    assert torch.equal(a.quantized_x, ref.quantized_x) and torch.equal(a.scales_x, ref.scales_x)
    t.left_matrix.mul_(0.5)                                   # in-place update: re-planned
    t.fast_path = False
    ref2 = t(x)
    t.fast_path = True
    c = t(x)
    assert torch.equal(c.quantized_x, ref2.quantized_x) and not torch.equal(c.scales_x, a.scales_x)
    x2 = torch.randn(1, 3, 4096, generator=g, device="cuda", dtype=torch.float16)   # another shape
    assert t(x2).quantized_x.shape == (1, 3, 2048)
    st = t.__dict__["_fresh_state"]                           # alternating shapes (prefill / decode) keep BOTH plans
    plans = dict(st.plans)
    assert len(plans) == 2
    for _ in range(3):
        assert torch.equal(t(x).quantized_x, c.quantized_x) and t(x2).quantized_x.shape == (1, 3, 2048)
    assert all(st.plans[k] is v for k, v in plans.items()) and len(st.plans) == 2
    for k in range(4, 4 + 2 * ops.FreshPlanSet.KEEP):        # bounded
        t(torch.zeros(1, k, 4096, device="cuda", dtype=torch.float16))
    assert len(st.plans) == ops.FreshPlanSet.KEEP
    for name in ("clip_factor_a_max", "clip_factor_a_min"):  # the loader's floats
        v = getattr(t, name).item()
        delattr(t, name)
        setattr(t, name, v)
    d = t(x)
    assert torch.equal(d.quantized_x, c.quantized_x)
    # Quantizer, both branches, and the decode-sized Linear4bit
    for lac in (True, False):
        qz = deploy.nn.Quantizer(input_clip_ratio=0.9, lac=lac).cuda()
        y = torch.randn(5, 9, 14336, generator=g, device="cuda", dtype=torch.float16)
        p, p2 = qz(y), qz(y)
        qz.fast_path = False
        r = qz(y)
        assert p.quantized_x.data_ptr() != p2.quantized_x.data_ptr()
        assert torch.equal(p.quantized_x, r.quantized_x) and torch.equal(p.scales_x, r.scales_x) and p.scales_x.shape == r.scales_x.shape
    lin = deploy.nn.Linear4bit(4096, 1024).cuda()
    lin.weight.copy_(torch.randint(0, 256, lin.weight.shape, generator=g, device="cuda", dtype=torch.uint8))
    lin.weight_scales.fill_(0.01)
    pk = deploy.PackedQuantizedTensor(a.quantized_x[:, :4].contiguous(), a.scales_x[:, :, :4].contiguous())
    y1, y2 = lin(pk), lin(pk)
    lin.fast_path = False
    y0 = lin(pk)
    assert y1.data_ptr() != y2.data_ptr() and torch.equal(y1, y0) and torch.equal(y2, y0) and y1.shape == (2, 4, 1024)


@pytest.mark.parametrize("bsz,seq", [(2, 160), (1, 4)])
def test_modules_built_and_called_under_inference_mode(bsz, seq):
    """the reference decorates its generation / benchmark entry points with @torch.inference_mode() and
    benchmarks/qlinear_benchmark.py builds its modules inside one — inference tensors have no version counter (``t._version`` raises).
    Modules built INSIDE inference mode, called there, fused there: the bits of the same layer under no_grad."""
    import flatquant_amd.deploy as deploy
    from ref_layer import RefLayer
    g = torch.Generator(device="cuda").manual_seed(21)
    x = torch.randn(bsz, seq, 4096, generator=g, device="cuda", dtype=torch.float16)
    with torch.no_grad():
        want = RefLayer("tiny", seed=7)(x)
    with torch.inference_mode():
        layer = RefLayer("tiny", seed=7)
        assert layer.self_attn.q_proj.weight.is_inference()
        xi = x.clone()
        assert xi.is_inference()
        for a, b in zip(layer(xi), want):                     # the default prepared-call paths (Quantizer._fresh, OnlineTrans._fresh, Linear4bit plans)
            assert torch.equal(a, b)
        deploy.fuse(layer)
        for _ in range(2):
            for a, b in zip(layer(xi), want):                 # TransformGroup.get on an inference activation
                assert torch.equal(a, b)
        q = deploy.nn.Quantizer(lac=True).to("cuda")
        p = q(xi.reshape(-1, 4096))
        assert p.quantized_x.shape == (bsz * seq, 2048)


def test_lazy_group_results_materialise_when_somebody_else_asks():
    """Under capture a decode-sized transform group hands out LAZY packed tensors (the projections' launch carries the transform); a caller
    that reads the packed bytes itself gets them — the group's transform launch runs then, once — and the projections still agree."""
    import flatquant_amd.deploy as deploy
    from ref_layer import RefLayer
    with torch.no_grad():
        layer = RefLayer("tiny", seed=7)
        g = torch.Generator(device="cuda").manual_seed(17)
        x = torch.randn(2, 1, 4096, generator=g, device="cuda", dtype=torch.float16)
        a = layer.self_attn
        want_p = [t(x) for t in (a.inp_trans_q, a.inp_trans_k, a.inp_trans_v)]
        want_y = [l(p) for l, p in zip((a.q_proj, a.k_proj, a.v_proj), want_p)]
        deploy.fuse(layer)
        tga, lga, _, _ = _groups(layer)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            [l(t(x)) for t, l in zip((a.inp_trans_q, a.inp_trans_k, a.inp_trans_v), (a.q_proj, a.k_proj, a.v_proj))]   # warm-up (eager: own calls)
            with torch.cuda.graph(graph, stream=s):
                pq, pk, pv = a.inp_trans_q(x), a.inp_trans_k(x), a.inp_trans_v(x)
                assert type(pq).__name__ == "LazyPackedQuantizedTensor" and isinstance(pq, deploy.PackedQuantizedTensor)
                assert tuple(pq.size()) == (2, 1, 2048) and tga.launches == 0
                kq = pk.quantized_x                                   # somebody else asks: the transform launch runs now
                assert tga.launches == 1
                ys = [a.q_proj(pq), a.k_proj(pk), a.v_proj(pv)]       # ... and the projections take the materialised tensors
                assert lga.fused == 0 and lga.launches == 1
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(kq, want_p[1].quantized_x)
        for y, w in zip(ys, want_y):
            assert torch.equal(y, w)
