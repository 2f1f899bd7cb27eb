#!/usr/bin/env python3
"""One decoder layer's quantised decode step (one new token per request) through flatquant_amd's deploy modules, replayed
from a captured HIP graph (from Python every launch costs ~10 us of allocation + ctypes, more than most of these kernels):
norm + q/k/v transform (one launch) -> three Linear4bit (weight-streaming kernel) -> K transform + K/V INT4 pack ->
paged-cache append -> INT4 decode attention -> o_proj head transform -> Linear4bit -> norm + up/gate transform ->
two Linear4bit -> SiLU.mul + down_proj transform -> Linear4bit.  RoPE and the residual adds are left out (torch ops of
the host model). Llama-3-8B shapes, random weights."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flatquant_amd.deploy as deploy  # noqa: E402
import flatquant_amd.deploy.transformers as dt  # noqa: E402
from flatquant_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bsz", type=int, default=16)
    ap.add_argument("--cache", type=int, default=2048, help="cached tokens per request")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--two-launches", action="store_true", help="round 5's launch structure: transform launch + projection launch per group (no fused prologue)")
    ap.add_argument("--share-kv", action="store_true", help="the cache holds the KV heads once (share_kv_heads=True: 1 / group of the memory and of the rows a step reads)")
    ap.add_argument("--layers", type=int, default=1, help="L decoder layers with their OWN weights and cache layers in one step (the time printed is per layer). One "
                                                          "layer's INT4 weights are 109 MB and stay in the 256 MB Infinity Cache from one replay to the next; with L >= 4 "
                                                          "(436 MB; fp16: 1.7 GB) every replay streams them from HBM, as a 32-layer model's step does")
    ap.add_argument("--no-fuse-append", action="store_true", help="K transform + K / V quantise + append as a launch of its own in front of the attention (rounds 1-6) instead of inside it")
    ap.add_argument("--no-one-copy", action="store_true", help="the replicated cache read head by head (every query head its own copy of the rows: rounds 1-6) instead of one copy per KV head")
    ap.add_argument("--single", action="store_true", help="one launch per projection (rounds 1-4) instead of the multi-problem launches")
    ap.add_argument("--fp16", action="store_true", help="also time the same step in fp16 (nn.Linear, rms_norm, SiLU.mul, the fp16 configuration of the paged cache) "
                                                        "— the baseline of the reference's decode table, README.md:300-310")
    a = ap.parse_args()
    hidden, ffn, heads, kv_heads, hd = 4096, 14336, 32, 8, 128
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)

    def trans(dim, decompose=True):
        t = deploy.nn.OnlineTrans(dim, trans="matmul", decompose=decompose, lac=True).to(dev)
        for name in ("left_matrix", "right_matrix"):
            if hasattr(t, name):
                b = getattr(t, name)
                b.copy_(torch.randn(b.shape, generator=g, device=dev) / b.shape[0] ** 0.5)
        t.clip_factor_a_max.fill_(4.0), t.clip_factor_a_min.fill_(4.0)
        return t

    def share(ts):
        for t in ts[1:]:
            for name in ("left_matrix", "right_matrix"):
                del t._buffers[name]
                t.register_buffer(name, getattr(ts[0], name))
        return ts

    def lin(k_in, n_out):
        m = deploy.nn.Linear4bit(k_in, n_out).to(dev)
        m.weight_scales.fill_(0.01)
        return m

    def make_layer():
        qkv_t, ug_t = share([trans(hidden) for _ in range(3)]), share([trans(hidden) for _ in range(2)])
        o_t, down_t = trans(heads, decompose=False), trans(ffn)
        q_l, k_l, v_l, o_l = lin(hidden, hidden), lin(hidden, kv_heads * hd), lin(hidden, kv_heads * hd), lin(hidden, hidden)
        up_l, gate_l, down_l = lin(hidden, ffn), lin(hidden, ffn), lin(ffn, hidden)
        return qkv_t, ug_t, o_t, down_t, q_l, k_l, v_l, o_l, up_l, gate_l, down_l

    norm = deploy.nn.RMSNorm(hidden)
    L = max(1, a.layers)
    layers = [make_layer() for _ in range(L)]
    qkv_t, ug_t, o_t, down_t, q_l, k_l, v_l, o_l, up_l, gate_l, down_l = layers[0]
    tk = (torch.randn(hd, hd, generator=g, device=dev) / hd ** 0.5).half()
    cache = dt.MultiLayerPagedKVCache4Bit(a.bsz, 2048, a.cache + 8, dev, L, heads, hd, trans="matmul", group_size=heads // kv_heads, share_kv_heads=a.share_kv, fuse_append=not a.no_fuse_append, read_one_copy=not a.no_one_copy)
    kw = {"trans_matrix_k": tk, "trans_matrix_k_inv_t": tk}
    for li in range(L):
        cache.update(torch.randn(a.bsz, a.cache, kv_heads, hd, generator=g, device=dev).half(),
                     torch.randn(a.bsz, a.cache, kv_heads, hd, generator=g, device=dev).half(), li, dict(kw))
    x = torch.randn(a.bsz, 1, hidden, generator=g, device=dev).half()

    def layer_step(h, li):
        qkv_t, ug_t, o_t, down_t, q_l, k_l, v_l, o_l, up_l, gate_l, down_l = layers[li]
        if a.single or a.two_launches:
            pq, pk, pv = deploy.nn.fused_forward(h, qkv_t, norm=norm)
            q, k, v = deploy.nn.linear.linear4bit_multi([q_l, k_l, v_l], [pq, pk, pv]) if not a.single else (q_l(pq), k_l(pk), v_l(pv))
        else:   # (round 6) the transform as the projections' prologue where that pays (<= 8 tokens), the two launches otherwise
            q, k, v = deploy.nn.fused_transform_linear(h, qkv_t, [q_l, k_l, v_l], norm=norm)
        attend = cache.update(k.view(a.bsz, 1, kv_heads, hd), v.view(a.bsz, 1, kv_heads, hd), li, dict(kw))
        att_t = attend(q.view(a.bsz, 1, heads, hd), transposed=True)                # [bsz, 1, hd, heads]
        po = o_t(att_t)
        po.quantized_x = po.quantized_x.contiguous().reshape(a.bsz, 1, -1)
        h2 = o_l(po)
        if a.single or a.two_launches:
            pu, pg = deploy.nn.fused_forward(h2, ug_t, norm=norm)
            if a.single:
                return down_l(down_t(gate_l(pg), up=up_l(pu)))
            yu, yg = deploy.nn.linear.linear4bit_multi([up_l, gate_l], [pu, pg])
        else:
            yu, yg = deploy.nn.fused_transform_linear(h2, ug_t, [up_l, gate_l], norm=norm)
        return down_l(down_t(yg, up=yu))

    def step(h):
        for li in range(L):
            h = layer_step(h, li)
        return h

    class Layer(torch.nn.Module):
        """the same step as a decoder LAYER (a module with self_attn / mlp children, called as layer(h, cache)): what deploy.fuse(model,
        capture=True) wraps — the caller writes no graph code and the cache really grows by one token per call"""

        def __init__(self):
            super().__init__()
            self.self_attn = torch.nn.ModuleList([m for ly in layers for m in (*ly[0], ly[4], ly[5], ly[6], ly[2], ly[7])])
            self.mlp = torch.nn.ModuleList([m for ly in layers for m in (*ly[1], ly[8], ly[9], ly[3], ly[10])])

        def forward(self, h, cache):
            for li in range(L):
                qkv_t, ug_t, o_t, down_t, q_l, k_l, v_l, o_l, up_l, gate_l, down_l = layers[li]
                q, k, v = deploy.nn.fused_transform_linear(h, qkv_t, [q_l, k_l, v_l], norm=norm)
                attend = cache.update(k.view(a.bsz, 1, kv_heads, hd), v.view(a.bsz, 1, kv_heads, hd), li, dict(kw))
                po = o_t(attend(q.view(a.bsz, 1, heads, hd), transposed=True))
                po.quantized_x = po.quantized_x.contiguous().reshape(a.bsz, 1, -1)
                yu, yg = deploy.nn.fused_transform_linear(o_l(po), ug_t, [up_l, gate_l], norm=norm)
                h = down_l(down_t(yg, up=yu))
            return h

    def time_calls(fn, cache, n):
        """us per call of fn() launched from Python, the cache growing by one token per call (rewound afterwards): the MEDIAN of ten windows of
        n / 10 calls — one host stall (a 40 ms pause around the hundredth call of the fp16 step, seen at every batch size: a Python GC pass) inside
        a single window of 100 calls once read 500 us per call for a 165 us step"""
        l0 = cache.length
        for _ in range(12):      # (two eager warm-up calls, the capture, the first replays)
            fn()
        torch.cuda.synchronize()
        per, wins = max(1, n // 10), []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(per):
                fn()
            e1.record()
            torch.cuda.synchronize()
            wins.append(e0.elapsed_time(e1) / per * 1e3)
        cache.length = l0
        return sorted(wins)[len(wins) // 2]

    def measure(step, cache):
        """-> (us per step from a captured graph, us per step launched eagerly)"""
        for _ in range(3):                      # eager warm-up (weight images, workspaces, scalar caches), rewinding the cache length each time
            step(x)
            cache.length -= 1
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        wins = []
        for _ in range(5):          # (the median of five windows of four steps: see the INT4 step's eager timing below)
            e0.record()
            for _ in range(4):
                step(x)
                cache.length -= 1
            e1.record()
            torch.cuda.synchronize()
            wins.append(e0.elapsed_time(e1) / 4 * 1e3)
        eager = sorted(wins)[2]
        graph = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step(x)
            cache.length -= 1
            with torch.cuda.graph(graph, stream=s):
                step(x)
            cache.length -= 1
        torch.cuda.synchronize()
        for _ in range(10):
            graph.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters * 1e3 / L, eager / L

    if a.fp16:
        import torch.nn.functional as F
        mk = lambda i, o: torch.nn.Linear(i, o, bias=False, device=dev, dtype=torch.float16)
        with torch.no_grad():
            f16 = [(mk(hidden, hidden), mk(hidden, kv_heads * hd), mk(hidden, kv_heads * hd), mk(hidden, hidden),
                    mk(hidden, ffn), mk(hidden, ffn), mk(ffn, hidden)) for _ in range(L)]
        w1 = torch.ones(hidden, device=dev, dtype=torch.float16)
        cache16 = dt.MultiLayerPagedKVCache4Bit(a.bsz, 2048, a.cache + 8, dev, L, heads, hd, disable_quant=True, trans="none", group_size=heads // kv_heads, share_kv_heads=a.share_kv, read_one_copy=not a.no_one_copy)
        for li in range(L):
            cache16.update(torch.randn(a.bsz, a.cache, kv_heads, hd, generator=g, device=dev).half(),
                           torch.randn(a.bsz, a.cache, kv_heads, hd, generator=g, device=dev).half(), li, {})

        @torch.no_grad()
        def step16(h):
            for li in range(L):
                fq, fk, fv, fo, fu, fg, fd = f16[li]
                xn = F.rms_norm(h, (hidden,), w1, 1e-6)
                q, k, v = fq(xn), fk(xn), fv(xn)
                att = cache16.update(k.view(a.bsz, 1, kv_heads, hd), v.view(a.bsz, 1, kv_heads, hd), li, {})(q.view(a.bsz, 1, heads, hd))
                h2 = fo(att.reshape(a.bsz, 1, hidden))
                xn2 = F.rms_norm(h2, (hidden,), w1, 1e-6)
                h = fd(F.silu(fg(xn2)) * fu(xn2))
            return h
        us16, eager16 = measure(step16, cache16)
        g16 = deploy.GraphedDecode(lambda h, c: step16(h))        # the fp16 step through the SAME transparent helper (its cache passed so that
        with torch.no_grad():                                      # the host steps are recorded): what the baseline gains from it
            trans16 = time_calls(lambda: g16(x, cache16), cache16, min(a.iters, 100)) / L
        g16_counts = (g16.captures, g16.replays, g16.eager_calls)
        del g16, f16, cache16
        torch.cuda.empty_cache()

    # eager warm-up (fills the weight-image / workspace / scalar caches), rewinding the cache length each time
    for _ in range(3):
        y = step(x)
        cache.length -= 1
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    wins = []
    for _ in range(5):          # the median of five windows of four steps: one host pause (a GC pass: tens of ms) inside a single window of 20 read 3 ms per step
        e0.record()
        for _ in range(4):
            step(x)
            cache.length -= 1
        e1.record()
        torch.cuda.synchronize()
        wins.append(e0.elapsed_time(e1) / 4 * 1e3 / L)
    eager = sorted(wins)[2]
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step(x)
        cache.length -= 1
        with torch.cuda.graph(graph, stream=s):
            y = step(x)
        cache.length -= 1
    torch.cuda.synchronize()
    for _ in range(10):
        graph.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.iters):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3 / L
    # (round 6) no caller-side graph code: the layer as a module, deploy.fuse(layer, capture=True), then plain calls from Python
    layer = Layer()
    with torch.no_grad():
        rep = deploy.fuse(layer, capture=True)
        y2 = layer(x, cache)
        transparent = time_calls(lambda: layer(x, cache), cache, min(a.iters, 100)) / L
        gd = layer.__dict__["forward"]
    print(f"   deploy.fuse(layer, capture=True) {rep}: {transparent:.1f} us per call from Python, no graph code on the caller's side "
          f"({gd.captures} capture(s), {gd.replays} replays, {gd.eager_calls} eager warm-up calls); output finite: {bool(torch.isfinite(y2.float()).all())}"
          + (f"   | fp16 step through the same helper: {trans16:.1f} us (captures / replays / eager {g16_counts}) -> {trans16 / transparent:.2f}x; against the EAGER fp16 step "
             f"({eager16:.1f} us): {eager16 / transparent:.2f}x" if a.fp16 else ""))
    print(f"Llama-3-8B decoder layer, decode step, {a.bsz} requests x {a.cache} cached tokens"
          + (f" [{L} layers with their own weights per step: {L * 109} MB of INT4 weights streamed per replay]" if L > 1 else "") + f": {us:.1f} us per layer from a "
          f"captured graph ({eager:.1f} us launched eagerly from Python); output finite: {bool(torch.isfinite(y.float()).all())}"
          + (f"   | the same step in fp16 (fp16 paged cache): {us16:.1f} us captured ({eager16:.1f} eager): speed-up {us16 / us:.2f}x captured, "
             f"{eager16 / eager:.2f}x eager" if a.fp16 else ""))


if __name__ == "__main__":
    main()
