#!/usr/bin/env python3
"""A decoder layer with the MODULE STRUCTURE AND CALL ORDER of the reference's deploy model, built from the mirror modules — what
`import flatquant_amd.deploy as deploy` gives a maintainer who changes nothing else — and the same layer in fp16.

Structure restated from deploy/transformers/modeling_llama.py (the reference file cannot be imported here: it needs transformers
4.45's LlamaFlashAttention2 and flash-attn): attention :30-42, 66-78, 143-150 (inp_trans_q/k/v, quantizer_q/k/v, q/k/v_proj,
o_proj_trans, o_proj = Sequential(Quantizer, Linear4bit) under trans == "matmul": the transform returns a packed tensor the Quantizer
passes through); MLP :236-280 (inp_trans_u/g, up/gate_proj, down_proj = Sequential(OnlineTrans, Quantizer, Linear4bit)); the loader's
matrix sharing :518-529. The attention core (rotary, KV cache, flash attention) sits between v_proj and o_proj_trans in the reference
and is NOT part of either layer here — both layers get the same stand-in (the query states reshaped), as tools/bench_layer.py's
piece-wise table leaves it out on both sides.

    python tools/ref_layer.py [--model llama-2-7b] [--bsz 1] [--seq 2048]     W4A4 layer: default modules / after deploy.fuse / fp16
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flatquant_amd.deploy as deploy  # noqa: E402

MODELS = {
    "llama-2-7b": dict(hidden=4096, ffn=11008, heads=32, head_dim=128, kv_heads=32),
    "llama-3-8b": dict(hidden=4096, ffn=14336, heads=32, head_dim=128, kv_heads=8),
    "llama-2-70b": dict(hidden=8192, ffn=28672, heads=64, head_dim=128, kv_heads=8),
    "tiny": dict(hidden=4096, ffn=5120, heads=32, head_dim=128, kv_heads=8),        # (tests: every kernel family, small GEMMs)
}


def _fill_trans(t, g):
    for name in ("left_matrix", "right_matrix"):
        if name in t._buffers:
            b = t._buffers[name]
            b.copy_(torch.randn(b.shape, generator=g, device=b.device) / b.shape[0] ** 0.5)


def _lin(k_in, n_out, g, dev):
    lin = deploy.nn.Linear4bit(k_in, n_out).to(dev)
    lin.weight.copy_(torch.randint(0, 256, lin.weight.shape, generator=g, device=dev, dtype=torch.uint8))
    lin.weight_scales.copy_(torch.rand(lin.weight_scales.shape, generator=g, device=dev) * 0.02 + 0.005)
    return lin


class RefAttention(torch.nn.Module):
    def __init__(self, m, g, dev, trans="matmul"):
        super().__init__()
        H, kv = m["hidden"], m["kv_heads"] * m["head_dim"]
        self.num_heads, self.head_dim, self.hidden_size = m["heads"], m["head_dim"], H
        lac = trans == "matmul"
        for n in "qkv":
            setattr(self, f"inp_trans_{n}", deploy.nn.OnlineTrans(H, trans=trans, lac=lac).to(dev))
            setattr(self, f"quantizer_{n}", deploy.nn.Quantizer(lac=lac).to(dev))
        self.q_proj, self.k_proj, self.v_proj = _lin(H, H, g, dev), _lin(H, kv, g, dev), _lin(H, kv, g, dev)
        self.o_proj_trans = deploy.nn.OnlineTrans(m["heads"], trans=trans, decompose=False).to(dev)
        self.o_proj = torch.nn.Sequential(deploy.nn.Quantizer(lac=lac).to(dev), _lin(H, H, g, dev))
        _fill_trans(self.inp_trans_q, g), _fill_trans(self.o_proj_trans, g)
        for i, n in enumerate("qkv"):        # the loader: one matrix pair for the three transforms (:518-523), a clip pair each
            t = getattr(self, f"inp_trans_{n}")
            if n != "q" and "left_matrix" in t._buffers:
                for name in ("left_matrix", "right_matrix"):
                    del t._buffers[name]
                    t.register_buffer(name, self.inp_trans_q._buffers[name])
            t.clip_factor_a_max.fill_(4.0 - 0.7 * i)
            t.clip_factor_a_min.fill_(3.0 + 0.4 * i)

    def forward(self, hidden_states):
        bsz, q_len, _ = hidden_states.size()
        hq = self.quantizer_q(self.inp_trans_q(hidden_states))            # :66-78, in the reference's order
        hk = self.quantizer_k(self.inp_trans_k(hidden_states))
        hv = self.quantizer_v(self.inp_trans_v(hidden_states))
        query_states, key_states, value_states = self.q_proj(hq), self.k_proj(hk), self.v_proj(hv)
        attn_output = query_states.view(bsz, q_len, self.num_heads, self.head_dim)          # (stand-in for the attention core)
        attn_output = self.o_proj_trans(attn_output.transpose(-1, -2).contiguous())          # :143-146
        attn_output.quantized_x = attn_output.quantized_x.contiguous().reshape(bsz, q_len, -1)
        return self.o_proj(attn_output), key_states, value_states


class RefMLP(torch.nn.Module):
    def __init__(self, m, g, dev, trans="matmul", down="matmul"):
        super().__init__()
        H, F = m["hidden"], m["ffn"]
        lac = trans == "matmul"
        self.inp_trans_u = deploy.nn.OnlineTrans(H, trans=trans, lac=lac).to(dev)
        self.inp_trans_g = deploy.nn.OnlineTrans(H, trans=trans, lac=lac).to(dev)
        self.up_proj, self.gate_proj = _lin(H, F, g, dev), _lin(H, F, g, dev)
        dt = deploy.nn.OnlineTrans(F, trans=down).to(dev)
        _fill_trans(dt, g)
        # :248-253 (options.trans == "matmul": FlatQuant's own Kronecker transform of the ffn width; "had": the QuaRot-style rotation)
        self.down_proj = torch.nn.Sequential(dt, deploy.nn.Quantizer(lac=True).to(dev), _lin(F, H, g, dev))
        _fill_trans(self.inp_trans_u, g)
        for name in ("left_matrix", "right_matrix"):                      # :525-528
            if name in self.inp_trans_g._buffers:
                del self.inp_trans_g._buffers[name]
                self.inp_trans_g.register_buffer(name, self.inp_trans_u._buffers[name])
        self.inp_trans_g.clip_factor_a_max.fill_(3.1)
        self.inp_trans_g.clip_factor_a_min.fill_(2.7)
        self.act_fn = torch.nn.SiLU()

    def forward(self, x):                                                 # :268-280
        x_up = self.up_proj(self.inp_trans_u(x))
        x_gate = self.gate_proj(self.inp_trans_g(x))
        ac = self.act_fn(x_gate)
        x = x_up * ac
        return self.down_proj(x)


class RefLayer(torch.nn.Module):
    def __init__(self, model="llama-2-7b", seed=0, dev="cuda", down="matmul"):
        super().__init__()
        m = MODELS[model]
        g = torch.Generator(device=dev).manual_seed(seed)
        self.input_layernorm = deploy.nn.RMSNorm(m["hidden"])
        self.post_attention_layernorm = deploy.nn.RMSNorm(m["hidden"])
        self.self_attn = RefAttention(m, g, dev)
        self.mlp = RefMLP(m, g, dev, down=down)

    def forward(self, hidden_states):
        a, k, v = self.self_attn(self.input_layernorm(hidden_states))
        h = hidden_states + a
        return h + self.mlp(self.post_attention_layernorm(h)), k, v


class Fp16Layer(torch.nn.Module):
    """the same layer in fp16: seven nn.Linear, two RMSNorm, SiLU.mul (the reference's baseline, benchmarks/layer_benchmark.py:200-274)"""

    def __init__(self, model="llama-2-7b", dev="cuda"):
        super().__init__()
        m = MODELS[model]
        H, F, kv = m["hidden"], m["ffn"], m["kv_heads"] * m["head_dim"]
        mk = lambda i, o: torch.nn.Linear(i, o, bias=False, device=dev, dtype=torch.float16)
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = mk(H, H), mk(H, kv), mk(H, kv), mk(H, H)
        self.up_proj, self.gate_proj, self.down_proj = mk(H, F), mk(H, F), mk(F, H)
        self.w1 = torch.ones(H, device=dev, dtype=torch.float16)
        self.w2 = torch.ones(H, device=dev, dtype=torch.float16)
        self.H = H

    def forward(self, x):
        h = torch.nn.functional.rms_norm(x, (self.H,), self.w1, 1e-5)
        q, k, v = self.q_proj(h), self.k_proj(h), self.v_proj(h)
        x = x + self.o_proj(q)
        h = torch.nn.functional.rms_norm(x, (self.H,), self.w2, 1e-5)
        return x + self.down_proj(self.up_proj(h) * torch.nn.functional.silu(self.gate_proj(h))), k, v


def timeit(fn, steps=20, warm=5, settle_ms=80.0):
    import time
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < settle_ms:
        fn()
        torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / steps * 1e3)
    return sorted(ts)[1]


def graph_time(fn, x, steps=200):
    """us per replay of fn(x) captured in a HIP graph (x is a fixed input buffer)"""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn(x)
        with torch.cuda.graph(gr, stream=s):
            fn(x)
    torch.cuda.synchronize()
    for _ in range(10):
        gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-2-7b", choices=sorted(MODELS))
    ap.add_argument("--bsz", type=int, default=1)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--down", default="matmul", choices=["matmul", "had"], help="the down_proj input transform (options.trans)")
    ap.add_argument("--graph", action="store_true", help="also: the layer replayed from a captured HIP graph (decode sizes: the host cost is out of the picture)")
    a = ap.parse_args()
    xs = [torch.randn(a.bsz, a.seq, MODELS[a.model]["hidden"], device="cuda", dtype=torch.float16) for _ in range(3)]
    it = [0]

    def nxt():
        it[0] += 1
        return xs[it[0] % 3]
    with torch.no_grad():
        f16 = Fp16Layer(a.model)
        t16 = timeit(lambda: f16(nxt()), a.steps)
        g16 = graph_time(f16, xs[0]) if a.graph else None
        del f16
        print(f"{a.model}, {a.bsz} x {a.seq} tokens, one decoder layer WITHOUT the attention core (both sides), end to end, Python included")
        print(f"  fp16 layer (7 nn.Linear, 2 rms_norm, SiLU.mul)                              {t16:9.1f} us")
        layer = RefLayer(a.model, down=a.down)
        deploy.nn.Linear4bit.fp6_image = False
        t0 = timeit(lambda: layer(nxt()), a.steps)
        print(f"  W4A4 layer, reference structure, Linear4bit.fp6_image = False (round 4's default) {t0:9.1f} us   {t16 / t0:5.2f}x")
        for mod in layer.modules():
            if isinstance(mod, deploy.nn.Linear4bit):
                mod.release_images()
        deploy.nn.Linear4bit.fp6_image = True
        t1 = timeit(lambda: layer(nxt()), a.steps)
        print(f"  W4A4 layer, reference structure, default modules (round 5)                  {t1:9.1f} us   {t16 / t1:5.2f}x")
        if a.graph:
            g1 = graph_time(layer, xs[0])
            print(f"  captured HIP graph: fp16 layer {g16:9.1f} us, W4A4 layer (reference structure, one launch per module) {g1:9.1f} us   {g16 / g1:5.2f}x")
        rep = deploy.fuse(layer, linears=False)
        t2 = timeit(lambda: layer(nxt()), a.steps)
        print(f"  ... after deploy.fuse(model, linears=False) {rep}: {t2:9.1f} us   {t16 / t2:5.2f}x")
        deploy.unfuse(layer)
        rep = deploy.fuse(layer)
        t3 = timeit(lambda: layer(nxt()), a.steps)
        print(f"  ... after deploy.fuse(model) {rep}: {t3:9.1f} us   {t16 / t3:5.2f}x")
        if a.graph:
            g3 = graph_time(layer, xs[0])
            print(f"  captured HIP graph after deploy.fuse(model) (the groups serve decode-sized calls under capture) {g3:9.1f} us   {g16 / g3:5.2f}x")
        deploy.fuse(layer, static_outputs=True)
        t4 = timeit(lambda: layer(nxt()), a.steps)
        print(f"  ... after deploy.fuse(model, static_outputs=True)                             {t4:9.1f} us   {t16 / t4:5.2f}x")
        if a.bsz * a.seq <= 128:     # (round 6) decode sizes: the layer's calls served from a captured graph, no graph code here
            deploy.unfuse(layer)
            for mod in layer.modules():
                if isinstance(mod, (deploy.nn.OnlineTrans, deploy.nn.Quantizer, deploy.nn.Linear4bit)):
                    mod.static_outputs = False
            rep = deploy.fuse(layer, capture=True)
            t5 = timeit(lambda: layer(nxt()), a.steps)
            f16 = Fp16Layer(a.model)
            g16t = deploy.GraphedDecode(f16.forward)
            t16g = timeit(lambda: g16t(nxt()), a.steps)
            print(f"  ... after deploy.fuse(model, capture=True) {rep}: {t5:9.1f} us   {t16 / t5:5.2f}x the eager fp16 layer, "
                  f"{t16g / t5:5.2f}x the fp16 layer through the same helper ({t16g:.1f} us)")


if __name__ == "__main__":
    main()
