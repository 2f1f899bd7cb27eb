#!/usr/bin/env python3
"""BASELINE config C3/C4: the activation path of ONE decoder layer through the deploy modules (GPU box only).

Per layer the reference's deploy model runs, in front of its 4-bit linears (deploy/transformers/modeling_llama.py):
  ln_trans      OnlineTrans(hidden, matmul, decompose)   -> packed  (input of q/k/v_proj)
  o_proj trans  OnlineTrans(num_heads, matmul, no decompose) on [bsz, seq, head_dim, heads] -> packed
  up_gate_trans OnlineTrans(hidden, matmul, decompose)   -> packed  (input of up/gate_proj)
  down_proj     OnlineTrans(ffn, had) -> fp16, then Quantizer(lac) -> packed
This script times exactly those module calls (random matrices, synthetic activations, 8 x 2048 tokens) and prints
microseconds per call, the algorithmic GB/s and the layer total. The GEMMs themselves are out of scope.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flatquant_amd.deploy as deploy  # noqa: E402

MODELS = {
    "llama-2-7b": dict(hidden=4096, ffn=11008, heads=32, head_dim=128, kv_heads=32),   # the model of the reference's own layer benchmark (README.md:288-310)
    "llama-3-8b": dict(hidden=4096, ffn=14336, heads=32, head_dim=128, kv_heads=8),
    "llama-2-70b": dict(hidden=8192, ffn=28672, heads=64, head_dim=128, kv_heads=8),
}


GRAPH = False   # --graph: every timed callable is captured once into a HIP graph (torch.cuda.graph) and replayed — at 2048 tokens the
                # Python module path (output allocation, ctypes marshalling: ~25 us per call) is longer than most of the kernels


def timeit(fn, steps, warm=10, settle_ms=60.0):
    """events around `steps` calls, after `warm` calls AND at least `settle_ms` of the same launches (the part needs tens of
    milliseconds of load before its clocks settle: ten 200 us launches measure the ramp, 20 % slow)"""
    import time
    for _ in range(warm):
        fn()
    if GRAPH:
        torch.cuda.synchronize()
        from flatquant_amd import ops
        ops.images_ready()      # the warm-up's fragment images are complete: captured launches share them instead of preparing their own
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph):
            keep = fn()   # noqa: F841  (outputs stay alive for the replays)
        fn = gph.replay
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < settle_ms:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    # median of three event-timed windows: one window is hostage to a single slow event inside it (an allocator round trip to the
    # driver after the previous piece's tensors were freed measured 4x on one run of the q/k/v launch: profiles/r04_final_layer_l3_bs8.txt)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b", choices=sorted(MODELS))
    ap.add_argument("--bsz", type=int, default=8)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--graph", action="store_true", help="time HIP-graph replays of every piece (kernel time, no Python launch path)")
    a = ap.parse_args()
    global GRAPH
    GRAPH = a.graph
    m = MODELS[a.model]
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    T = a.bsz * a.seq

    def act(*shape):
        return torch.randn(*shape, generator=g, device=dev, dtype=torch.float16)

    def trans(dim, decompose=True):
        t = deploy.nn.OnlineTrans(dim, trans="matmul", decompose=decompose, lac=True).to(dev)
        for name in ("left_matrix", "right_matrix"):
            if hasattr(t, name):
                b = getattr(t, name)
                b.copy_(torch.randn(b.shape, generator=g, device=dev) / b.shape[0] ** 0.5)
        t.clip_factor_a_max.fill_(4.0)
        t.clip_factor_a_min.fill_(4.0)
        return t

    ln_trans, ug_trans = trans(m["hidden"]), trans(m["hidden"])
    o_trans = trans(m["heads"], decompose=False)
    had = deploy.nn.OnlineTrans(m["ffn"], trans="had").to(dev)
    quant = deploy.nn.Quantizer(lac=True).to(dev)
    xs = [act(a.bsz, a.seq, m["hidden"]) for _ in range(3)]
    xo = [act(a.bsz, a.seq, m["head_dim"], m["heads"]) for _ in range(3)]
    xf = [act(a.bsz, a.seq, m["ffn"]) for _ in range(2)]
    it = [0]

    def nxt(lst):
        it[0] += 1
        return lst[it[0] % len(lst)]

    rows = [
        ("ln_trans (q/k/v input)", lambda: ln_trans(nxt(xs)), 2.5 * m["hidden"] + 2),
        ("o_proj head transform", lambda: o_trans(nxt(xo)), 2.5 * m["hidden"] + 2),
        ("up_gate_trans", lambda: ug_trans(nxt(xs)), 2.5 * m["hidden"] + 2),
        ("down_proj Hadamard", lambda: had(nxt(xf)), 4.0 * m["ffn"]),
        ("down_proj Quantizer", lambda: quant(nxt(xf)), 2.5 * m["ffn"] + 2),
    ]
    fused = ("down_proj Hadamard+Quantizer, one launch (OnlineTrans.forward(x, quantizer=...))",
             lambda: had(nxt(xf), quantizer=quant), 2.5 * m["ffn"] + 2)
    total = 0.0
    print(f"{a.model}: {a.bsz} x {a.seq} tokens, one decoder layer, activation path through flatquant_amd.deploy.nn"
          + (" — every piece replayed from a HIP graph" if GRAPH else ""))
    for name, fn, bpt in rows:
        us = timeit(fn, a.steps)
        total += us
        print(f"  {name:26s} {us:9.1f} us   {T * bpt / us / 1e3:7.0f} GB/s algorithmic")
    print(f"  {'layer total':26s} {total:9.1f} us   ({T / total:.1f} tokens/us through the activation path)")
    fused_us = us = timeit(fused[1], a.steps)
    two = sum(timeit(fn, a.steps) for _, fn, _ in rows[3:])
    print(f"  {fused[0]}: {us:.1f} us ({T * fused[2] / us / 1e3:.0f} GB/s) instead of {two:.1f} us"
          f" -> layer total {total - two + us:.1f} us")

    # RMSNorm in front of ln_trans / up_gate_trans: separate launch vs fused into the transform launch
    norm = deploy.nn.RMSNorm(m["hidden"])
    sep = timeit(lambda: ln_trans(norm(nxt(xs))), a.steps)
    fus = timeit(lambda: ln_trans(nxt(xs), norm=norm), a.steps)
    print(f"  RMSNorm + ln_trans: {sep:.1f} us as two launches, {fus:.1f} us fused (OnlineTrans.forward(x, norm=...))")

    # q/k/v (three clip pairs) and up/gate (two) share one factor pair: one launch each instead of three / two
    qkv = [trans(m["hidden"]) for _ in range(3)]
    for t in qkv[1:]:
        for name in ("left_matrix", "right_matrix"):
            del t._buffers[name]
            t.register_buffer(name, getattr(qkv[0], name))
    three = timeit(lambda: [t(nxt(xs)) for t in qkv], a.steps)
    one = timeit(lambda: deploy.nn.fused_forward(nxt(xs), qkv), a.steps)
    one_n = timeit(lambda: deploy.nn.fused_forward(nxt(xs), qkv, norm=norm), a.steps)
    ug_n = timeit(lambda: deploy.nn.fused_forward(nxt(xs), qkv[:2], norm=norm), a.steps)
    print(f"  inp_trans_q/k/v: {three:.1f} us as three launches (reference structure), {one:.1f} us as one, "
          f"{one_n:.1f} us with RMSNorm inside; up/gate pair with RMSNorm: {ug_n:.1f} us")

    # x_up * silu(x_gate) in front of the down_proj transform: eager torch, one HIP launch, fused into the transform
    from flatquant_amd import ops
    gates, ups = xf, [act(a.bsz, a.seq, m["ffn"]) for _ in range(2)]
    eager = timeit(lambda: torch.nn.functional.silu(nxt(gates)) * ups[0], a.steps)
    alone = timeit(lambda: ops.silu_mul(nxt(gates), ups[0]), a.steps)
    hq = timeit(lambda: had(nxt(gates), quantizer=quant, up=ups[0]), a.steps)
    down_mm = trans(m["ffn"])
    mm = timeit(lambda: down_mm(nxt(xf)), a.steps)
    mmf = timeit(lambda: down_mm(nxt(gates), up=ups[0]), a.steps)
    print(f"  SiLU.mul: torch eager {eager:.1f} us, fq_silu_mul_f16 {alone:.1f} us")
    print(f"  down_proj input, Hadamard + Quantizer: {fused_us:.1f} us -> with SiLU.mul inside {hq:.1f} us")
    print(f"  down_proj input, FlatQuant matmul transform ({m['ffn']}): {mm:.1f} us -> with SiLU.mul inside {mmf:.1f} us")

    # FlatQuant's own deploy layer (trans="matmul" everywhere): the reference's launch structure on these kernels vs
    # the fused launches
    o_us = timeit(lambda: o_trans(nxt(xo)), a.steps)
    norm_us = timeit(lambda: norm(nxt(xs)), a.steps)
    pair = timeit(lambda: [t(nxt(xs)) for t in qkv[:2]], a.steps)
    ref_struct = norm_us + three + o_us + norm_us + pair + alone + mm
    fused_struct = one_n + o_us + ug_n + mmf
    print(f"  FlatQuant layer, activation path: reference launch structure {ref_struct:.1f} us "
          f"(RMSNorm {norm_us:.1f} x2, q/k/v {three:.1f}, o {o_us:.1f}, up/gate {pair:.1f}, SiLU.mul {alone:.1f}, down {mm:.1f}; "
          f"with torch-eager SiLU.mul {ref_struct - alone + eager:.1f}) -> fused launches {fused_struct:.1f} us "
          f"(norm+q/k/v {one_n:.1f}, o {o_us:.1f}, norm+up/gate {ug_n:.1f}, SiLU.mul+down {mmf:.1f})")

    # KV-cache side: K transform + asym INT4 pack of the new keys / values (kv_cache.py:262-297), fused vs torch eager
    import flatquant_amd.deploy.transformers as dt
    ks = [act(a.bsz, a.seq, m["kv_heads"], m["head_dim"]) for _ in range(2)]
    tk = (torch.randn(m["head_dim"], m["head_dim"], generator=g, device=dev) / m["head_dim"] ** 0.5).half()

    def eager(k, v):
        out = []
        for t in (torch.matmul(k, tk), v):
            xmax, xmin = t.amax(-1, keepdim=True), t.amin(-1, keepdim=True)
            scale = (xmax - xmin).clamp(min=1e-5) / 15
            q = torch.clamp(torch.round((t - xmin) / scale), 0, 15).to(torch.uint8)
            out.append((q[..., 0::2] | (q[..., 1::2] << 4), scale, -xmin))
        return out
    kv_e = timeit(lambda: eager(nxt(ks), ks[0]), a.steps)
    kv_f = timeit(lambda: dt.transform_quantize_kv(nxt(ks), ks[0], tk), a.steps)
    print(f"  K transform + K/V asym INT4 pack ({a.bsz}x{a.seq} tokens x {m['kv_heads']} heads x {m['head_dim']}): "
          f"torch eager (the reference's op sequence) {kv_e:.1f} us, fq_kv_quant_f16 x2 {kv_f:.1f} us")

    # the seven 4-bit linears that consume those packed activations (Linear4bit = INT4 GEMM + dequant epilogue)
    kv = m["kv_heads"] * m["head_dim"]
    lins = [("q_proj", m["hidden"], m["hidden"]), ("k_proj", m["hidden"], kv), ("v_proj", m["hidden"], kv),
            ("o_proj", m["hidden"], m["hidden"]), ("up_proj", m["hidden"], m["ffn"]),
            ("gate_proj", m["hidden"], m["ffn"]), ("down_proj", m["ffn"], m["hidden"])]
    packed = {m["hidden"]: ln_trans(xs[0]), m["ffn"]: had(xf[0], quantizer=quant)}
    deploy.nn.Linear4bit.fp6_image = False   # (round 4's default: the weights converted per call)
    gtot = 0.0
    for name, k_in, n_out in lins:
        lin = deploy.nn.Linear4bit(k_in, n_out).to(dev)
        lin.weight_scales.fill_(0.01)
        us = timeit(lambda: lin(packed[k_in]), max(a.steps // 5, 5), warm=3)
        gtot += us
        print(f"  {'Linear4bit ' + name:26s} {us:9.1f} us   {2.0 * T * k_in * n_out / us / 1e6:7.0f} TOP/s")
        del lin
    print(f"  {'seven linears':26s} {gtot:9.1f} us;  FlatQuant layer, fused activation path + linears: {fused_struct + gtot:.1f} us")

    # the same seven linears with the FP6 operand image of the weights (Linear4bit.fp6_image: +0.75 B/param, same bits out)
    deploy.nn.Linear4bit.fp6_image = True   # (the default since round 5)
    g6 = 0.0
    for name, k_in, n_out in lins:
        lin = deploy.nn.Linear4bit(k_in, n_out).to(dev)
        lin.weight_scales.fill_(0.01)
        us = timeit(lambda: lin(packed[k_in]), max(a.steps // 5, 5), warm=3)
        g6 += us
        del lin
    print(f"  {'seven linears, FP6 path':26s} {g6:9.1f} us;  FlatQuant layer, fused activation path + linears: {fused_struct + g6:.1f} us")
    # (round 4) q / k / v as ONE GEMM launch and up / gate as one (deploy.nn.linear.linear4bit_multi, fq_int4_linear_fp6_multi_f16): each
    # projection keeps its own packed input and weights; at a few thousand tokens a single projection does not fill the chip
    from flatquant_amd.deploy.nn.linear import linear4bit_multi
    gm, gm_by = 0.0, {}
    for names in (("q_proj", "k_proj", "v_proj"), ("o_proj",), ("up_proj", "gate_proj"), ("down_proj",)):
        mods, ins = [], []
        for name, k_in, n_out in lins:
            if name in names:
                lin = deploy.nn.Linear4bit(k_in, n_out).to(dev)
                lin.weight_scales.fill_(0.01)
                mods.append(lin)
                # (each projection its own packed input, as under FlatQuant's per-projection clip factors: a shared one flatters the launch)
                ins.append(packed[k_in] if not ins else deploy.PackedQuantizedTensor(packed[k_in].quantized_x.clone(), packed[k_in].scales_x.clone()))
        us = timeit(lambda: linear4bit_multi(mods, ins), max(a.steps // 5, 5), warm=3)
        gm += us
        gm_by[names[0]] = us
        print(f"  {'Linear4bit ' + ' + '.join(names) + ', one launch':58s} {us:9.1f} us")
        del mods
    print(f"  {'seven linears, FP6 path, q/k/v and up/gate as one launch each':62s} {gm:9.1f} us;  FlatQuant layer: {fused_struct + gm:.1f} us")
    # FP16 baseline of the same layer pieces (what benchmarks/layer_benchmark.py:200-274 compares against): the seven
    # nn.Linear GEMMs in fp16 (rocBLAS / hipBLASLt through torch), two RMSNorms and SiLU.mul in torch eager; the
    # attention core is in neither number.
    f16tot = 0.0
    for name, k_in, n_out in lins:
        lin = torch.nn.Linear(k_in, n_out, bias=False, device=dev, dtype=torch.float16)
        xin = xs[0] if k_in == m["hidden"] else xf[0]
        us = timeit(lambda: lin(xin), max(a.steps // 5, 5), warm=3)
        f16tot += us
        print(f"  {'fp16 nn.Linear ' + name:26s} {us:9.1f} us   {2.0 * T * k_in * n_out / us / 1e6:7.0f} TFLOP/s")
        del lin
    wn = torch.ones(m["hidden"], device=dev, dtype=torch.float16)
    n16 = timeit(lambda: torch.nn.functional.rms_norm(nxt(xs), (m["hidden"],), wn, 1e-5), a.steps)
    sm16 = timeit(lambda: torch.nn.functional.silu(xf[0]) * xf[1], a.steps)
    f16layer = f16tot + 2 * n16 + sm16
    print(f"  fp16 layer (seven linears {f16tot:.1f} us + 2 x torch rms_norm {n16:.1f} + SiLU.mul eager {sm16:.1f}): {f16layer:.1f} us; "
          f"FlatQuant W4A4 layer {fused_struct + gtot:.1f} us -> {f16layer / (fused_struct + gtot):.2f}x "
          f"(linears alone {f16tot / gtot:.2f}x); with the FP6 operand image {fused_struct + g6:.1f} us -> "
          f"{f16layer / (fused_struct + g6):.2f}x (linears alone {f16tot / g6:.2f}x); with q/k/v and up/gate as one launch each "
          f"{fused_struct + gm:.1f} us -> {f16layer / (fused_struct + gm):.2f}x (linears alone {f16tot / gm:.2f}x)")


if __name__ == "__main__":
    main()
