#!/usr/bin/env python3
"""The reference's decode-attention benchmark (benchmarks/qattention_benchmark.py:39-124), same method, on flatquant_amd's
MultiLayerPagedKVCache4Bit: a "fake prefill" (the cache is declared seq_len - 1 tokens long; its pages hold whatever they hold), then per
step ``update`` (K transform + K/V quantise + append of one token per request) + the decode attention of one query token — fp16 cache
(disable_quant), INT4, INT4 + Hadamard rotation, INT4 + learned K transform (matmul, with its inverse-transpose on the query side).
5 warm-up + 100 timed steps on the wall clock (host launches included, as there), 10 repetitions, mean +- 1.96 sigma.
    tools/bench_qattention.py [--bsz B] [--seq_len S] [--graph]      --graph: the same step replayed from a captured HIP graph as well
Model sizes as there: (layers, heads, head_dim) = (1, 32, 128) llama-7b, (1, 40, 128) llama-13b, (1, 64, 128) llama-70b."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flatquant_amd.deploy.transformers as dt  # noqa: E402

SIZES = [(1, 32, 128, "llama-7b"), (1, 40, 128, "llama-13b"), (1, 64, 128, "llama-70b")]
WARM, STEPS, REPS = 5, 100, 10


def make_step(n_layers, heads, hd, bsz, seq_len, fp16, trans):
    dev = torch.device("cuda:0")
    cache = dt.MultiLayerPagedKVCache4Bit(bsz, seq_len, seq_len, dev, n_layers, heads, hd, disable_quant=fp16,
                                          trans_dtype=torch.float16, trans=trans or "none")
    if not fp16:
        cache.pages.random_(0, 256)
        cache.scales.uniform_(0.05, 0.25)
    else:
        cache.pages.normal_()
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.rand((bsz, 1, heads, hd), device=dev, dtype=torch.float16, generator=g)
    k = torch.rand((bsz, 1, heads, hd), device=dev, dtype=torch.float16, generator=g)
    v = torch.rand((bsz, 1, heads, hd), device=dev, dtype=torch.float16, generator=g)
    kw = {}
    if trans == "matmul":
        t = (torch.randn(hd, hd, device=dev, generator=g) / hd ** 0.5).half()
        kw = {"trans_matrix_k": t, "trans_matrix_k_inv_t": torch.linalg.inv(t.float()).t().half().contiguous()}

    def step():
        cache._needs_init = [False] * len(cache._needs_init)
        cache.length = seq_len - 1
        return cache.update(k, v, 0, dict(kw))(q)
    return step


def wall(step):
    for _ in range(WARM):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / STEPS


def graphed(step):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        step()
        with torch.cuda.graph(gr, stream=s):
            step()
    torch.cuda.synchronize()
    for _ in range(WARM):
        gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(STEPS):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / STEPS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bsz", type=int, default=None)
    ap.add_argument("--seq_len", type=int, default=2048)
    ap.add_argument("--graph", action="store_true")
    a = ap.parse_args()
    for bsz in ([a.bsz] if a.bsz is not None else [1, 2, 4, 8, 16, 32]):
        for n_layers, heads, hd, name in SIZES:
            res = {}
            for label, fp16, trans in (("FP16", True, None), ("Int4", False, None), ("Int4 (+had)", False, "had"), ("Int4 (+inv)", False, "matmul")):
                step = make_step(n_layers, heads, hd, bsz, a.seq_len, fp16, trans)
                ts = [wall(step) for _ in range(REPS)]
                res[label] = (float(np.mean(ts)), float(1.96 * np.std(ts)), graphed(step) if a.graph else None)
                del step
                torch.cuda.empty_cache()
            base, gbase = res["FP16"][0], res["FP16"][2]
            print(f"bsz {bsz:3d}  seq_len {a.seq_len}  {name} ({heads} heads x {hd})")
            for label, (m, ci, gms) in res.items():
                line = f"    {label:12s} {m:7.3f} +- {ci:5.3f} ms   speed-up {base / m:5.3f}"
                if gms is not None:
                    line += f"      captured graph {gms * 1e3:8.1f} us   speed-up {gbase / gms:5.3f}"
                print(line, flush=True)


if __name__ == "__main__":
    main()
