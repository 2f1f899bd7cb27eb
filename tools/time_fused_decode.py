#!/usr/bin/env python3
"""The fused decode launch (fq_kron64_linear_multi_f16: [RMSNorm +] 64 x 64 transform + quantiser + q / k / v or up / gate projections) against
the two launches it replaces, Llama-3-8B shapes, M = 1 .. 16 tokens: us per call from a captured HIP graph of REPS calls each (what a captured
decode step sees), the weights rotated over NB layers' worth of images so that they stream from HBM as in a real model."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flatquant_amd import ops  # noqa: E402
from flatquant_amd._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED  # noqa: E402

REPS = int(os.environ.get("REPS", "40"))
NB = int(os.environ.get("NB", "6"))
TAG = os.environ.get("FQHIP_OVERLAY", "default").split("/")[-1]


def graph_time(fn, warm=None):
    warm = NB if warm is None else warm
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(warm):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(REPS):
                fn(i)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / REPS)
    return sorted(ts)[len(ts) // 2]


def main():
    gen = torch.Generator(device="cuda").manual_seed(0)
    L = (torch.randn(64, 64, generator=gen, device="cuda") / 8).half()
    R = (torch.randn(64, 64, generator=gen, device="cuda") / 8).half()
    for name, Ns in (("q/k/v", (4096, 1024, 1024)), ("up/gate", (14336, 14336))):
        sigs = [(0.98, 0.98)] * len(Ns)
        imgs = [[ops.int4_to_frag(torch.randint(0, 256, (N, 2048), generator=gen, device="cuda", dtype=torch.uint8)) for N in Ns] for _ in range(NB)]
        wss = [torch.rand(N, generator=gen, device="cuda").half() * 0.01 for N in Ns]
        for M in (1, 4, 8, 16):
            x = torch.randn(M, 4096, generator=gen, device="cuda").half()

            def two(i):
                o = ops.rmsnorm_kron_quant(x, 1e-5, L, R, sigs, FQ_OUT_PACKED | FQ_NO_CLAMP0)
                return ops.int4_skinny_linear_multi([(o.q[p], o.scale[p], imgs[i % NB][p], wss[p], None) for p in range(len(Ns))])

            def one(i):
                return ops.kron64_linear_multi(x, L, R, sigs, [(imgs[i % NB][p], wss[p], None) for p in range(len(Ns))], eps=1e-5, flags=FQ_NO_CLAMP0)

            a, b = two(0), one(0)
            ok = all(torch.equal(u, v) for u, v in zip(a, b))
            t2, t1 = graph_time(two), graph_time(one)
            mb = sum(Ns) * 2048 / 1e6
            print(f"[{TAG}] {name:8s} M={M:2d}: two launches {t2:6.2f} us, fused {t1:6.2f} us ({mb / t1:5.2f} TB/s of weights)  exact={ok}", flush=True)


def main_lone():
    """the lone projections of the step (o_proj 4096 x 4096, down_proj 4096 x 14336) on the weight-streaming kernel, graph-replayed"""
    gen = torch.Generator(device="cuda").manual_seed(2)
    for name, N, K in (("o_proj", 4096, 4096), ("down", 4096, 14336)):
        nb = max(NB, int(400e6 // (N * K // 2)))
        imgs = [ops.int4_to_frag(torch.randint(0, 256, (N, K // 2), generator=gen, device="cuda", dtype=torch.uint8)) for _ in range(nb)]
        wss = [torch.rand(N, generator=gen, device="cuda").half() * 0.01 for _ in range(nb)]   # (a layer's own scales: cold, like its weights)
        for M in (1, 8, 16):
            x = torch.randint(0, 256, (M, K // 2), generator=gen, device="cuda", dtype=torch.uint8)
            sx = torch.rand(M, generator=gen, device="cuda").half() * 0.01
            t = graph_time(lambda i: ops.int4_skinny_linear(x, sx, imgs[i % nb], wss[i % nb], None, N), warm=nb)
            print(f"[{TAG}] {name:8s} M={M:2d}: {t:6.2f} us ({N * K / 2 / 1e6 / t:5.2f} TB/s of weights)", flush=True)


def main_silu():
    """the down_proj input of a decode step: SiLU.mul + 112 x 128 transform + quantiser (fq_silu_mul_kron_quant_f16), graph-replayed"""
    gen = torch.Generator(device="cuda").manual_seed(3)
    L = (torch.randn(112, 112, generator=gen, device="cuda") / 10).half()
    R = (torch.randn(128, 128, generator=gen, device="cuda") / 11).half()
    for M in (1, 4, 8, 16, 64):
        gate = torch.randn(M, 14336, generator=gen, device="cuda").half()
        up = torch.randn(M, 14336, generator=gen, device="cuda").half()
        t = graph_time(lambda i: ops.silu_mul_kron_quant(gate, up, L, R, [(0.98, 0.98)], FQ_OUT_PACKED | 0x08), warm=3)
        print(f"[{TAG}] silu.mul + 112x128  M={M:2d}: {t:6.2f} us", flush=True)


if __name__ == "__main__":
    if os.environ.get("SILU"):
        main_silu()
        sys.exit(0)
    if os.environ.get("LONE"):
        main_lone()
        sys.exit(0)
    main()
