"""Timing of the online Hadamard rotation (+ deploy Quantizer) per route, C ABI through flatquant_amd.ops (GPU box only).
  python tools/time_had.py [n:K ...]      default: 28672:28 14336:28 11008:172
Routes: default (structured matrix-pipe kernel for n = K * 512 / K * 1024, dense Kronecker launch elsewhere), kron (dense pair), fwht
(register FWHT + K-factor: the bit-identical route), the rotation alone (fp16 out), the Quantizer alone. 16384 tokens (8192 for n > 20000)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from flatquant_amd import ops  # noqa: E402
from flatquant_amd._lib import FQ_OUT_PACKED, FQ_QUANT_F16, FQ_SIG_F16  # noqa: E402
from tests.conftest import hadk_matrix  # noqa: E402


def timeit(f, steps=50, rounds=5):
    for i in range(10):
        f(i)
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            f(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / steps * 1e3)
    return statistics.median(ts), min(ts)


def main():
    shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(28672, 28), (14336, 28), (11008, 172)]
    g = torch.Generator(device="cuda").manual_seed(0)
    sig = (0.9820137619972229, 0.9820137619972229)
    for n, K in shapes:
        r = 16384 if n < 20000 else 8192
        xs = [torch.randn(r, n, generator=g, device="cuda", dtype=torch.float32).half() for _ in range(2)]
        hk = torch.from_numpy(hadk_matrix(K)).cuda()
        pb = r * (2.5 * n + 2)
        cases = [("hadamard_quant default", lambda i: ops.hadamard_quant(xs[i % 2], K, hk, sig), pb),
                 ("hadamard_quant kron (dense)", lambda i: ops.hadamard_quant(xs[i % 2], K, hk, sig, route="kron"), pb),
                 ("hadamard_quant fwht", lambda i: ops.hadamard_quant(xs[i % 2], K, hk, sig, route="fwht"), pb),
                 ("hadamard default (fp16 out)", lambda i: ops.hadamard(xs[i % 2], K, hk), r * 4.0 * n),
                 ("hadamard fwht (fp16 out)", lambda i: ops.hadamard(xs[i % 2], K, hk, fwht_route=True), r * 4.0 * n),
                 ("deploy Quantizer (rowquant fp16)",
                  lambda i: ops.rowquant(xs[i % 2], [sig], FQ_OUT_PACKED | FQ_QUANT_F16 | FQ_SIG_F16), pb)]
        if os.environ.get("TIME_HAD_SILU"):      # the SiLU.mul input (x_gate, up): fused per route, and the separate launch
            ups = [torch.randn(r, n, generator=g, device="cuda", dtype=torch.float32).half() for _ in range(2)]
            sb = r * (4.5 * n + 2)
            cases = [("silu.mul + hadamard_quant default", lambda i: ops.hadamard_quant(xs[i % 2], K, hk, sig, up=ups[i % 2]), sb),
                     ("silu.mul + hadamard_quant kron", lambda i: ops.hadamard_quant(xs[i % 2], K, hk, sig, up=ups[i % 2], route="kron"), sb),
                     ("silu_mul alone (fq_silu_mul_f16)", lambda i: ops.silu_mul(xs[i % 2], ups[i % 2]), r * 6.0 * n),
                     cases[0]]
        elif os.environ.get("TIME_HAD_FAST"):      # A/B runs: the two structured-kernel launches only
            cases = [cases[0], cases[3]]
        for name, f, b in cases:
            us, mn = timeit(f)
            print(f"n={n:5d} rows={r:5d} {name:34s} {us:8.1f} us (min {mn:.1f})  {b / us / 1e3:7.0f} GB/s  {b / us / 8e6:5.3f} of 8 TB/s",
                  flush=True)
        del xs


if __name__ == "__main__":
    main()
