#!/usr/bin/env python3
"""The reference's kernel benchmark (benchmarks/kernel_benchmark.py:234-276) on one MI355X: for every factor pair of its
list, batch sizes 1..64, prefill (bs x 2048 tokens) and decode (bs tokens):
  baseline  the unfused sequence it calls 'cublas': two torch.matmul (rocBLAS / hipBLASLt here) + the Quantizer module
            (ours: one launch; the reference's is 5-8 torch ops + a pack kernel, so this baseline is FASTER than its own)
  fused     deploy.functional.kronecker_matmul — the drop-in for the Triton matmul(a, b, c, S) — one HIP launch
Method as triton.testing.do_bench: every repetition timed on its own with the caches flushed in between (a 512 MB
memset: the part has 256 MB of MALL), median of the repetitions (p20 / p80 with QUANTILES=1). Milliseconds, like the
reference's table."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flatquant_amd.deploy.functional.online_trans import kronecker_matmul  # noqa: E402
from flatquant_amd.deploy.nn import Quantizer  # noqa: E402

PAIRS = [(64, 64), (64, 80), (64, 128), (86, 128), (108, 128), (112, 128), (128, 224)]
BS = [1, 2, 4, 8, 16, 32, 64]
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")


def bench(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        flush.zero_()
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2], t[int(0.2 * len(t))], t[int(0.8 * len(t))]


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    quantizer = Quantizer().cuda()
    show_q = bool(os.environ.get("QUANTILES"))
    for M, N in PAIRS:
        a = (torch.randn(M, M, generator=g, device="cuda") / M ** 0.5).half()
        c = (torch.randn(N, N, generator=g, device="cuda") / N ** 0.5).half()
        aT = a.T.contiguous()
        print(f"==================== Dimension Size: {M * N} ({M} x {N}) ====================")
        print(" bs | prefill baseline | decode baseline | prefill fused | decode fused | prefill speedup | decode speedup   (ms)")
        for bs in BS:
            row = {}
            for name, seq in (("prefill", 2048), ("decode", 1)):
                B = bs * seq
                b = torch.randn(B, M, N, generator=g, device="cuda", dtype=torch.float16)
                x3 = b.view(bs, seq, M * N)
                reps = 20 if B >= 16384 else 40
                row[name, "base"] = bench(lambda: quantizer(torch.matmul(torch.matmul(a, b), c).view(B, -1)), reps)
                row[name, "fused"] = bench(lambda: kronecker_matmul(x3, [aT, c]), reps)
                del b, x3
            pb, db, pf, df = (row[k][0] for k in (("prefill", "base"), ("decode", "base"), ("prefill", "fused"), ("decode", "fused")))
            line = f" {bs:2d} | {pb:8.4f} | {db:8.4f} | {pf:8.4f} | {df:8.4f} | {pb / pf:5.2f}x | {db / df:5.2f}x"
            if show_q:
                line += "   p20/p80 fused prefill %.4f/%.4f" % row["prefill", "fused"][1:]
            print(line, flush=True)


if __name__ == "__main__":
    main()
