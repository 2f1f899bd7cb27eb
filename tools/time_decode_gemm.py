#!/usr/bin/env python3
"""Decode-sized Linear4bit GEMMs (M = 16 .. 128 rows) on their two routes: the weight-streaming int8 kernel over the fragment image (0.5 B/param,
fq_int4_skinny_linear[_multi]_f16) and the FP6 tile kernel over the kept FP6 image (0.75 B/param) INCLUDING the conversion of the packed activations
(fq_int4_linear_fp6[_multi]_f16). Llama-3-8B shapes; weights rotate over SETS images so that no launch re-reads a weight from the Infinity Cache.
us per call (60 back-to-back calls between one event pair) and bit-exactness of the two routes against each other."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flatquant_amd import ops  # noqa: E402

REPS = int(os.environ.get("REPS", "60"))
g = torch.Generator(device="cuda").manual_seed(0)


def timeit(fn):
    for i in range(8):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(REPS):
        fn(i)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REPS


def main():
    print(f"[{os.environ.get('FQHIP_OVERLAY', 'default').split('/')[-1]}] us per call: weight-streaming int8 kernel | FP6 tile kernel incl. activation conversion")
    for name, Ns, K in (("q or o 4096x4096", [4096], 4096), ("q/k/v multi", [4096, 1024, 1024], 4096), ("up or gate 14336x4096", [14336], 4096),
                        ("up+gate multi", [14336, 14336], 4096), ("down 4096x14336", [4096], 14336)):
        sets = max(2, int(300e6 // (sum(Ns) * K // 2)) + 1)
        ws = [[torch.randint(0, 256, (N, K // 2), generator=g, device="cuda", dtype=torch.uint8) for N in Ns] for _ in range(sets)]
        dimg = [[ops.int4_to_frag(w) for w in s] for s in ws]
        fimg = [[ops.int4_to_bf6(w, weights=True) for w in s] for s in ws]
        wsc = [torch.full((N,), 0.01, device="cuda", dtype=torch.float16) for N in Ns]
        row = []
        for M in (16, 32, 48, 64, 96, 128):
            xs = [torch.randint(0, 256, (M, K // 2), generator=g, device="cuda", dtype=torch.uint8) for _ in Ns]
            sx = torch.full((M,), 0.02, device="cuda", dtype=torch.float16)
            if len(Ns) == 1:
                a = lambda i: ops.int4_skinny_linear(xs[0], sx, dimg[i % sets][0], wsc[0], None, Ns[0])
                b = lambda i: ops.int4_linear_fp6(xs[0], sx, ws[i % sets][0], fimg[i % sets][0], wsc[0], None)
                same = torch.equal(a(0), b(0))
            else:
                a = lambda i: ops.int4_skinny_linear_multi([(xs[p], sx, dimg[i % sets][p], wsc[p], None) for p in range(len(Ns))])
                b = lambda i: ops.int4_linear_fp6_multi([(xs[p], sx, ws[i % sets][p], fimg[i % sets][p], wsc[p], None) for p in range(len(Ns))])
                same = all(torch.equal(u, v) for u, v in zip(a(0), b(0)))
            ta, tb = timeit(a), timeit(b)
            row.append(f"M={M}: {ta:6.1f} | {tb:6.1f}{'' if same else ' MISMATCH'}")
        print(f"  {name:24s} " + "   ".join(row))
        del ws, dimg, fimg
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
