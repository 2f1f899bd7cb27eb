#!/usr/bin/env python3
"""Structured probes of v_mfma_f32_32x32x16_f16 accumulation (run on the GPU box).

Each case places chosen PRODUCTS p_k = a_k * b_k (k = 0..15, exact in fp16 x fp16) in output element (0,0)
and an accumulator input C, and prints the fp32 result next to (a) the exactly rounded sum and (b) a
sequential RNE chain, so that rounding mode, grouping and extra internal bits can be read off.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flatquant_amd import _probe, ops  # noqa: E402


def run(prods, c=0.0):
    """prods: dict k -> (a, b) fp16-representable factors; returns D[0,0]."""
    A = np.zeros((32, 16), np.float16)
    B = np.zeros((16, 32), np.float16)
    for k, (a, b) in prods.items():
        A[0, k], B[k, 0] = a, b
        assert float(A[0, k]) == a and float(B[k, 0]) == b, (k, a, b)
    C = np.zeros((32, 32), np.float32)
    C[0, 0] = c
    D = _probe.probe_mfma(torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(), torch.from_numpy(C).cuda())
    return float(D[0, 0].cpu())


def exact(prods, c=0.0):
    from fractions import Fraction
    s = Fraction(c)
    for a, b in prods.values():
        s += Fraction(a) * Fraction(b)
    return float(np.float32(float(s)))  # double is exact enough for these few-term sums


def seq(prods, c=0.0):
    acc = np.float32(c)
    for k in sorted(prods):
        a, b = prods[k]
        acc = np.float32(np.float64(acc) + np.float64(a) * np.float64(b))
    return float(acc)


def show(name, prods, c=0.0):
    d, e, s = run(prods, c), exact(prods, c), seq(prods, c)
    print(f"{name:58s} mfma={d!r:24} exact_rne={e!r:24} seq_rne={s!r:24} "
          f"{'=exact' if d == e else ''} {'=seq' if d == s else ''}")


def main():
    u = 2.0 ** -12           # u*u = 2^-24 = half an ulp of 1.0f
    h = 2.0 ** -12
    one = (1.0, 1.0)
    print("# rounding of 1 + x, x placed at different k (ulp(1) = 2^-23)")
    for k in (1, 7, 8, 15):
        show(f"1 + 2^-24 (tie) at k={k}", {0: one, k: (u, h)})
        show(f"1 + 3*2^-25 at k={k}", {0: one, k: (3 * 2.0 ** -13, 2.0 ** -12)})
        show(f"1 + 2^-25 at k={k}", {0: one, k: (2.0 ** -13, 2.0 ** -12)})
        show(f"1 - 2^-25 at k={k}", {0: one, k: (-(2.0 ** -13), 2.0 ** -12)})
        show(f"1 - 2^-26 at k={k}", {0: one, k: (-(2.0 ** -13), 2.0 ** -13)})
    print("# two half-ulps: exact sum representable; where are they combined before rounding?")
    for k1, k2 in [(1, 2), (1, 3), (1, 4), (1, 7), (1, 8), (2, 9), (7, 8), (8, 9), (14, 15), (4, 12)]:
        show(f"1 + 2^-24 + 2^-24 at k={k1},{k2}", {0: one, k1: (u, h), k2: (u, h)})
    print("# accumulator C participates how?")
    show("C=1, p1=p2=2^-24", {1: (u, h), 2: (u, h)}, c=1.0)
    show("C=1, p0=2^-24 only (tie)", {0: (u, h)}, c=1.0)
    show("C=1, p0=3*2^-25", {0: (3 * 2.0 ** -13, 2.0 ** -12)}, c=1.0)
    show("C=2^-24, p0=1, p1=2^-24", {0: one, 1: (u, h)}, c=2.0 ** -24)
    print("# many quarter-ulps: extra internal bits?")
    q = (2.0 ** -13, 2.0 ** -12)   # 2^-25
    show("1 + 15 * 2^-25", {0: one, **{k: q for k in range(1, 16)}})
    show("1 + 7 * 2^-25 (k=1..7)", {0: one, **{k: q for k in range(1, 8)}})
    show("1 + 3 * 2^-25 (k=1..3)", {0: one, **{k: q for k in range(1, 4)}})
    show("1 + 2 * 2^-25 (k=1,2)", {0: one, 1: q, 2: q})
    show("1 + 2 * 2^-26 (k=1,2)", {0: one, 1: (2.0 ** -13, 2.0 ** -13), 2: (2.0 ** -13, 2.0 ** -13)})
    show("1 + 4 * 2^-26 (k=1..4)", {0: one, **{k: (2.0 ** -13, 2.0 ** -13) for k in range(1, 5)}})
    show("1 + 8 * 2^-27 (k=1..8)", {0: one, **{k: (2.0 ** -14, 2.0 ** -13) for k in range(1, 9)}})
    show("1 + 2 * 2^-30 + 2^-24", {0: one, 1: (2.0 ** -15, 2.0 ** -15), 2: (2.0 ** -15, 2.0 ** -15), 3: (u, h)})
    print("# large cancellation / alignment")
    show("2^10 + (1+2^-10) ", {0: (1024.0, 1.0), 1: (1.0 + 2.0 ** -10, 1.0)})
    show("2^10 + 2^-14", {0: (1024.0, 1.0), 1: (2.0 ** -7, 2.0 ** -7)})
    show("2^10 - 2^10 + 2^-20", {0: (1024.0, 1.0), 1: (-1024.0, 1.0), 2: (2.0 ** -10, 2.0 ** -10)})
    show("2^10 + 2^-20 - 2^10", {0: (1024.0, 1.0), 1: (2.0 ** -10, 2.0 ** -10), 2: (-1024.0, 1.0)})
    show("2^12 + 2^-14 - 2^12 (k 0,1,2)", {0: (4096.0, 1.0), 1: (2.0 ** -7, 2.0 ** -7), 2: (-4096.0, 1.0)})
    show("2^12 + 2^-14 - 2^12 (k 0,1,9)", {0: (4096.0, 1.0), 1: (2.0 ** -7, 2.0 ** -7), 9: (-4096.0, 1.0)})
    show("2^14 + 2^-14 - 2^14 (k 0,1,2)", {0: (16384.0, 1.0), 1: (2.0 ** -7, 2.0 ** -7), 2: (-16384.0, 1.0)})
    show("2^15*2 + 2^-14 - 2^16 (k 0,1,2)", {0: (32768.0, 2.0), 1: (2.0 ** -7, 2.0 ** -7), 2: (-32768.0, 2.0)})
    show("C=2^12, p0=2^-14, p1=-2^12", {0: (2.0 ** -7, 2.0 ** -7), 1: (-4096.0, 1.0)}, c=4096.0)
    print("# fp16 subnormal inputs")
    show("subnormal a: 2^-24 * 2^10", {0: (2.0 ** -24, 1024.0)})
    show("subnormal a*b: 2^-20 * 2^-20 (product 2^-40)", {0: (2.0 ** -20, 2.0 ** -20)})


if __name__ == "__main__":
    main()
