#!/usr/bin/env python3
"""Characterise how v_mfma_f32_32x32x16_f16 adds its 16 partial products (run on the GPU box).

fp16 x fp16 products are exact in fp32, so models differ only in how sums are rounded.  Each candidate sums
GROUPS of k-indices exactly (float64) and folds group sums into the fp32 accumulator left to right, one
rounding per fold.  Prints the fraction of the 1024 outputs x trials that match bit-for-bit per model.
"""
import itertools
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flatquant_amd import _probe, ops  # noqa: E402


def model(A, B, C, groups, acc_first=True):
    a, b = A.astype(np.float64), B.astype(np.float64)
    acc = C.astype(np.float32)
    for g in groups:
        part = a[:, g] @ b[g, :]
        acc = (acc.astype(np.float64) + part).astype(np.float32)
    return acc


def chunks(order, size):
    return [order[i:i + size] for i in range(0, len(order), size)]


def main():
    nat = list(range(16))
    inter = [k for pair in zip(range(8), range(8, 16)) for k in pair]   # 0,8,1,9,...
    models = {}
    for name, order in [("nat", nat), ("interleave", inter)]:
        for size in (1, 2, 4, 8, 16):
            models[f"{name}/g{size}"] = chunks(order, size)
    models["nat/rev_g1"] = chunks(nat[::-1], 1)
    # pairwise tree over groups of 4 then sum: approximated by g4 then g16 variants above
    rng = np.random.RandomState(0)
    hits = {k: 0 for k in models}
    total = 0
    extra = {"C_last_exact": 0}
    for trial in range(200):
        scale = 2.0 ** rng.randint(-6, 7, size=(32, 16))
        A = (rng.randn(32, 16) * scale).astype(np.float16)
        B = (rng.randn(16, 32) * 2.0 ** rng.randint(-6, 7, size=(16, 32))).astype(np.float16)
        C = (rng.randn(32, 32) * (0 if trial % 2 else 4)).astype(np.float32)
        D = _probe.probe_mfma(torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(),
                           torch.from_numpy(C).cuda()).cpu().numpy()
        for k, g in models.items():
            hits[k] += int(np.sum(model(A, B, C, g).view(np.uint32) == D.view(np.uint32)))
        # all 16 products summed exactly, C added last with one rounding == nat/g16; also truncation variant
        total += D.size
    for k in models:
        print(f"{k:20s} match {hits[k] / total:.6f}")


if __name__ == "__main__":
    main()
