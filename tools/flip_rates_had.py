#!/usr/bin/env python3
"""MEASURED digit-flip rates of the matrix-pipe Hadamard routes against the bit-identical route (register FWHT + K-factor, then the
deploy Quantizer) — VERDICT r04 weak #1: tests/test_gpu_had_mfma.py bounded them by 2e-3 where SURVEY §7's bar is 1e-3; this prints
what they are, per width, so that the bound can be the measurement (GPU box only):
    python tools/flip_rates_had.py > profiles/r05_flip_rates.txt
Routes: the structured kernel (n = K * 512 / K * 1024: 14336, 28672, ...), the dense Kronecker launch of the rotation (11008 = 172 x 64
on two-wave token groups since round 5, 8960, 5120). 2048 rows of LLM-like data (randn, every 61st channel x 9), both clip sets of the tests."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flatquant_amd import ops  # noqa: E402
from flatquant_amd.flatquant.hadamard_utils import get_hadK  # noqa: E402
from oracle import fq_oracle as O  # noqa: E402
from tests.conftest import hadk_matrix  # noqa: E402

ROWS = 2048
print(f"{'n':>6s} {'K':>4s} {'sig':>12s} {'digits':>10s} {'flipped':>8s} {'rate':>10s} {'max|dq|':>7s} {'rows w/ other scale':>19s} {'max rel scale diff':>18s} {'max |dy| / row max':>18s}")
for n in (14336, 6144, 10240, 28672, 12288, 20480, 11008, 8960, 5120):
    _, K = get_hadK(n)
    hk = torch.from_numpy(hadk_matrix(K)).cuda() if K > 1 else None
    g = torch.Generator().manual_seed(n)
    x = torch.randn(ROWS, n, generator=g).half()
    x[:, ::61] *= 9
    xc = x.cuda()
    y = ops.hadamard(xc, K, hk).float()
    yf = ops.hadamard(xc, K, hk, fwht_route=True).float()
    dy = float(((y - yf).abs() / yf.abs().amax(dim=1, keepdim=True)).max())
    for sig in [(0.9820137619972229, 0.9820137619972229), (0.7, 0.9)]:
        q, s = ops.hadamard_quant(xc, K, hk, sig)
        qf, sf = ops.hadamard_quant(xc, K, hk, sig, fwht_route=True)
        qa, qb = O.unpack_i4(q.cpu().numpy().reshape(ROWS, -1)), O.unpack_i4(qf.cpu().numpy().reshape(ROWS, -1))
        sa, sb = s.float().cpu().numpy().reshape(-1), sf.float().cpu().numpy().reshape(-1)
        nf = int(np.sum(qa != qb))
        print(f"{n:6d} {K:4d} {sig[0]:5.3f}/{sig[1]:5.3f} {qa.size:10d} {nf:8d} {nf / qa.size:10.2e} {int(np.max(np.abs(qa - qb))):7d} "
              f"{int(np.sum(sa != sb)):19d} {float(np.max(np.abs(sa - sb) / np.maximum(np.abs(sb), 1e-6))):18.2e} {dy:18.2e}", flush=True)
