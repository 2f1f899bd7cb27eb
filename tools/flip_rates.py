#!/usr/bin/env python3
"""MEASURED INT4 flip rates of the HIP path against the reference's own outputs (tests/golden/*, written by the imported
reference): per factor pair and clip set, the fraction of digits that differ, the largest |dq|, the largest relative scale
difference. The tests only BOUND these (<= 1e-3, |dq| <= 1); this prints what they are.
    python tools/flip_rates.py > profiles/r03_flip_rates.txt        (on the GPU box)
Path B = deploy Triton kernels (kron_B_*: left factor first, fp32 accumulator / fp16 round trip for M > 64);
path A = flatquant/ fake-quant path (kron_A_*: right factor first, Y rounded to fp16, fp32 quantiser)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flatquant_amd import deploy, ops  # noqa: E402
from flatquant_amd._lib import FQ_OUT_FAKEQUANT, FQ_OUT_PACKED, FQ_OUT_TRANSFORM, FQ_ROUND_Y_F16  # noqa: E402
from oracle import fq_oracle as O  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


print("# reference path B (deploy.functional.online_trans.kronecker_matmul -> Triton kron_matmul, interpreter) vs "
      "flatquant_amd.deploy.functional.kronecker_matmul")
print(f"{'pair':10s} {'clip':>4s} {'digits':>9s} {'flipped':>8s} {'rate':>10s} {'max|dq|':>7s} {'max rel scale diff':>18s}")
for pair in ("64x64", "64x128", "32x64", "112x128"):
    g = np.load(os.path.join(G, f"kron_B_{pair}.npz"))
    bsz, seq = (int(v) for v in g["bsz_seq"])
    x = dev(g["x"]).reshape(bsz, seq, -1)
    for ci in range(3):
        cmax, cmin = (float(v) for v in g["clips"][ci])
        p = deploy.functional.kronecker_matmul(x, [dev(g["L"]), dev(g["R"])], cmax, cmin)
        q = O.unpack_i4(p.quantized_x.cpu().numpy().reshape(bsz * seq, -1))
        qb = O.unpack_i4(g[f"b_packed{ci}"])
        sb = g[f"b_scale{ci}"].astype(np.float32).reshape(-1)
        sd = np.max(np.abs(p.scales_x.cpu().numpy().astype(np.float32).reshape(-1) - sb) / sb)
        n = q.size
        f = int(np.sum(q != qb))
        print(f"{pair:10s} {ci:4d} {n:9d} {f:8d} {f / n:10.2e} {int(np.max(np.abs(q - qb))):7d} {sd:18.2e}")

print("\n# reference path A (InvDecomposeTransMatrix -> ActivationQuantizer(lac), fp16 on CPU) vs ops.kron_quant(FQ_OUT_FAKEQUANT | FQ_ROUND_Y_F16)")
print(f"{'pair':10s} {'clip':>4s} {'digits':>9s} {'flipped':>8s} {'rate':>10s} {'max|dq|':>7s} {'transform: elements != ref':>27s}")
for pair in ("64x64", "64x128", "112x128", "128x224", "86x128", "64x112", "32x64", "56x64"):
    g = np.load(os.path.join(G, f"kron_A_{pair}.npz"))
    x, L, R = dev(g["x"]), dev(g["L"]), dev(g["R"])
    for ci in range(2):
        sig = (float(g["sig"][ci][0]), float(g["sig"][ci][1]))
        o = ops.kron_quant(x, L, R, [sig], FQ_OUT_PACKED | FQ_OUT_TRANSFORM | FQ_ROUND_Y_F16)
        q = O.unpack_i4(o.q[0].cpu().numpy())
        qa = g[f"a16_lac{ci}_q"].astype(np.int32)
        ydiff = float(np.mean(o.y.cpu().numpy() != g[f"a16_lac{ci}_y"]))
        n = q.size
        f = int(np.sum(q != qa))
        print(f"{pair:10s} {ci:4d} {n:9d} {f:8d} {f / n:10.2e} {int(np.max(np.abs(q - qa))):7d} {ydiff:27.2e}")

print("\n# reference path A on bfloat16 (bf16_path_a.npz) vs ops.kron_quant on bf16 tensors, per promotion route")
print(f"{'pair':10s} {'route':>6s} {'digits':>9s} {'flipped':>8s} {'rate':>10s} {'max|dq|':>7s} {'transform: elements != ref':>27s}")
g = np.load(os.path.join(G, "bf16_path_a.npz"))
BF = torch.bfloat16
for pair in ("64x64", "64x112", "32x64", "112x128", "56x64", "128x148"):
    k = "k" + pair
    tb = lambda b: torch.from_numpy(np.ascontiguousarray(b).view(np.int16)).view(BF).cuda()
    x, L, R = tb(g[k + "_x_bits"]), tb(g[k + "_L_bits"]), tb(g[k + "_R_bits"])
    for mode, fl in (("lac32", 0), ("lac16", 0x20 | 0x400), ("nolac", 0x20)):
        sig = tuple(float(v) for v in g[f"{k}_{mode}_sig"]) if mode != "nolac" else (1.0, 1.0)
        o = ops.kron_quant(x, L, R, [sig], FQ_OUT_PACKED | FQ_OUT_TRANSFORM | FQ_ROUND_Y_F16 | fl)
        q = O.unpack_i4(o.q[0].cpu().numpy())
        qa = g[f"{k}_{mode}_q"].astype(np.int32)
        yb = o.y.cpu().view(torch.int16).numpy().view(np.uint16)
        ydiff = float(np.mean(yb != g[f"{k}_{mode}_y_bits"]))
        n = q.size
        f = int(np.sum(q != qa))
        print(f"{pair:10s} {mode:>6s} {n:9d} {f:8d} {f / n:10.2e} {int(np.max(np.abs(q - qa))):7d} {ydiff:27.2e}")
