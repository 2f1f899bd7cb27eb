#!/usr/bin/env python3
"""Linear4bit GEMM on the FP6 matrix path: per-launch times (median / min of event pairs) of fq_bf6_linear_f16 for a list of shapes,
and a bit-exactness check against the int8-path kernel. One process per A/B build: FQHIP_OVERLAY=variants/ov_x.so python tools/time_gemm_bf6.py
SHAPES="16384x4096x4096,..." (M x N x K), REPS launches per shape."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flatquant_amd import ops  # noqa: E402

SHAPES = os.environ.get("SHAPES", "16384x4096x4096,16384x14336x4096,16384x4096x14336,2048x4096x4096")
REPS = int(os.environ.get("REPS", "60"))
TAG = os.environ.get("FQHIP_OVERLAY", "default").split("/")[-1]


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    out = []
    for shp in SHAPES.split(","):
        M, N, K = (int(v) for v in shp.split("x"))
        x = torch.randint(0, 256, (M, K // 2), generator=g, device="cuda", dtype=torch.uint8)
        w = torch.randint(0, 256, (N, K // 2), generator=g, device="cuda", dtype=torch.uint8)
        sx = torch.rand(M, generator=g, device="cuda").half() * 0.01
        sw = torch.rand(N, generator=g, device="cuda").half() * 0.01
        wb = ops.int4_to_bf6(w, weights=True)
        xb = ops.int4_to_bf6(x)
        y = ops.bf6_linear(xb, sx, wb, sw, None, M, N, K)
        ok = torch.equal(y, ops.int4_linear(x, sx, w, sw, None))
        for _ in range(30):
            ops.bf6_linear(xb, sx, wb, sw, None, M, N, K)
        ts = []
        for _ in range(REPS):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.bf6_linear(xb, sx, wb, sw, None, M, N, K)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        # back-to-back launches (what a layer sees): REPS launches between one event pair
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            ops.bf6_linear(xb, sx, wb, sw, None, M, N, K)
        e1.record()
        e1.synchronize()
        b2b = e0.elapsed_time(e1) * 1e3 / REPS
        med = statistics.median(ts)
        out.append(f"{shp}: median {med:7.1f} min {min(ts):7.1f} b2b {b2b:7.1f} us ({2.0 * M * N * K / b2b / 1e9:4.2f} Pop/s) exact={ok}")
    print(f"[{TAG}] " + " | ".join(out))


if __name__ == "__main__":
    main()
