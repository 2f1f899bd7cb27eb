#!/usr/bin/env python3
"""Launch ONE op of the library N times (for rocprofv3 --pmc / --kernel-trace runs): tools/run_op.py <op> [n]
  ops: kron112 (112x128 packed), kron128x224, kron64x128, kron64x112, kron32x64g (grouped, 131072 rows), kron64fq (fake-quant output), hadq14336, kvk / kvv (KV-cache quantisers), kvdec / kvdec16 / kvdec4 (paged decode attention, INT4 / fp16 pages / INT4 with one copy per group of four read)
       gemmbf6 / gemmi8 (Linear4bit 16384 x 4096 x 4096), (Hadamard 28x512 + Quantizer), rowq4096 / rowq14336 (deploy Quantizer), block32, block64"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flatquant_amd import ops  # noqa: E402
from flatquant_amd._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED, FQ_QUANT_F16, FQ_SIG_F16  # noqa: E402

op, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
SIG = [(0.9820137619972229, 0.9820137619972229)]


def act(rows, d):
    x = torch.randn(rows, d, generator=g, device=dev, dtype=torch.float16)
    x[:, ::97] *= 20
    return x


def mat(k):
    return (torch.randn(k, k, generator=g, device=dev) / k ** 0.5).half()


P = FQ_OUT_PACKED | FQ_NO_CLAMP0
if op == "kron64fq":   # C1: the fake-quant contract at 64 x 64 (FlatQuantizedLinear._eval_forward)
    from flatquant_amd._lib import FQ_OUT_FAKEQUANT, FQ_ROUND_Y_F16
    xs = [act(16384, 4096) for _ in range(2)]
    L, R = mat(64), mat(64)
    fn = lambda i: ops.kron_quant(xs[i % 2], L, R, SIG, FQ_OUT_FAKEQUANT | FQ_ROUND_Y_F16)
elif op.startswith("kron") and not op.endswith("g"):
    M, N = {"kron64": (64, 64), "kron172x64": (172, 64), "kron112": (112, 128), "kron128x224": (128, 224), "kron64x128": (64, 128), "kron64x112": (64, 112),
            "kron86": (86, 128), "kron32x64": (32, 64), "kron128x148": (128, 148), "kron144x192": (144, 192),
            "kron168x176": (168, 176), "kron96": (96, 96), "kron128x144": (128, 144), "kron80x112": (80, 112)}[op]
    rows = 8192 if M * N > 20000 else 16384
    xs = [act(rows, M * N) for _ in range(2)]
    L, R = mat(M), mat(N)
    fn = lambda i: ops.kron_quant(xs[i % 2], L, R, SIG, P)
elif op == "kron32x64g":
    rows, G = 131072, 256
    xs = [act(rows, 2048) for _ in range(2)]
    L, R = mat(32), mat(64)
    offs = torch.linspace(0, rows, G + 1, device=dev).long()
    sm = torch.full((G,), 0.98, device=dev)
    fn = lambda i: ops.kron_quant_grouped(xs[i % 2], L, R, offs, sm, sm, P)
elif op in ("hadq14336", "hadq11008", "hadq11008fwht", "hadq28672", "hadq14336silu"):
    from flatquant_amd.flatquant.hadamard_utils import get_hadK
    width = int(op[4:9])      # (not `n`: that is the launch count of the loop below — round 5 found 11008 launches per counter pass)
    hk, K = get_hadK(width)
    hk = hk.half().to(dev).contiguous()
    xs = [act(16384 if width < 20000 else 8192, width) for _ in range(2)]
    ups = [act(16384, width) for _ in range(2)] if op.endswith("silu") else None   # (the SiLU.mul input: x_gate and up)
    fn = lambda i: ops.hadamard_quant(xs[i % 2], K, hk, SIG[0], fwht_route=op.endswith("fwht"), up=None if ups is None else ups[i % 2])
elif op.startswith("rowq"):
    d = int(op[4:])
    xs = [act(16384, d) for _ in range(2)]
    fn = lambda i: ops.rowquant(xs[i % 2], SIG, FQ_OUT_PACKED | FQ_QUANT_F16 | FQ_SIG_F16)
elif op in ("kvk", "kvv"):   # K transform + asymmetric INT4 pack / V pack, 16384 tokens x 8 heads x 128
    xs = [act(16384 * 8, 128) for _ in range(2)]
    Tm = mat(128)
    fn = (lambda i: ops.kv_quant(xs[i % 2], Tm)) if op == "kvk" else (lambda i: ops.kv_quant(xs[i % 2]))
elif op in ("gemmbf6", "gemmi8"):   # Linear4bit 16384 x 4096 x 4096: the FP6 matrix path / the int8 matrix path
    Mg, Ng, Kg = 16384, 4096, 4096
    xq = torch.randint(0, 256, (Mg, Kg // 2), generator=g, device=dev, dtype=torch.uint8)
    wq = torch.randint(0, 256, (Ng, Kg // 2), generator=g, device=dev, dtype=torch.uint8)
    sx = torch.rand(Mg, generator=g, device=dev).half() * 0.01
    sw = torch.rand(Ng, generator=g, device=dev).half() * 0.01
    if op == "gemmbf6":
        wb, xb = ops.int4_to_bf6(wq, weights=True), ops.int4_to_bf6(xq)
        fn = lambda i: ops.bf6_linear(xb, sx, wb, sw, None, Mg, Ng, Kg)
    else:
        fn = lambda i: ops.int4_linear(xq, sx, wq, sw, None)
elif op in ("kvdec", "kvdec16", "kvdec4"):   # paged decode attention: 64 requests x 2048 cached tokens x 32 heads x 128, INT4 / fp16 pages (570 MB / 2.1 GB per launch)
    bsz, seq, heads, hd, page = 64, 2048, 32, 128, 2048
    f16c = op == "kvdec16"
    data = (torch.randn(bsz, 1, 2, heads, page, hd, generator=g, device=dev).half() if f16c
            else torch.randint(0, 256, (bsz, 1, 2, heads, page, hd // 2), generator=g, device=dev, dtype=torch.uint8))
    par = (torch.rand(bsz, 1, 2, heads, page, 2, generator=g, device=dev) * 0.2 + 0.05).half()
    indptr = torch.arange(bsz + 1, device=dev, dtype=torch.int32)
    indices = torch.arange(bsz, device=dev, dtype=torch.int32)
    last = torch.full((bsz,), seq, device=dev, dtype=torch.int32)
    qd = torch.randn(bsz, heads, hd, generator=g, device=dev).half()
    fn = (lambda i: ops.kv_batch_decode(qd, data, par, indptr, indices, last, 0, kv_copies=4)) if op == "kvdec4" else (lambda i: ops.kv_batch_decode(qd, data, par, indptr, indices, last, 0))   # kvdec4: one copy per group of four read, one workgroup per group
elif op.startswith("block"):
    H = int(op[5:])
    xs = [act(16384, 128 * H).reshape(16384, 128, H) for _ in range(2)]
    Pm = mat(H)
    fn = lambda i: ops.block_quant(xs[i % 2], Pm, SIG, P)
else:
    raise SystemExit(__doc__)
for i in range(n):
    fn(i)
torch.cuda.synchronize()
if os.environ.get("TIME_OP"):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(100):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    print(f"{op}: {e0.elapsed_time(e1) * 10:.1f} us per call")
