#!/usr/bin/env python3
"""Accuracy of the decode attention's p . v on the matrix pipe (weights rounded to fp16) against the VALU form (fp32 weights) and against an fp64 dense
attention over the de-quantised cache: the error as a fraction of each output row's largest magnitude, for realistic score spreads.
FQHIP_OVERLAY=variants/ov_kv_nopvm.so selects the VALU form."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flatquant_amd import ops

def dense(q, data, par, lens, layer, copies):
    bsz, heads, hd = q.shape
    out = torch.zeros(bsz, heads, hd, dtype=torch.float64, device=q.device)
    for b in range(bsz):
        n = lens[b]
        for h in range(heads):
            ch = (h // copies) * copies
            kq = data[b, layer, 0, ch, :n].to(torch.int32)
            vq = data[b, layer, 1, ch, :n].to(torch.int32)
            kn = torch.stack([kq & 15, kq >> 4], -1).reshape(n, hd).double()
            vn = torch.stack([vq & 15, vq >> 4], -1).reshape(n, hd).double()
            ks, kz = par[b, layer, 0, ch, :n, 0].double(), par[b, layer, 0, ch, :n, 1].double()
            vs, vz = par[b, layer, 1, ch, :n, 0].double(), par[b, layer, 1, ch, :n, 1].double()
            K = kn * ks[:, None] - kz[:, None]
            V = vn * vs[:, None] - vz[:, None]
            s = (K @ q[b, h].double()) / hd ** 0.5
            out[b, h] = torch.softmax(s, 0) @ V
    return out

tag = os.environ.get("FQHIP_OVERLAY", "product").split("/")[-1]
g = torch.Generator(device="cuda").manual_seed(0)
for qscale in (0.5, 2.0, 6.0):          # flat, typical and peaked softmax
    bsz, kv, copies, hd, n = 64, 8, 4, 128, 2048
    data = torch.randint(0, 256, (bsz, 1, 2, kv, 2048, hd // 2), generator=g, device="cuda", dtype=torch.uint8).repeat_interleave(copies, dim=3).contiguous()
    par = torch.stack([torch.rand(bsz, 1, 2, kv, 2048, generator=g, device="cuda") * 0.2 + 0.05,
                       torch.rand(bsz, 1, 2, kv, 2048, generator=g, device="cuda") * 1.5 + 0.3], -1).half().repeat_interleave(copies, dim=3).contiguous()
    indptr = torch.arange(bsz + 1, device="cuda", dtype=torch.int32)
    indices = torch.arange(bsz, device="cuda", dtype=torch.int32)
    lens = [n - 7 * (b % 5) for b in range(bsz)]
    last = torch.tensor(lens, device="cuda", dtype=torch.int32)
    q = (torch.randn(bsz, kv * copies, hd, generator=g, device="cuda") * qscale * 0.1).half()
    o_m = ops.kv_batch_decode(q, data, par, indptr, indices, last, 0, kv_copies=copies).double()     # one workgroup per group (>= 256 pairs)
    o_h = ops.kv_batch_decode(q, data, par, indptr, indices, last, 0).double()                        # a workgroup per head, VALU p . v
    ref = dense(q[:4], data[:4], par[:4], lens[:4], 0, copies)
    rel = lambda a, b: ((a - b).abs().amax(-1) / b.abs().amax(-1)).max().item()
    print(f"[{tag}] q scale {qscale}: merged vs per-head launch {rel(o_m, o_h):.2e};  merged vs fp64 dense {rel(o_m[:4], ref):.2e};  per-head vs fp64 dense {rel(o_h[:4], ref):.2e}", flush=True)
