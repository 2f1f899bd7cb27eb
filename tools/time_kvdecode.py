#!/usr/bin/env python3
"""Decode attention over the paged KV cache (ops.kv_batch_decode -> fq_kv_batch_decode_*): us per launch and the cache bytes per second.
    tools/time_kvdecode.py [bsz:seq[:heads[:hd[:page]]]] ...      default: the rows of tools/bench_shapes.py + two caches beyond the MALL
One query token per request; the INT4 cache reads (hd / 2 + 4) bytes per cached row for K and for V, the fp16 cache 2 hd (env FP16=1)."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flatquant_amd import ops  # noqa: E402


def case(bsz, seq, heads=32, hd=128, page=2048, fp16=False, rounds=5, steps=50):
    g = torch.Generator(device="cuda").manual_seed(0)
    n_pg = (seq + page - 1) // page
    if fp16:
        data = torch.randn(bsz * n_pg, 1, 2, heads, page, hd, generator=g, device="cuda").half()
    else:
        data = torch.randint(0, 256, (bsz * n_pg, 1, 2, heads, page, hd // 2), generator=g, device="cuda", dtype=torch.uint8)
    par = (torch.rand(bsz * n_pg, 1, 2, heads, page, 2, generator=g, device="cuda") * 0.2 + 0.05).half()
    indptr = torch.arange(bsz + 1, device="cuda", dtype=torch.int32) * n_pg
    indices = torch.randperm(bsz * n_pg, generator=g, device="cuda").to(torch.int32)
    last = torch.full((bsz,), (seq - 1) % page + 1, device="cuda", dtype=torch.int32)
    q = torch.randn(bsz, heads, hd, generator=g, device="cuda").half()
    split = os.environ.get("SPLIT", "1") != "0"   # SPLIT=0: one workgroup per (request, head) pair at every size
    fn = lambda: ops.kv_batch_decode(q, data, par, indptr, indices, last, 0, seq_hint=seq, split=split)
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    us = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            fn()
        b.record()
        torch.cuda.synchronize()
        us.append(a.elapsed_time(b) * 1e3 / steps)
    t = statistics.median(us)
    byt = bsz * heads * seq * 2 * (hd * 2 if fp16 else hd // 2 + 4)
    print(f"{'fp16' if fp16 else 'INT4'} paged decode attention  bsz={bsz:<4d} seq={seq:<6d} heads={heads} hd={hd} page={page:<5d} {t:9.1f} us  "
          f"{byt / 1e6:8.0f} MB  {byt / t / 1e3:7.0f} GB/s  {byt / t / 1e3 / 8000:.3f} of 8 TB/s  (min {min(us):.1f})", flush=True)


def main():
    fp16 = os.environ.get("FP16") == "1"
    specs = sys.argv[1:] or ["16:2048", "64:2048", "8:8192", "128:4096", "64:2048:32:128:16", "32:16384", "64:2048:32:64"]
    for sp in specs:
        f = [int(v) for v in sp.split(":")]
        case(*f, fp16=fp16) if len(f) <= 5 else None


if __name__ == "__main__":
    main()
