"""Randomised cross-checks of the round-2 kernels (GPU box; prints mismatches): tools/scratch/fuzz_round2.py [seed]
  * three-group kernel (packed-only, 64 < M <= 128, N = 128) == workgroup-per-token kernel, random M / rows / clips / flags
  * general MFMA kernel: every output set bit-exact against the oracle's quantiser on its own transform, random (M, N)
  * packed-only block kernel == the all-output-sets build (C = 32 / 64, both output orders)
  * grouped launches == one launch per group
  * every quantiser route (magic / clamp / true division) of the specialised kernels vs the oracle; the asymmetric quantiser;
    the row quantisers (fp32 / fp16 contracts, fake-quant) and the KV-cache quantisers
Run it several times in FRESH processes (different seeds): races show on cold launches."""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flatquant_amd import ops  # noqa: E402
from oracle import fq_oracle as O  # noqa: E402

P, F, T, R16, NC0, Q16 = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
random.seed(seed)
g = torch.Generator(device="cuda").manual_seed(seed)
bad = 0


def mats(M, N):
    return ((torch.randn(M, M, generator=g, device="cuda") / M ** 0.5).half(),
            (torch.randn(N, N, generator=g, device="cuda") / N ** 0.5).half())


ONLY = os.environ.get("FUZZ_ONLY", "")
for it in range(40 if ONLY in ("", "trio") else 0):    # three-group kernel vs workgroup-per-token kernel
    M = random.choice([65, 68, 86, 96, 100, 102, 108, 112, 120, 127, 128])
    rows = random.choice([1, 2, 3, 4, 5, 255, 256, 257, 767, 768, 769, 2000, 5000])
    x = (torch.randn(rows, M * 128, generator=g, device="cuda") * random.choice([0.01, 1.0, 30.0])).half()
    if it % 3 == 0:
        x[:, ::53] *= 25
    L, R = mats(M, 128)
    sigs = [(random.uniform(0.05, 1.0), random.uniform(0.05, 1.0)) for _ in range(random.randint(1, 4))]
    fl = random.choice([0, R16, NC0, R16 | NC0])
    a = ops.kron_quant(x, L, R, sigs, P | fl)
    b = ops.kron_quant(x, L, R, sigs, P | T | fl)
    for ci in range(len(sigs)):
        if not (torch.equal(a.q[ci], b.q[ci]) and torch.equal(a.scale[ci], b.scale[ci])):
            bad += 1
            print("trio mismatch", M, rows, sigs[ci], fl)
for it in range(40 if ONLY in ("", "general") else 0):    # general kernel: own transform -> oracle quantiser
    while True:
        M, N = random.randint(1, 256), 2 * random.randint(1, 128)
        if M * N <= 32768 and not (M == 64 and N == 64):
            break
    rows = random.choice([1, 2, 5, 33])
    x = torch.randn(rows, M * N, generator=g, device="cuda").half()
    L, R = mats(M, N)
    sig = (random.uniform(0.3, 1.0), random.uniform(0.3, 1.0))
    o = ops.kron_quant(x, L, R, [sig], T | P | F | R16)
    ref = O.quant_outputs(o.y.cpu().numpy().astype(np.float32), sig[0], sig[1])
    y32 = O.kron_transform(x.cpu().numpy(), L.cpu().numpy(), R.cpu().numpy()).reshape(rows, -1)
    err = np.max(np.abs(o.y.float().cpu().numpy() - y32) / (np.abs(y32).max(axis=1, keepdims=True) + 1e-30))
    if not (np.array_equal(o.q[0].cpu().numpy(), ref["packed"]) and np.array_equal(o.scale[0].cpu().numpy(), ref["scale16"])
            and np.array_equal(o.fq[0].cpu().numpy(), ref["fq"])) or err > 1e-3:
        bad += 1
        print("general mismatch", M, N, rows, err)
    o2 = ops.kron_quant(x, L, R, [sig], P | R16)                    # packed-only launch of the same pair: same bytes
    if not (torch.equal(o2.q[0], o.q[0]) and torch.equal(o2.scale[0], o.scale[0])):
        bad += 1
        print("general packed-only vs all outputs", M, N, rows)
for it in range(30 if ONLY in ("", "block") else 0):    # block kernel: packed-only build vs the generic build
    hd, H = random.choice([32, 64, 96, 128]), random.choice([32, 64])
    rows = random.choice([1, 3, 4, 5, 1023, 1024, 1025, 4100])
    x = (torch.randn(rows, hd, H, generator=g, device="cuda") * random.choice([0.02, 1.0, 20.0])).half()
    Pm = (torch.randn(H, H, generator=g, device="cuda") / H ** 0.5).half()
    sigs = [(random.uniform(0.05, 1.0), random.uniform(0.05, 1.0)) for _ in range(random.randint(1, 3))]
    tr = random.choice([True, False])
    fl = random.choice([0, NC0])
    a = ops.block_quant(x, Pm, sigs, P | fl, tr)
    b = ops.block_quant(x, Pm, sigs, P | T | fl, tr)
    for ci in range(len(sigs)):
        if not (torch.equal(a.q[ci], b.q[ci]) and torch.equal(a.scale[ci], b.scale[ci])):
            bad += 1
            print("block mismatch", hd, H, rows, tr, sigs[ci])
for it in range(15 if ONLY in ("", "grouped") else 0):    # grouped launches (wave kernel, three-group kernel, general kernel)
    M, N = random.choice([(32, 64), (64, 112), (112, 128), (60, 62), (128, 148)])
    G = random.randint(1, 9)
    cuts = sorted(random.randint(0, 300) for _ in range(G - 1))
    offs = torch.tensor([0] + cuts + [300], dtype=torch.int64, device="cuda")
    x = torch.randn(300, M * N, generator=g, device="cuda").half()
    L, R = mats(M, N)
    sm = (torch.rand(G, generator=g, device="cuda") * 0.7 + 0.3).float()
    sn = (torch.rand(G, generator=g, device="cuda") * 0.7 + 0.3).float()
    o = ops.kron_quant_grouped(x, L, R, offs, sm, sn, P | NC0)
    for gi in range(G):
        a0, a1 = int(offs[gi]), int(offs[gi + 1])
        if a0 == a1:
            continue
        one = ops.kron_quant(x[a0:a1].contiguous(), L, R, [(float(sm[gi]), float(sn[gi]))], P | NC0)
        if not (torch.equal(o.q[0][a0:a1], one.q[0]) and torch.equal(o.scale[0][a0:a1], one.scale[0])):
            bad += 1
            print("grouped mismatch", M, N, gi, a0, a1)
for it in range(40 if ONLY in ("", "clamp") else 0):    # quantiser routes of the wave / 64x64 / workgroup kernels vs the oracle on the kernel's own transform
    M, N = random.choice([(64, 64), (64, 128), (64, 112), (64, 80), (32, 64), (56, 64), (112, 128), (128, 224), (86, 128),
                          (128, 148), (120, 148), (144, 192), (168, 176), (128, 144), (80, 112), (32, 48)])
    rows = random.choice([1, 3, 64, 257, 1500])
    x = (torch.randn(rows, M * N, generator=g, device="cuda") * random.choice([0.01, 1.0, 30.0])).half()
    if it % 2 == 0:
        x[:, ::37] *= 40
    L, R = mats(M, N)
    sigs = [(random.choice([1.0, 0.98, 0.9, 0.5, 0.1, 1e-3, 1e-7]), random.choice([1.0, 0.98, 0.9, 0.5, 0.1, 1e-3, 1e-7])) for _ in range(3)]
    both = ops.kron_quant(x, L, R, sigs, P | T | R16)
    y = both.y.cpu().numpy().astype(np.float32)
    only = ops.kron_quant(x, L, R, sigs, P | R16)
    for ci, sg in enumerate(sigs):
        ref = O.quant_outputs(y, sg[0], sg[1])
        for o in (both, only):
            if not (np.array_equal(o.q[ci].cpu().numpy(), ref["packed"]) and np.array_equal(o.scale[ci].cpu().numpy(), ref["scale16"])):
                bad += 1
                print("quantiser route mismatch", M, N, rows, sg, "packed-only" if o is only else "all outputs")
for it in range(40 if ONLY in ("", "asym") else 0):    # asymmetric fake quantiser vs the oracle
    cols = 8 * random.choice([1, 2, 3, 8, 16, 17, 64, 65, 128, 256, 257, 512, 513, 1000, 4096])
    rows = random.choice([1, 2, 63, 64, 65, 1000, 4097])
    x = (torch.randn(rows, cols, generator=g, device="cuda") * random.choice([0.01, 1.0, 100.0])).half()
    if it % 3 == 0:
        x[::2] = x[::2].abs()
    sigs = [(random.uniform(0.05, 1.0), random.uniform(0.05, 1.0)) for _ in range(random.randint(1, 4))]
    f16 = random.random() < 0.5
    o = ops.rowquant(x, sigs, F | 0x800 | (Q16 if f16 else 0))
    for ci, sg in enumerate(sigs):
        ref = O.rowquant_asym(x.cpu().numpy(), sg[0], sg[1], quant_f16=f16)
        if not np.array_equal(o.fq[ci].cpu().numpy().view(np.uint16), ref.view(np.uint16)):
            bad += 1
            print("asym mismatch", rows, cols, sg, f16)
for it in range(30 if ONLY in ("", "rowq") else 0):    # row quantisers (fp32 contract, deploy fp16 contract, fake-quant) vs the oracle
    cols = 8 * random.choice([1, 7, 8, 64, 100, 512, 513, 1376, 1792, 3584])
    rows = random.choice([1, 5, 64, 1000, 4097])
    x = (torch.randn(rows, cols, generator=g, device="cuda") * random.choice([0.01, 1.0, 100.0])).half()
    if it % 2 == 0:
        x[:, ::29] *= 30
    sigs = [(random.choice([1.0, 0.98, 0.9, 0.4]), random.choice([1.0, 0.98, 0.9, 0.4])) for _ in range(random.randint(1, 3))]
    fl, kw = random.choice([(P, {}), (P | NC0, dict(clamp0=False)), (P | Q16, dict(quant_f16=True)), (P | Q16 | 0x400, dict(quant_f16=True, sig_f16=True)),
                            (F, {}), (F | Q16, dict(quant_f16=True)), (P | F, {})])
    o = ops.rowquant(x, sigs, fl)
    for ci, sg in enumerate(sigs):
        ref = O.rowquant(x.cpu().numpy(), sg[0], sg[1], **kw)
        ok = True
        if fl & P:
            ok = ok and np.array_equal(o.q[ci].cpu().numpy(), ref["packed"]) and np.array_equal(o.scale[ci].cpu().numpy(), ref["scale16"])
        if fl & F:
            ok = ok and np.array_equal(o.fq[ci].cpu().numpy().view(np.uint16), ref["fq"].view(np.uint16))   # bits: no -0.0 (rule 10)
        if not ok:
            bad += 1
            print("rowquant mismatch", rows, cols, hex(fl), sg)
for it in range(30 if ONLY in ("", "kv") else 0):    # KV-cache quantisers vs the oracle on the kernel's own transform
    hd = random.choice([64, 128])
    rows = random.choice([1, 31, 32, 33, 1000, 40000])
    x = (torch.randn(rows, hd, generator=g, device="cuda") * random.choice([0.01, 1.0, 50.0])).half()
    if it % 3 == 0:
        x[::3] = x[::3].abs()
    lac = random.random() < 0.5
    clip = (random.uniform(0.3, 1.0), random.uniform(0.3, 1.0))
    if random.random() < 0.5:
        Tm = (torch.randn(hd, hd, generator=g, device="cuda") / hd ** 0.5).half()
        q, p, y = ops.kv_quant(x, Tm, clip=clip, lac=lac, return_transformed=True)
    else:
        q, p = ops.kv_quant(x, clip=clip, lac=lac)
        y = x
    pk, sc, z, _ = O.kv_asym_quant(y.cpu().numpy(), clip[0], clip[1], lac=lac)
    pn = p.cpu().numpy().reshape(-1, 2)
    if not (np.array_equal(q.cpu().numpy().reshape(-1, hd // 2), pk) and np.array_equal(pn[:, 0].view(np.uint16), sc[:, 0].view(np.uint16))
            and np.array_equal(pn[:, 1].view(np.uint16), z[:, 0].view(np.uint16))):
        bad += 1
        print("kv mismatch", rows, hd, lac, clip)
print(f"fuzz_round2 seed {seed}: mismatches {bad}")
