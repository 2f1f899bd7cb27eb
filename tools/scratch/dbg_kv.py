import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from flatquant_amd import ops
from oracle import fq_oracle as O
g = torch.Generator(device="cuda").manual_seed(7)
k = (torch.randn(8, 2048, 8, 128, generator=g, device="cuda") * 2).half()
Tm = (torch.randn(128, 128, generator=g, device="cuda") / 128 ** 0.5).half()
q, p, y = ops.kv_quant(k, Tm, return_transformed=True)
q2, p2 = ops.kv_quant(y)
print("q equal", torch.equal(q, q2), "p equal", torch.equal(p, p2))
dq = (q != q2).reshape(-1, 64); dp = (p != p2).reshape(-1, 2)
print("rows with q diff", int(dq.any(1).sum()), "rows with p diff", int(dp.any(1).sum()))
rows = dq.any(1).nonzero().flatten()[:4].tolist()
for r in rows:
    cols = dq[r].nonzero().flatten()[:4].tolist()
    yr = y.reshape(-1, 128)[r]
    print("row", r, "bytes", cols, "q", q.reshape(-1, 64)[r, cols].tolist(), "q2", q2.reshape(-1, 64)[r, cols].tolist(), "p", p.reshape(-1, 2)[r].tolist(), p2.reshape(-1, 2)[r].tolist(),
          "y", [(float(yr[2 * c]), float(yr[2 * c + 1])) for c in cols])
pk, s, z, _ = O.kv_asym_quant(y.reshape(-1, 128)[rows].cpu().numpy()) if rows else (None,) * 4
if rows:
    print("oracle bytes", [pk[i, dq[r].nonzero().flatten()[:4].cpu().numpy()].tolist() for i, r in enumerate(rows)])
