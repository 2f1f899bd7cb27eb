import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flatquant_amd import ops
from oracle import fq_oracle as O
g = np.load("tests/golden/kron_A_" + sys.argv[1] + ".npz")
x, L, R = (torch.from_numpy(g[k]).cuda() for k in ("x", "L", "R"))
sigs = [(0.9820137619972229, 0.9820137619972229), (0.9, 0.33)]
for flags in (0x2 | 0x8, 0x4 | 0x2 | 0x8, 0x1 | 0x2 | 0x8, 0x4 | 0x1 | 0x2 | 0x8):
    o = ops.kron_quant(x, L, R, sigs, flags | 0x4)
    y16 = o.y.cpu().numpy()
    for ci, (a, b) in enumerate(sigs):
        ref = O.quant_outputs(y16.astype(np.float32), a, b)
        fq = o.fq[ci].cpu().numpy()
        bad = np.argwhere(fq != ref["fq"])
        print(hex(flags), ci, "bad", len(bad), bad[:6].tolist(), [(float(fq[i, j]), float(ref["fq"][i, j])) for i, j in bad[:3]])
