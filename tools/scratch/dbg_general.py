import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flatquant_amd import ops
from oracle import fq_oracle as O
T = 0x04
for M, N in [(168, 176), (160, 176), (168, 160), (192, 170), (256, 128), (129, 254), (168, 192), (136, 176)]:
    gen = torch.Generator().manual_seed(M * 1000 + N)
    x = torch.randn(3, M * N, generator=gen).half()
    L = (torch.randn(M, M, generator=gen) / M ** 0.5).half()
    R = (torch.randn(N, N, generator=gen) / N ** 0.5).half()
    y = ops.kron_quant(x.cuda(), L.cuda(), R.cuda(), flags=T).y.float().cpu().numpy().reshape(3, M, N)
    y32 = O.kron_transform(x.numpy(), L.numpy(), R.numpy()).reshape(3, M, N)
    err = np.abs(y - y32) / np.abs(y32).max(axis=(1, 2), keepdims=True)
    print(M, N, "max rel err", err.max())
    if err.max() > 1e-3:
        bad = err.max(axis=0) > 1e-3
        rows = np.where(bad.any(axis=1))[0]; cols = np.where(bad.any(axis=0))[0]
        print("   bad rows", rows.min(), rows.max(), len(rows), "bad cols", cols.min(), cols.max(), len(cols))
