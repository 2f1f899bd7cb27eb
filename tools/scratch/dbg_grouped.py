import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flatquant_amd import ops
P, F, T, R16 = 1, 2, 4, 8
M, N, rows, n_groups = 32, 64, 4100, 256
rng = np.random.default_rng(rows)
x = (rng.standard_normal((rows, M * N)) * (1 + 5 * (rng.random((rows, 1)) < 0.1))).astype(np.float16)
L = (rng.standard_normal((M, M)) / np.sqrt(M)).astype(np.float16)
R = (rng.standard_normal((N, N)) / np.sqrt(N)).astype(np.float16)
cuts = np.sort(rng.integers(0, rows + 1, size=n_groups - 1))
offs = np.concatenate([[0], cuts, [rows]]).astype(np.int64)
offs[1] = offs[0]; offs[-2] = offs[-1]
smax = rng.uniform(0.3, 1.0, n_groups).astype(np.float32)
smin = rng.uniform(0.3, 1.0, n_groups).astype(np.float32)
smax[n_groups // 2] = 0.05
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
xd, Ld, Rd = d(x), d(L), d(R)
y0 = ops.kron_quant(xd, Ld, Rd, flags=T).y
print("offs monotone:", bool(np.all(np.diff(offs) >= 0)), offs[:5], offs[-5:])
for fl in (P | T | R16, F | T | R16):
    for trial in range(0):
        y2 = ops.kron_quant_grouped(xd, Ld, Rd, d(offs), d(smax), d(smin), fl).y
        bad2 = (y2 != y0).any(dim=1).nonzero().flatten()
        print(hex(fl), "grouped bad rows", bad2.numel(), bad2[:10].tolist())
        if bad2.numel():
            r = int(bad2[0]); diff = (y2[r] != y0[r]).nonzero().flatten()
            print("  row", r, "n diff", diff.numel(), diff[:8].tolist(), y2[r][diff[:4]].tolist(), y0[r][diff[:4]].tolist())
print("---- repeated trials")
for name, fn in (("ungrouped P|T|R16", lambda: ops.kron_quant(xd, Ld, Rd, [(0.9, 0.8)], P | T | R16).y),
                 ("grouped   P|T|R16", lambda: ops.kron_quant_grouped(xd, Ld, Rd, d(offs), d(smax), d(smin), P | T | R16).y),
                 ("grouped   F|T|R16", lambda: ops.kron_quant_grouped(xd, Ld, Rd, d(offs), d(smax), d(smin), F | T | R16).y),
                 ("ungrouped T", lambda: ops.kron_quant(xd, Ld, Rd, flags=T).y)):
    nbad = 0
    for trial in range(400):
        y2 = fn()
        nbad += int((y2 != y0).any(dim=1).sum())
    print(name, "bad rows over 400 trials:", nbad)
