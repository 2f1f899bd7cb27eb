"""Full-size (8 x 2048 tokens) checks of the rows built beyond the headline kernel, through size-independent
properties — the oracle runs these sizes in minutes, the properties in milliseconds:
  * fused launches == the un-fused pair (bit for bit where the arithmetic is shared, tolerance where the summation order
    is not), determinism, token-permutation invariance;
  * the FP6-path GEMM == the int8-path GEMM (which the small-shape tests pin to the integer oracle);
  * quantiser outputs re-derived from the kernel's own transform on a sample of rows (oracle on 8 rows)."""
import numpy as np
import pytest
import torch

from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu
P, F, T, R16, NC0 = 0x01, 0x02, 0x04, 0x08, 0x10
ROWS = 16384


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


def mats(M, N, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return ((torch.randn(M, M, generator=g, device="cuda") / M ** 0.5).half(),
            (torch.randn(N, N, generator=g, device="cuda") / N ** 0.5).half())


def test_rmsnorm_fused_full_size(ops):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.randn(ROWS, 4096, generator=g, device="cuda") * torch.rand(ROWS, 1, generator=g, device="cuda") * 4).half()
    L, R = mats(64, 64, 2)
    sig = [(0.98, 0.95), (0.9, 0.9), (1.0, 1.0)]
    a = ops.rmsnorm_kron_quant(x, 1e-5, L, R, sig, P | NC0)
    b = ops.rmsnorm_kron_quant(x, 1e-5, L, R, sig, P | NC0)
    for ci in range(3):
        assert torch.equal(a.q[ci], b.q[ci]) and torch.equal(a.scale[ci], b.scale[ci])          # deterministic
    perm = torch.randperm(ROWS, device="cuda", generator=g)
    c = ops.rmsnorm_kron_quant(x[perm].contiguous(), 1e-5, L, R, sig[:1], P | NC0)
    assert torch.equal(c.q[0], a.q[0][perm]) and torch.equal(c.scale[0], a.scale[0][perm])       # tokens are independent
    two = ops.kron_quant(ops.rmsnorm(x, 1e-5), L, R, sig[:1], P | NC0)                             # other summation order
    qa, qb = a.q[0].view(torch.uint8), two.q[0].view(torch.uint8)
    assert (qa != qb).float().mean().item() <= 5e-3
    assert torch.allclose(a.scale[0].float(), two.scale[0].float(), rtol=2e-3, atol=0)
    o = ops.rmsnorm_kron_quant(x[:8].contiguous(), 1e-5, L, R, sig[:1], T | P | R16)
    ref = O.quant_outputs(o.y.cpu().numpy().astype(np.float32), sig[0][0], sig[0][1])
    assert np.array_equal(o.q[0].cpu().numpy(), ref["packed"]) and np.array_equal(o.scale[0].cpu().numpy(), ref["scale16"])


def test_silu_fused_full_size(ops):
    g = torch.Generator(device="cuda").manual_seed(3)
    gate = (torch.randn(ROWS, 14336, generator=g, device="cuda") * 3).half()
    up = (torch.randn(ROWS, 14336, generator=g, device="cuda") * 2).half()
    L, R = mats(112, 128, 4)
    sig = [(0.98, 0.95)]
    a = ops.silu_mul_kron_quant(gate, up, L, R, sig, P | NC0)
    b = ops.kron_quant(ops.silu_mul(gate, up), L, R, sig, P | NC0)
    assert torch.equal(a.q[0], b.q[0]) and torch.equal(a.scale[0], b.scale[0])
    from conftest import hadk_matrix
    hk = torch.from_numpy(hadk_matrix(28)).cuda()
    for route in (None, "kron"):   # the structured kernel (default since the late-round-4 build) and the dense pair: both take up=
        q, s = ops.hadamard_quant(gate, 28, hk, sig[0], up=up, route=route)
        q2, s2 = ops.hadamard_quant(ops.silu_mul(gate, up), 28, hk, sig[0], route=route)
        assert torch.equal(q, q2) and torch.equal(s, s2), route


def test_fp6_gemm_full_size_equals_int8_path(ops):
    g = torch.Generator(device="cuda").manual_seed(5)
    for N, K in ((4096, 4096), (4096, 14336)):
        x = torch.randint(0, 256, (ROWS, K // 2), generator=g, device="cuda", dtype=torch.uint8)
        w = torch.randint(0, 256, (N, K // 2), generator=g, device="cuda", dtype=torch.uint8)
        sx = (torch.rand(ROWS, generator=g, device="cuda") * 0.01).half()
        sw = (torch.rand(N, generator=g, device="cuda") * 0.01).half()
        xb, wb = ops.int4_to_bf6(x), ops.int4_to_bf6(w, weights=True)
        assert torch.equal(ops.bf6_matmul(xb, wb, ROWS, N, K), ops.int4_matmul(x, w))
        assert torch.equal(ops.bf6_linear(xb, sx, wb, sw, None, ROWS, N, K), ops.int4_linear(x, sx, w, sw, None))
        del x, w, xb, wb


def test_kv_quant_full_size(ops):
    g = torch.Generator(device="cuda").manual_seed(7)
    k = (torch.randn(8, 2048, 8, 128, generator=g, device="cuda") * 2).half()
    Tm = (torch.randn(128, 128, generator=g, device="cuda") / 128 ** 0.5).half()
    q, p, y = ops.kv_quant(k, Tm, return_transformed=True)
    q2, p2 = ops.kv_quant(y)                                   # quantising the kernel's own transform: same bits
    assert torch.equal(q, q2) and torch.equal(p, p2)
    deq = ops.kv_dequant(q, p)
    assert torch.all((deq.float() - y.float()).abs() <= 0.55 * p[..., 0:1].float())
    idx = torch.randint(0, 8 * 2048 * 8, (16,), generator=g, device="cuda")
    ys = y.reshape(-1, 128)[idx].cpu().numpy()
    pk, s, z, _ = O.kv_asym_quant(ys)
    assert np.array_equal(q.reshape(-1, 64)[idx].cpu().numpy(), pk)
    assert np.array_equal(p.reshape(-1, 2)[idx][:, 0].cpu().numpy().view(np.uint16), s[:, 0].view(np.uint16))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_c5_full_size_properties(ops, dtype):
    """BASELINE config 5 at its full size (VERDICT r05 missing #4; flatquant/model_tools/deepseekv3_utils.py:427-452): 131,072 routed rows of
    2048 in 256 expert groups (Zipf routing, empty and one-row groups included), per-expert clip pairs, shared AND per-expert 32 x 64
    matrices. Properties: every group's bytes and scales equal the UNGROUPED launch on that group's rows with that group's pair / matrices
    (checked on every non-empty group for the shared transform, on 24 groups for the per-expert one); the launch is repeatable; the
    quantise + pack stage re-derived by the oracle from the kernel's own transform on sample rows."""
    TOK, E, K, d = ROWS, 256, 8, 2048
    g = torch.Generator().manual_seed(5)
    pop = 1.0 / torch.arange(1, E + 1, dtype=torch.float64) ** 0.8
    idx = torch.multinomial(pop[torch.randperm(E, generator=g)].expand(TOK, E), K, replacement=False, generator=g)
    counts = torch.bincount(idx.flatten(), minlength=E)
    counts[7] += counts[3]                      # an EMPTY group and, below, a one-row group (whatever the routing drew)
    counts[3] = 0
    counts[200] += counts[11] - 1
    counts[11] = 1
    offs = torch.zeros(E + 1, dtype=torch.int64)
    offs[1:] = torch.cumsum(counts, 0)
    rows = int(offs[-1])
    assert rows == TOK * K and int((counts == 0).sum()) >= 1
    gd = torch.Generator(device="cuda").manual_seed(6)
    x = (torch.randn(rows, d, generator=gd, device="cuda") * (torch.rand(rows, 1, generator=gd, device="cuda") * 3 + 0.1)).to(dtype)
    L, R = mats(32, 64, 7)
    L, R = L.to(dtype), R.to(dtype)
    smax = torch.sigmoid(torch.rand(E, generator=g) * 4 + 1).float().cuda()
    smin = torch.sigmoid(torch.rand(E, generator=g) * 4 + 1).float().cuda()
    offs_d = offs.cuda()
    a = ops.kron_quant_grouped(x, L, R, offs_d, smax, smin, P | NC0)
    b = ops.kron_quant_grouped(x, L, R, offs_d, smax, smin, P | NC0)
    assert torch.equal(a.q[0], b.q[0]) and torch.equal(a.scale[0], b.scale[0])                       # repeatable
    sm, sn = smax.cpu().tolist(), smin.cpu().tolist()
    for e in range(E):
        r0, r1 = int(offs[e]), int(offs[e + 1])
        if r1 == r0:
            continue
        one = ops.kron_quant(x[r0:r1], L, R, [(sm[e], sn[e])], P | NC0)
        assert torch.equal(one.q[0], a.q[0][r0:r1]) and torch.equal(one.scale[0].reshape(-1), a.scale[0].reshape(-1)[r0:r1]), e
    # the quantiser stage on the launch's own transform (oracle, 8 rows of three groups)
    for e in (0, 11, 255):
        r0, r1 = int(offs[e]), min(int(offs[e + 1]), int(offs[e]) + 8)
        if r1 == r0:
            continue
        o = ops.kron_quant(x[r0:r1], L, R, [(sm[e], sn[e])], T | P | R16 | NC0)      # (the quantiser reads the ROUNDED transform it returns)
        ref = O.quant_outputs(o.y.float().cpu().numpy(), sm[e], sn[e], clamp0=False, lowp="bf16" if dtype == torch.bfloat16 else "f16")
        assert np.array_equal(o.q[0].cpu().numpy(), ref["packed"])
    # per-expert matrices (routed_w2_trans[i], :443-446)
    gm = torch.Generator(device="cuda").manual_seed(8)
    Lg = (torch.randn(E, 32, 32, generator=gm, device="cuda") / 32 ** 0.5).to(dtype)
    Rg = (torch.randn(E, 64, 64, generator=gm, device="cuda") / 8).to(dtype)
    c = ops.kron_quant_grouped(x, Lg, Rg, offs_d, smax, smin, P | NC0)
    c2 = ops.kron_quant_grouped(x, Lg, Rg, offs_d, smax, smin, P | NC0)
    assert torch.equal(c.q[0], c2.q[0]) and torch.equal(c.scale[0], c2.scale[0])
    for e in list(range(0, E, 16)) + [3, 7, 11, 200, 254, 255, 1, 2]:
        r0, r1 = int(offs[e]), int(offs[e + 1])
        if r1 == r0:
            continue
        one = ops.kron_quant(x[r0:r1], Lg[e].contiguous(), Rg[e].contiguous(), [(sm[e], sn[e])], P | NC0)
        assert torch.equal(one.q[0], c.q[0][r0:r1]) and torch.equal(one.scale[0].reshape(-1), c.scale[0].reshape(-1)[r0:r1]), e
