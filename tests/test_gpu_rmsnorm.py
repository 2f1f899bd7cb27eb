"""RMSNorm (deploy.nn.RMSNorm) alone and fused in front of the 64 x 64 Kronecker transform + INT4 quantiser.

Floating point: the kernel sums the squares in its own order and uses the hardware rsqrt (1 ulp), so the fp16 result
may differ from the reference in the last place on a few elements: tolerance <= 1 fp16 ulp on <= 1 % of the elements
for the normalisation, 1e-3 of the row maximum for the transform behind it (north_star's FP tolerance). The integer
stage stays bit-exact given the kernel's own transform output.
"""
import numpy as np
import pytest

from conftest import BOUND37, flip_ok
import torch

from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu

P, F, T, R16, NC0 = 0x01, 0x02, 0x04, 0x08, 0x10


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


def ulp_diff(a, b):
    """distance in fp16 representable steps (monotone integer mapping of the bit patterns)."""
    def key(v):
        u = np.asarray(v, dtype=np.float16).view(np.uint16).astype(np.int32)
        return np.where(u & 0x8000, -(u & 0x7FFF), u)
    return np.abs(key(a) - key(b))


@pytest.mark.parametrize("d", [4096, 11008, 40])
@pytest.mark.parametrize("eps", [1e-5, 1e-6])
def test_rmsnorm_vs_reference_golden(ops, golden, d, eps):
    g = golden("rmsnorm")
    x, ref = g[f"x_{d}"], g[f"y_{d}_eps{eps:g}"]
    y = ops.rmsnorm(dev(x), eps).cpu().numpy()
    ud = ulp_diff(y, ref)
    assert ud.max() <= 1
    assert np.mean(ud != 0) <= 1e-2
    assert np.array_equal(O.rmsnorm(x, eps), ref)        # the oracle is the reference bit for bit on this fixture


@pytest.mark.parametrize("rows,cols", [(1, 8), (3, 64), (257, 4096), (5, 8192), (2, 16384), (7, 1000)])
def test_rmsnorm_shapes(ops, rows, cols):
    rng = np.random.default_rng(rows * 131 + cols)
    x = (rng.standard_normal((rows, cols)) * rng.uniform(0.01, 30, (rows, 1))).astype(np.float16)
    y = ops.rmsnorm(dev(x), 1e-5).cpu().numpy()
    ud = ulp_diff(y, O.rmsnorm(x, 1e-5))
    assert ud.max() <= 1 and np.mean(ud != 0) <= 1e-2


def test_rmsnorm_zero_rows_and_zero_input(ops):
    assert ops.rmsnorm(torch.empty(0, 4096, dtype=torch.float16, device="cuda")).shape == (0, 4096)
    z = ops.rmsnorm(torch.zeros(4, 4096, dtype=torch.float16, device="cuda"), 1e-5)
    assert torch.count_nonzero(z).item() == 0


def test_module_mirror(ops, golden):
    import flatquant_amd.deploy as deploy
    g = golden("rmsnorm")
    m = deploy.nn.RMSNorm(4096, eps=1e-5)
    y = m(dev(g["x_4096"]).reshape(2, 6, 4096)).cpu().numpy().reshape(12, 4096)
    assert ulp_diff(y, g["y_4096_eps1e-05"]).max() <= 1
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 64, dtype=torch.float16, device="cuda"))


def test_fused_transform_vs_oracle(ops, golden):
    g = golden("rmsnorm")
    x, L, Rm = g["x_4096"], g["L"], g["R"]
    y = ops.rmsnorm_kron_quant(dev(x), 1e-5, dev(L), dev(Rm), flags=T).y.cpu().numpy()
    y32 = O.kron_transform(g["y_4096_eps1e-05"], L, Rm).reshape(y.shape)
    den = np.abs(y32).max(axis=1, keepdims=True)
    assert np.max(np.abs(y.astype(np.float32) - y32) / den) <= 1e-3
    # and agrees with the un-fused pair of launches (the two kernels sum the squares in different orders)
    y2 = ops.kron_quant(ops.rmsnorm(dev(x), 1e-5), dev(L), dev(Rm), flags=T).y.cpu().numpy()
    assert np.max(np.abs(y.astype(np.float32) - y2.astype(np.float32)) / den) <= 1e-3


@pytest.mark.parametrize("sig", [(0.9820137619972229, 0.9525741338729858), (0.9, 0.33), (1.0, 1.0)])
def test_fused_quant_stage_bit_exact(ops, golden, sig):
    g = golden("rmsnorm")
    x, L, Rm = dev(g["x_4096"]), dev(g["L"]), dev(g["R"])
    o = ops.rmsnorm_kron_quant(x, 1e-5, L, Rm, [sig], T | P | R16)
    ref = O.quant_outputs(o.y.cpu().numpy().astype(np.float32), sig[0], sig[1])
    assert np.array_equal(o.q[0].cpu().numpy(), ref["packed"])
    assert np.array_equal(o.scale[0].cpu().numpy(), ref["scale16"])
    o2 = ops.rmsnorm_kron_quant(x, 1e-5, L, Rm, [sig], P | R16)          # packed-only specialisation
    assert np.array_equal(o2.q[0].cpu().numpy(), ref["packed"])
    assert np.array_equal(o2.scale[0].cpu().numpy(), ref["scale16"])


def test_fused_vs_reference_path_b_golden(ops, golden):
    """reference: RMSNorm -> deploy kronecker_matmul (Triton kernel, interpreter) -> packed INT4 + scales."""
    g = golden("rmsnorm")
    s = (float(g["sig"][0]), float(g["sig"][1]))
    o = ops.rmsnorm_kron_quant(dev(g["x_4096"]), 1e-5, dev(g["L"]), dev(g["R"]), [s], P | NC0)
    q, qr = O.unpack_i4(o.q[0].cpu().numpy()), O.unpack_i4(g["b_packed"])
    assert flip_ok(q.astype(np.int32), qr, "rmsnorm+kron64 vs oracle", BOUND37)
    sg = o.scale[0].cpu().numpy().astype(np.float32).reshape(-1)
    assert np.max(np.abs(sg - g["b_scale"].astype(np.float32)) / g["b_scale"].astype(np.float32)) <= 1e-3


def test_online_trans_norm_argument(ops, golden):
    import flatquant_amd.deploy as deploy
    g = golden("rmsnorm")
    t = deploy.nn.OnlineTrans(4096, trans="matmul", decompose=True, lac=True).cuda()
    t.left_matrix.copy_(dev(g["L"])), t.right_matrix.copy_(dev(g["R"]))
    t.clip_factor_a_max.fill_(4.0), t.clip_factor_a_min.fill_(3.0)
    norm = deploy.nn.RMSNorm(4096, 1e-5)
    x = dev(g["x_4096"]).reshape(2, 6, 4096)
    a, b = t(x, norm=norm), t(norm(x))
    qa, qb = O.unpack_i4(a.quantized_x.cpu().numpy()), O.unpack_i4(b.quantized_x.cpu().numpy())
    assert a.quantized_x.shape == b.quantized_x.shape and a.scales_x.shape == b.scales_x.shape
    assert flip_ok(qa.astype(np.int32), qb, "rmsnorm fused vs two launches", BOUND37)
    assert torch.allclose(a.scales_x.float(), b.scales_x.float(), rtol=1e-3, atol=0)


def test_fallback_shapes_and_errors(ops, golden):
    g = golden("kron_A_64x128")
    x, L, Rm = dev(g["x"]), dev(g["L"]), dev(g["R"])
    a = ops.rmsnorm_kron_quant(x, 1e-5, L, Rm, [(0.9, 0.8)], P)
    b = ops.kron_quant(ops.rmsnorm(x, 1e-5), L, Rm, [(0.9, 0.8)], P)
    assert torch.equal(a.q[0], b.q[0]) and torch.equal(a.scale[0], b.scale[0])
    with pytest.raises(Exception):
        ops.rmsnorm(torch.zeros(2, 28672, dtype=torch.float16, device="cuda"))
    with pytest.raises(Exception):
        ops.rmsnorm(torch.zeros(2, 12, dtype=torch.float16, device="cuda"))
    from flatquant_amd._lib import FQ_IN_RMSNORM
    g = golden("kron_A_64x64")
    with pytest.raises(Exception):
        ops.kron_quant(dev(g["x"]), dev(g["L"]), dev(g["R"]), [(1.0, 1.0)], P | FQ_IN_RMSNORM)


def test_fused_forward_shared_factors(ops, golden):
    """q/k/v (and up/gate) transforms share the factor pair: one launch, each output == the module's own forward."""
    import flatquant_amd.deploy as deploy
    g = golden("rmsnorm")
    ts = [deploy.nn.OnlineTrans(4096, trans="matmul", decompose=True, lac=True).cuda() for _ in range(3)]
    L, R = dev(g["L"]), dev(g["R"])
    for i, t in enumerate(ts):
        for name, m in (("left_matrix", L), ("right_matrix", R)):
            del t._buffers[name]
            t.register_buffer(name, m)
        t.clip_factor_a_max.fill_(4.0 - i), t.clip_factor_a_min.fill_(2.0 + 0.5 * i)
    x = dev(g["x_4096"]).reshape(2, 6, 4096)
    outs = deploy.nn.fused_forward(x, ts)
    for t, o in zip(ts, outs):
        r = t(x)
        assert torch.equal(o.quantized_x, r.quantized_x) and torch.equal(o.scales_x, r.scales_x)
    norm = deploy.nn.RMSNorm(4096, 1e-5)
    outs = deploy.nn.fused_forward(x, ts[:2], norm=norm)
    for t, o in zip(ts, outs):
        r = t(x, norm=norm)
        assert torch.equal(o.quantized_x, r.quantized_x) and torch.equal(o.scales_x, r.scales_x)
    other = deploy.nn.OnlineTrans(4096, trans="matmul", decompose=True, lac=True).cuda()
    with pytest.raises(RuntimeError):
        deploy.nn.fused_forward(x, [ts[0], other])


@pytest.mark.parametrize("M,N", [(64, 128), (64, 112), (56, 64), (32, 64), (64, 80)])
def test_fused_rmsnorm_on_the_wave_per_token_pairs(ops, M, N):
    """Round 3: deploy.nn.RMSNorm fused in front of the transform for the pairs of the wave-per-token kernel (the hidden sizes
    8192, 7168, 3584, 2048, 5120), packed output, one to three clip sets. Against the un-fused pair of launches (rmsnorm, then
    kron_quant: the two kernels sum the squares in different orders, so the fp16 normalisation may differ in the last place on a
    few elements) and against the oracle (RMSNorm restated from normalization.py:16-23, then the transform + quantiser): INT4
    digits differ on <= 2e-3 of the elements by one step, scales within 1e-3."""
    rng = np.random.default_rng(M * 1000 + N)
    rows, d = 300, M * N
    x = (rng.standard_normal((rows, d)) * rng.uniform(0.05, 20, (rows, 1))).astype(np.float16)
    x[:, ::97] *= 8
    x[3] = 0
    L = (rng.standard_normal((M, M)) / np.sqrt(M)).astype(np.float16)
    R = (rng.standard_normal((N, N)) / np.sqrt(N)).astype(np.float16)
    sigs = [(0.982, 0.953), (0.7, 0.9), (1.0, 1.0)]
    xd, Ld, Rd = dev(x), dev(L), dev(R)
    fused = ops.rmsnorm_kron_quant(xd, 1e-5, Ld, Rd, sigs, P | NC0)
    again = ops.rmsnorm_kron_quant(xd, 1e-5, Ld, Rd, sigs, P | NC0)                 # prepared workspace
    two = ops.kron_quant(ops.rmsnorm(xd, 1e-5), Ld, Rd, sigs, P | NC0)
    xn = O.rmsnorm(x, 1e-5)
    for ci, (a, b) in enumerate(sigs):
        assert torch.equal(fused.q[ci], again.q[ci]) and torch.equal(fused.scale[ci], again.scale[ci])
        qf, q2 = O.unpack_i4(fused.q[ci].cpu().numpy()), O.unpack_i4(two.q[ci].cpu().numpy())
        assert flip_ok(qf, q2, f"rmsnorm+wave {M}x{N} clip {ci} vs two launches", BOUND37), (M, N, ci)
        sf, s2 = fused.scale[ci].float().cpu().numpy(), two.scale[ci].float().cpu().numpy()
        assert np.max(np.abs(sf - s2) / np.maximum(s2, 1e-30)) <= 1e-3
        ref = O.kron_quant(xn, L, R, a, b, clamp0=False)
        assert flip_ok(qf, ref["q"].astype(np.int32), f"rmsnorm+wave {M}x{N} vs oracle", BOUND37)
    # a single-clip launch larger than one round of resident waves (the persistent loop, counted waits)
    big = dev(np.tile(x, (40, 1)))
    fb = ops.rmsnorm_kron_quant(big, 1e-5, Ld, Rd, [sigs[0]], P | NC0)
    assert torch.equal(fb.q[0][:rows], fused.q[0]) and torch.equal(fb.q[0][-rows:], fused.q[0])
    assert torch.equal(fb.scale[0][:rows], fused.scale[0])
