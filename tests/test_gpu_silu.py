"""x_up * act_fn(x_gate) (FlatQuantLlamaMLP.forward, deploy/transformers/modeling_llama.py:277-279) alone and formed
inside the down_proj transform launches.

Floating point: SiLU is evaluated in fp32 with the hardware exp2 / reciprocal (1 ulp each) and rounded to fp16, so a few
results in 10^3 land on the neighbouring fp16 value of the reference's SiLU, which the product with x_up can stretch
to two steps: tolerance <= 2 fp16 steps on <= 0.5 % of the elements. The FUSED kernels form the product with the same device function as the standalone kernel, so fused ==
silu_mul() followed by the un-fused launch, bit for bit — which carries every parity property of the un-fused kernels
(tests/test_gpu_kron_generic.py, tests/test_gpu_hadamard.py) over to the fused ones.
"""
import numpy as np
import pytest
import torch

from conftest import hadk_matrix
from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu

P, F, T, R16, NC0 = 0x01, 0x02, 0x04, 0x08, 0x10


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


def steps(a, b):
    def key(v):
        u = np.asarray(v, dtype=np.float16).view(np.uint16).astype(np.int32)
        return np.where(u & 0x8000, -(u & 0x7FFF), u)
    return np.abs(key(a) - key(b))


def gate_up(rows, d, seed):
    g = torch.Generator().manual_seed(seed)
    gate = (torch.randn(rows, d, generator=g) * 3.0).half()
    up = (torch.randn(rows, d, generator=g) * 2.0).half()
    gate[:, ::97] *= 6
    return gate.cuda(), up.cuda()


def test_silu_mul_vs_reference_golden(ops, golden):
    g = golden("silu_mul")
    y = ops.silu_mul(dev(g["gate"]), dev(g["up"])).cpu().numpy()
    ok = np.isfinite(g["x"].astype(np.float32))
    st = steps(y[ok], g["x"][ok])
    assert st.max() <= 2 and np.mean(st != 0) <= 5e-3
    assert np.array_equal(np.isnan(y.astype(np.float32)), np.isnan(g["x"].astype(np.float32)))
    so = steps(O.silu_mul(g["gate"], g["up"])[ok], g["x"][ok])        # the oracle against the same golden
    assert so.max() <= 2 and np.mean(so != 0) <= 1e-3


@pytest.mark.parametrize("shape", [(8,), (3, 40), (2, 5, 14336), (1000, 4096)])
def test_silu_mul_shapes(ops, shape):
    g = torch.Generator().manual_seed(len(shape))
    gate, up = (torch.randn(*shape, generator=g) * 4).half(), torch.randn(*shape, generator=g).half()
    y = ops.silu_mul(gate.cuda(), up.cuda()).cpu().numpy()
    st = steps(y, O.silu_mul(gate.numpy(), up.numpy()))
    assert y.shape == tuple(shape) and st.max() <= 2 and np.mean(st != 0) <= 5e-3


def test_silu_mul_empty_and_errors(ops):
    e = torch.empty(0, 4096, dtype=torch.float16, device="cuda")
    assert ops.silu_mul(e, e).shape == (0, 4096)
    with pytest.raises(Exception):
        ops.silu_mul(torch.zeros(3, 4, dtype=torch.float16, device="cuda"), torch.zeros(3, 4, dtype=torch.float16, device="cuda"))
    with pytest.raises(ValueError):
        ops.silu_mul(torch.zeros(8, dtype=torch.float16, device="cuda"), torch.zeros(16, dtype=torch.float16, device="cuda"))


@pytest.mark.parametrize("M,N", [(112, 128), (86, 128), (128, 224), (96, 128),     # fused (ffn widths)
                                 (64, 64), (64, 128), (56, 64)])                    # two-launch route
@pytest.mark.parametrize("flags", [P, T | P | R16, P | NC0])
def test_fused_kron_equals_two_launches(ops, M, N, flags):
    d = M * N
    gate, up = gate_up(29, d, M + N)
    rng = np.random.default_rng(M * 1000 + N)
    L = dev((rng.standard_normal((M, M)) / np.sqrt(M)).astype(np.float16))
    Rm = dev((rng.standard_normal((N, N)) / np.sqrt(N)).astype(np.float16))
    sigs = [(0.9820137619972229, 0.9525741338729858)]
    a = ops.silu_mul_kron_quant(gate, up, L, Rm, sigs, flags)
    b = ops.kron_quant(ops.silu_mul(gate, up), L, Rm, sigs, flags)
    assert torch.equal(a.q[0], b.q[0]) and torch.equal(a.scale[0], b.scale[0])
    if flags & T:
        assert torch.equal(a.y, b.y)


def test_fused_kron_vs_oracle(ops, golden):
    """end to end against the oracle pipeline (reference SiLU.mul golden -> oracle transform + quantiser)."""
    g = golden("silu_mul")
    M, N = 112, 128
    rng = np.random.default_rng(5)
    L = (rng.standard_normal((M, M)) / np.sqrt(M)).astype(np.float16)
    Rm = (rng.standard_normal((N, N)) / np.sqrt(N)).astype(np.float16)
    s = (0.9820137619972229, 0.9525741338729858)
    ok = np.isfinite(g["x"].astype(np.float32)).all(axis=1)
    o = ops.silu_mul_kron_quant(dev(g["gate"][ok]), dev(g["up"][ok]), dev(L), dev(Rm), [s], T | P)
    y32 = O.kron_transform(g["x"][ok], L, Rm).reshape(o.y.shape)
    den = np.abs(y32).max(axis=1, keepdims=True)
    assert np.max(np.abs(o.y.cpu().numpy().astype(np.float32) - y32) / den) <= 1e-3
    ref = O.kron_quant(g["x"][ok], L, Rm, s[0], s[1])
    q = O.unpack_i4(o.q[0].cpu().numpy())
    assert np.mean(q != ref["q"]) <= 2e-3 and np.max(np.abs(q - ref["q"].astype(np.int32))) <= 1
    sg = o.scale[0].cpu().numpy().astype(np.float32)
    assert np.max(np.abs(sg - ref["scale"]) / ref["scale"]) <= 1e-3


def test_fused_kron_multi_clip_and_ragged(ops):
    M, N = 112, 128
    rng = np.random.default_rng(9)
    L = dev((rng.standard_normal((M, M)) / np.sqrt(M)).astype(np.float16))
    Rm = dev((rng.standard_normal((N, N)) / np.sqrt(N)).astype(np.float16))
    sigs = [(1.0, 1.0), (0.9, 0.4), (0.6, 0.95)]
    for rows in (1, 2, 513, 1031):
        gate, up = gate_up(rows, M * N, rows)
        a = ops.silu_mul_kron_quant(gate, up, L, Rm, sigs, P)
        b = ops.kron_quant(ops.silu_mul(gate, up), L, Rm, sigs, P)
        for ci in range(3):
            assert torch.equal(a.q[ci], b.q[ci]) and torch.equal(a.scale[ci], b.scale[ci])
    e = torch.empty(0, M * N, dtype=torch.float16, device="cuda")
    assert ops.silu_mul_kron_quant(e, e, L, Rm, sigs, P).q[0].shape == (0, M * N // 2)


@pytest.mark.parametrize("n,K", [(512, 1), (4096, 1), (8192, 1), (14336, 28), (28672, 28), (6144, 12), (11008, 172)])
def test_fused_hadamard_equals_two_launches(ops, n, K):
    gate, up = gate_up(37, n, n + K)
    hk = None if K == 1 else torch.from_numpy(hadk_matrix(K)).cuda()
    for sig in [(0.9820137619972229, 0.9820137619972229), (0.7, 0.9)]:
        q, s = ops.hadamard_quant(gate, K, hk, sig, up=up)
        # (every route takes up=: the structured kernel for n = K * 512 / K * 1024, the dense Kronecker launch, the FWHT kernel)
        q2, s2 = ops.hadamard_quant(ops.silu_mul(gate, up), K, hk, sig)
        assert torch.equal(q, q2) and torch.equal(s, s2)
        if ops.had_mfma_supported(n, K):
            q3, s3 = ops.hadamard_quant(gate, K, hk, sig, up=up, route="kron")
            q4, s4 = ops.hadamard_quant(ops.silu_mul(gate, up), K, hk, sig, route="kron")
            assert torch.equal(q3, q4) and torch.equal(s3, s4)


def test_module_arguments(ops):
    import flatquant_amd.deploy as deploy
    gate, up = gate_up(18, 14336, 3)
    gate, up = gate.reshape(2, 9, 14336), up.reshape(2, 9, 14336)
    x = ops.silu_mul(gate, up)
    # FlatQuant: the down_proj transform is the decomposed matmul
    t = deploy.nn.OnlineTrans(14336, trans="matmul", decompose=True, lac=True).cuda()
    for name in ("left_matrix", "right_matrix"):
        b = getattr(t, name)
        b.copy_(torch.randn(b.shape, generator=torch.Generator().manual_seed(5)).cuda() / b.shape[0] ** 0.5)
    t.clip_factor_a_max.fill_(4.0), t.clip_factor_a_min.fill_(3.0)
    a, b = t(gate, up=up), t(x)
    assert torch.equal(a.quantized_x, b.quantized_x) and torch.equal(a.scales_x, b.scales_x)
    # QuaRot-style: Hadamard + Quantizer
    h = deploy.nn.OnlineTrans(14336, trans="had").cuda()
    qz = deploy.nn.Quantizer(lac=True).cuda()
    a, b = h(gate, quantizer=qz, up=up), h(x, quantizer=qz)
    # (round 4: without up= the rotation of 14336 runs the structured matrix-pipe kernel, with up= the dense Kronecker launch that
    #  forms up * silu(gate) in registers: the two round the intermediate at different points — rounding-noise agreement)
    import numpy as np
    from oracle import fq_oracle as O
    qa, qb = O.unpack_i4(a.quantized_x.cpu().numpy().reshape(18, -1)), O.unpack_i4(b.quantized_x.cpu().numpy().reshape(18, -1))
    assert np.mean(qa != qb) <= 2e-3 and np.max(np.abs(qa - qb)) <= 1
    sa, sb = a.scales_x.float().cpu().numpy().reshape(-1), b.scales_x.float().cpu().numpy().reshape(-1)
    assert np.all(np.abs(sa - sb) <= 2e-3 * np.abs(sb))
    assert torch.equal(h(gate, up=up), h(x))            # no quantizer: silu_mul launch + Hadamard launch
    with pytest.raises(RuntimeError):
        t(gate, up=up, norm=deploy.nn.RMSNorm(14336))
