"""GPU parity for the o_proj block transform (block_matmul.py) — packed output in the reference's transposed order."""
import numpy as np
import pytest
from tests.conftest import same_bits
import torch

from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu
P_, F, T, R16, NC0, Q16 = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def mismatch(a, b):
    return float(np.mean(np.asarray(a).reshape(-1) != np.asarray(b).reshape(-1)))


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("shape", ["128x32", "128x64"])
def test_vs_reference_path_b_golden(ops, golden, shape):
    g = golden(f"block_B_{shape}")
    x, Pm = dev(g["x"]), dev(g["P"])                      # x [bsz, seq, head_dim, num_heads]
    for ci in range(2):
        s = (float(g[f"sig{ci}"][0]), float(g[f"sig{ci}"][1]))
        o = ops.block_quant(x, Pm, [s], P_ | NC0, transpose_out=True)
        q = O.unpack_i4(o.q[0].reshape(-1, o.q[0].shape[-1]).cpu().numpy())
        qb = O.unpack_i4(g[f"b_packed{ci}"])
        assert mismatch(q, qb) <= 1e-3 and np.max(np.abs(q - qb)) <= 1
        sb = g[f"b_scale{ci}"].astype(np.float32)
        assert np.max(np.abs(o.scale[0].cpu().numpy().astype(np.float32) - sb) / sb) <= 1e-3


@pytest.mark.parametrize("R,C", [(128, 32), (128, 64), (64, 32), (32, 64), (96, 32)])
def test_quant_stage_bit_exact_and_transform_tolerance(ops, R, C):
    gen = torch.Generator().manual_seed(R * 100 + C)
    x = torch.randn(9, R, C, generator=gen).half()
    x[0] = 0
    Pm = (torch.randn(C, C, generator=gen) / C ** 0.5).half()
    sigs = [(0.982, 0.982), (0.7, 0.95)]
    o = ops.block_quant(x.cuda(), Pm.cuda(), sigs, T | P_ | F | R16, transpose_out=True)
    y16 = o.y.cpu().numpy().reshape(9, -1)                # [C][R] per token
    for ci, (a, b) in enumerate(sigs):
        ref = O.quant_outputs(y16.astype(np.float32), a, b)
        assert np.array_equal(o.q[ci].cpu().numpy(), ref["packed"])
        assert np.array_equal(o.scale[ci].cpu().numpy(), ref["scale16"])
        assert same_bits(o.fq[ci].cpu().numpy().reshape(9, -1), ref["fq"])
    y32 = np.swapaxes(O.single_transform(x.numpy(), Pm.numpy()), -1, -2).reshape(9, -1)
    assert mismatch(y16, y32.astype(np.float16)) <= 5e-3
    den = np.abs(y32).max(axis=1, keepdims=True) + 1e-30
    assert np.max(np.abs(y16.astype(np.float32) - y32) / den) <= 1e-3
    o2 = ops.block_quant(x.cuda(), Pm.cuda(), [(1.0, 1.0)], F | R16 | Q16, transpose_out=True)
    ref = O.quant_outputs(y16.astype(np.float32), 1.0, 1.0, quant_f16=True)
    assert same_bits(o2.fq[0].cpu().numpy().reshape(9, -1), ref["fq"])


@pytest.mark.parametrize("R,C", [(128, 64), (128, 32)])
def test_many_rows_prefetch_loop_packed_only(ops, R, C):
    """More tokens than resident waves: every wave loops (C = 64: DMA prefetch of its next token + counted wait).
    The packed-only launch must equal the oracle's quantiser on the transform-only launch's output."""
    gen = torch.Generator().manual_seed(R + C)
    rows = 4500
    x = torch.randn(rows, R, C, generator=gen).half().cuda()
    Pm = (torch.randn(C, C, generator=gen) / C ** 0.5).half().cuda()
    y16 = ops.block_quant(x, Pm, flags=T, transpose_out=True).y.cpu().numpy().reshape(rows, -1).astype(np.float32)
    for sig in [(0.982, 0.982), (0.6, 0.8)]:
        o = ops.block_quant(x, Pm, [sig], P_ | R16, transpose_out=True)
        ref = O.quant_outputs(y16, sig[0], sig[1])
        assert np.array_equal(o.q[0].cpu().numpy().reshape(rows, -1), ref["packed"])
        assert np.array_equal(o.scale[0].cpu().numpy().reshape(-1), ref["scale16"].reshape(-1))


def test_dyadic_inputs_bit_exact_vs_oracle(ops):
    """Exact arithmetic: every partial sum representable -> independent of the MFMA accumulation order."""
    rng = np.random.RandomState(0)
    R, C = 128, 32
    x = (rng.randint(-16, 17, size=(6, R, C)) / 16.0).astype(np.float16)
    had = np.array([[1.0]])
    while had.shape[0] < C:
        had = np.block([[had, had], [had, -had]])
    Pm = (had[rng.permutation(C)] * rng.choice([-1.0, 1.0], size=(1, C)) / 8.0).astype(np.float16)
    o = ops.block_quant(dev(x), dev(Pm), [(0.982, 0.9)], P_ | NC0 | T, transpose_out=True)
    ref = O.block_quant(x, Pm, 0.982, 0.9, transpose_out=True, clamp0=False)
    assert np.array_equal(o.y.cpu().numpy().reshape(6, -1), ref["y16"])
    assert np.array_equal(o.q[0].cpu().numpy(), ref["packed"])
    assert np.array_equal(o.scale[0].cpu().numpy(), ref["scale16"])


def test_deploy_online_trans_block_path(ops, golden):
    """deploy.nn.OnlineTrans(decompose=False) -> functional.kronecker_matmul(x 4-D, [P]) -> PackedQuantizedTensor
    with the reference's shapes (deploy/functional/online_trans.py:124-139)."""
    from flatquant_amd import deploy
    g = golden("block_B_128x32")
    x = dev(g["x"])
    tr = deploy.nn.OnlineTrans(32, trans="matmul", decompose=False).cuda()
    tr.right_matrix.copy_(dev(g["P"]))
    p = tr(x)
    bsz, seq, hd, H = x.shape
    assert p.quantized_x.shape == (bsz, seq, hd // 2, H) and p.scales_x.shape == (bsz, 1, seq)
    qb = O.unpack_i4(g["b_packed0"])
    q = O.unpack_i4(p.quantized_x.reshape(bsz * seq, -1).cpu().numpy())
    assert mismatch(q, qb) <= 1e-3
