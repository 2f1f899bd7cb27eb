"""GPU parity, round 3: the o_proj head transform for ANY even head count up to 64 (28: Qwen2.5-7B; 40: Llama-2-13B,
Qwen2.5-14B / 32B; 48, 12, 14, 16), fp16 and bf16, both output orders — the masked kernel of csrc/fq_block.hip.

Pins: {SVD,Inv}SingleTransMatrix.forward outputs the REFERENCE wrote (trans_utils.py:21-25, natural order, fp16 and bf16);
the reference's Triton block_matmul at 16 heads (the one non-{32,64} size its tracer accepts: `tl.arange` wants powers of
two, tests/test_oracle_round3.py); the oracle's quantiser on the kernel's own transform, bit for bit, for the rest."""
import numpy as np
import pytest
import torch

from oracle import fq_oracle as O
from tests.conftest import same_bits

pytestmark = pytest.mark.gpu
P_, F, T, R16, NC0, Q16 = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20
HEADS = [(28, 128), (40, 128), (48, 64), (12, 128), (16, 128), (14, 64)]
BF = torch.bfloat16


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def bits(t):
    return t.detach().contiguous().cpu().view(torch.int16).numpy().view(np.uint16)


def rel_err(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("H,hd", HEADS)
def test_single_trans_matrix_forward_any_head_count(golden, H, hd):
    from flatquant_amd.flatquant import SVDSingleTransMatrix
    g = golden("heads_any")
    tag = f"h{H}x{hd}"
    st = SVDSingleTransMatrix(H)
    st.load_state_dict({"matrix": torch.from_numpy(g[tag + "_matrix"]), "matrix_inv_t": torch.from_numpy(g[tag + "_matrix_inv_t"])})
    st = st.cuda()
    x = dev(g[tag + "_x"])                                                # [T, head_dim, H] fp16
    y = st(x)
    assert y.dtype == torch.float16 and y.shape == x.shape
    assert rel_err(host(y), g[tag + "_y16"]) <= 1e-3
    assert rel_err(host(st(x, inv_t=True)), g[tag + "_y16_inv_t"]) <= 1e-3
    ref = O.single_transform(g[tag + "_x"], g[tag + "_matrix"].astype(np.float16)).astype(np.float16)
    assert np.mean(host(y) != ref) <= 5e-3
    yb = st(x.to(BF))
    assert yb.dtype == BF
    want = O.bf16_from_bits(g[tag + "_ybf_bits"])
    got = O.bf16_from_bits(bits(yb))
    assert np.mean(got != want) < 2e-2 and np.max(np.abs(got - want)) <= 8e-3 * np.max(np.abs(want))


@pytest.mark.parametrize("H,hd", HEADS + [(64, 128), (32, 96)])
@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("transpose_out", [True, False])
def test_block_quant_any_head_count_quant_stage_bit_exact(ops, H, hd, dtype, transpose_out):
    """Every output set of the masked kernel: transform within tolerance of the oracle, quantiser / pack / fake-quant stages
    bit-exact on the kernel's own transform, both output orders, both element types; padding columns never reach the
    extrema (a row of large values would otherwise change the scale) nor the stores (canaries behind the outputs)."""
    lowp = "bf16" if dtype == "bf16" else "f16"
    td = BF if dtype == "bf16" else torch.float16
    gen = torch.Generator().manual_seed(H * 1000 + hd)
    rows = 11
    x = torch.randn(rows, hd, H, generator=gen).to(td)
    x[0] = 0
    x[1] *= 50.0
    Pm = (torch.randn(H, H, generator=gen) / H ** 0.5).to(td)
    sigs = [(0.982, 0.982), (0.6, 0.95)]
    o = ops.block_quant(x.cuda(), Pm.cuda(), sigs, T | P_ | F | R16, transpose_out=transpose_out)
    yshape = (rows, H, hd) if transpose_out else (rows, hd, H)
    assert tuple(o.y.shape) == yshape and o.q[0].shape == (rows, H * hd // 2)
    if dtype == "bf16":
        y = O.bf16_from_bits(bits(o.y)).reshape(rows, -1)
        xin, pin = x.float().numpy(), Pm.float().numpy()
    else:
        y = host(o.y).reshape(rows, -1).astype(np.float32)
        xin, pin = x.numpy(), Pm.numpy()
    y32 = O.single_transform(xin, pin, lowp=lowp)
    if transpose_out:
        y32 = np.swapaxes(y32, -1, -2)
    y32 = y32.reshape(rows, -1)
    den = np.abs(y32).max(axis=1, keepdims=True) + 1e-30
    assert np.max(np.abs(y - y32) / den) <= (5e-3 if dtype == "bf16" else 1e-3)
    for ci, (a, b) in enumerate(sigs):
        ref = O.quant_outputs(y, a, b, round_y_f16=True, lowp=lowp)
        assert np.array_equal(host(o.q[ci]), ref["packed"]), (ci, "packed")
        if dtype == "bf16":
            assert np.array_equal(bits(o.scale[ci]), O.bf16_bits(ref["scale16"]))
            assert np.array_equal(bits(o.fq[ci]).reshape(rows, -1), O.bf16_bits(ref["fq"]))
        else:
            assert np.array_equal(host(o.scale[ci]), ref["scale16"])
            assert same_bits(host(o.fq[ci]).reshape(rows, -1), ref["fq"])
    # the all-low-precision quantiser route
    o2 = ops.block_quant(x.cuda(), Pm.cuda(), [(1.0, 1.0)], F | R16 | Q16, transpose_out=transpose_out)
    ref = O.quant_outputs(y, 1.0, 1.0, round_y_f16=True, quant_f16=True, lowp=lowp)
    if dtype == "bf16":
        assert np.array_equal(bits(o2.fq[0]).reshape(rows, -1), O.bf16_bits(ref["fq"]))
    else:
        assert same_bits(host(o2.fq[0]).reshape(rows, -1), ref["fq"])


def test_block_path_b_at_16_heads(ops, golden):
    """The one non-{32, 64} head count the reference's Triton kernel can be traced at."""
    g = golden("heads_any")
    tag = "h16x128"
    x = dev(g[tag + "_x"])
    s = (float(g[tag + "_b_sig"][0]), float(g[tag + "_b_sig"][1]))
    o = ops.block_quant(x, dev(g[tag + "_P"]), [s], P_ | NC0, transpose_out=True)
    q, qb = O.unpack_i4(host(o.q[0])), O.unpack_i4(g[tag + "_b_packed"])
    assert np.mean(q != qb) <= 1e-3 and np.max(np.abs(q - qb)) <= 1
    sb = g[tag + "_b_scale"].astype(np.float32).reshape(-1)
    assert np.max(np.abs(host(o.scale[0]).astype(np.float32) - sb) / sb) <= 1e-3


@pytest.mark.parametrize("H,hd", [(40, 128), (28, 128)])
def test_block_dyadic_bit_exact_and_no_out_of_bounds(ops, H, hd):
    """Exact arithmetic (every partial sum representable) -> bit-exact vs the oracle end to end, transposed pack included;
    and a many-token launch writes nothing outside its outputs and reads nothing that changes them (guard rows of NaN
    around the input, canaries around the outputs)."""
    rng = np.random.RandomState(H)
    rows = 700
    x = (rng.randint(-16, 17, size=(rows, hd, H)) / 16.0).astype(np.float16)
    Pm = (rng.randint(-4, 5, size=(H, H)) / 8.0).astype(np.float16)
    guard = torch.full((rows + 2, hd, H), float("nan"), dtype=torch.float16, device="cuda")
    guard[1:-1] = dev(x)
    xin = guard[1:-1]
    for tr in (True, False):
        o = ops.block_quant(xin, dev(Pm), [(0.982, 0.9)], P_ | NC0 | T | F, transpose_out=tr)
        ref = O.block_quant(x, Pm, 0.982, 0.9, transpose_out=tr, clamp0=False)
        assert np.array_equal(host(o.y).reshape(rows, -1), ref["y16"])
        assert np.array_equal(host(o.q[0]), ref["packed"])
        assert np.array_equal(host(o.scale[0]), ref["scale16"])
        assert same_bits(host(o.fq[0]).reshape(rows, -1), ref["fq"])


def test_block_refuses_what_it_cannot_run(ops):
    from flatquant_amd import _lib
    x = torch.zeros(2, 128, 66, dtype=torch.float16, device="cuda")
    with pytest.raises(_lib.FqError):
        ops.block_quant(x, torch.zeros(66, 66, dtype=torch.float16, device="cuda"), flags=T)       # C > 64
    x = torch.zeros(2, 80, 32, dtype=torch.float16, device="cuda")
    with pytest.raises(_lib.FqError):
        ops.block_quant(x, torch.zeros(32, 32, dtype=torch.float16, device="cuda"), flags=T)       # R % 32 != 0
