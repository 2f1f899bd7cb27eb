"""GPU parity, round 4: {SVD,Inv}SingleTransMatrix.forward at n = head_dim = 128 / 64 on HIP (fq_single_trans_{f16,bf16}) — the
kcache_trans(q, inv_t=True) / kcache_trans(k) / vcache_trans(v) calls of the fake-quant eval path (llama_utils.py:181-199) —
against the outputs the REFERENCE classes wrote (tests/golden/single128.npz, tools/gen_golden.py r4), and the per-head asymmetric
ActivationQuantizer that follows on the keys (llama_utils.py:124-132)."""
import numpy as np
import pytest
import torch

from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bf(bits_):
    return (np.asarray(bits_, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def from_bits(b):
    return torch.from_numpy(np.ascontiguousarray(b).view(np.int16)).view(BF)


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("tag", ["svd", "inv"])
def test_single_trans_128_vs_reference_outputs(ops, golden, tag):
    g = golden("single128")
    x = g[tag + "_x"]
    for key, mat in (("_y16", "_matrix"), ("_y16_inv_t", "_matrix_inv_t")):
        y = ops.single_trans(dev(x), dev(g[tag + mat].astype(np.float16))).cpu().numpy()
        want = g[tag + key]
        d = np.abs(y.astype(np.float32) - want.astype(np.float32))
        assert y.shape == want.shape and np.mean(y != want) < 2e-2 and d.max() <= 2e-3 * np.abs(want.astype(np.float32)).max(), (tag, key)
        # the oracle's restatement with the matrix pipe's K-order tolerance
        ref = O.single_transform(x, g[tag + mat].astype(np.float16)).astype(np.float16)
        assert np.mean(y != ref) < 2e-2
    xb = dev(x).to(BF)
    for key, mat in (("_ybf_bits", "_matrix"), ("_ybf_inv_t_bits", "_matrix_inv_t")):
        yb = ops.single_trans(xb, torch.from_numpy(g[tag + mat]).to(BF).cuda()).float().cpu().numpy()
        want = bf(g[tag + key])
        ulp = np.maximum(np.abs(want), 1e-30) * 2.0 ** -7
        assert np.all((np.abs(yb - want) <= ulp) | (np.abs(yb - want) <= 4e-3 * np.abs(want).max())), (tag, key)
        assert np.mean(yb != want) < 2e-2


@pytest.mark.parametrize("tag", ["svd", "inv"])
def test_module_mirror_and_key_quantiser(ops, golden, tag):
    """flatquant_amd.flatquant.trans_utils.{SVD,Inv}SingleTransMatrix(128): forward / inv_t through the HIP route (state-dict
    names of the reference's eval mode), then ActivationQuantizer(bits=4, sym=False, lac=True) on the module's own output —
    bit for bit the oracle's restatement of quant_utils.py:33-46,109-117 (itself pinned on the reference's bytes, CPU suite)."""
    from flatquant_amd.flatquant import quant_utils as qu
    from flatquant_amd.flatquant import trans_utils as tu
    g = golden("single128")
    cls = tu.SVDSingleTransMatrix if tag == "svd" else tu.InvSingleTransMatrix
    m = cls(128)
    m.load_state_dict({"matrix": torch.from_numpy(g[tag + "_matrix"]), "matrix_inv_t": torch.from_numpy(g[tag + "_matrix_inv_t"])})
    m = m.cuda()
    x = dev(g[tag + "_x"])
    y = m(x)
    assert y.shape == x.shape and y.dtype == torch.float16
    assert torch.equal(y, ops.single_trans(x, dev(g[tag + "_matrix"].astype(np.float16))))
    assert torch.equal(m(x, inv_t=True), ops.single_trans(x, dev(g[tag + "_matrix_inv_t"].astype(np.float16))))
    q = qu.ActivationQuantizer(bits=4, sym=False, lac=True).cuda()
    q.clip_factor_a_max.data.fill_(float(g["kq_clip"][0]))
    q.clip_factor_a_min.data.fill_(float(g["kq_clip"][1]))
    kq = q(y).cpu().numpy()
    ref = O.rowquant_asym(y.cpu().numpy().reshape(-1, 128), float(g["kq_sig"][0]), float(g["kq_sig"][1])).reshape(kq.shape)
    assert np.array_equal(kq.view(np.uint16), ref.view(np.uint16))
    # against the reference's own quantised keys: the transform differs in summation order on a few elements
    want = g[tag + "_kq16"]
    assert np.mean(kq != want) < 2e-2


@pytest.mark.parametrize("n", [128, 64])
@pytest.mark.parametrize("rows", [1, 31, 33, 1000, 131072])
def test_row_counts_and_both_widths(ops, n, rows):
    gen = torch.Generator().manual_seed(n + rows)
    x = torch.randn(rows, n, generator=gen).half()
    P = (torch.randn(n, n, generator=gen) / n ** 0.5).half()
    y = ops.single_trans(x.cuda(), P.cuda()).cpu()
    ref = (x.float() @ P.float())
    assert float((y.float() - ref).abs().max()) <= 2e-3 * float(ref.abs().max())
    yb = ops.single_trans(x.cuda().to(BF), P.cuda().to(BF)).float().cpu()
    refb = x.to(BF).float() @ P.to(BF).float()
    assert float((yb - refb).abs().max()) <= 1.2e-2 * float(refb.abs().max())


def test_single_trans_argument_errors(ops):
    from flatquant_amd import _lib
    x = torch.zeros(4, 96, dtype=torch.float16, device="cuda")
    P = torch.zeros(96, 96, dtype=torch.float16, device="cuda")
    with pytest.raises(_lib.FqError):
        ops.single_trans(x, P)              # n = 96: FQ_EUNSUPPORTED
    with pytest.raises((ValueError, TypeError, RuntimeError)):
        ops.single_trans(x, P[:64, :64])
