"""GPU parity for the structured matrix-pipe Hadamard rotation (fq_had_mfma.hip, round 4): n = K * 512 or K * 1024, K <= 32.
Oracle: matmul_hadU / matmul_hadU_cuda (hadamard_utils.py:89-110,132-141) restated in oracle/fq_oracle.py and pinned by the
reference-written fixtures tests/golden/had_A.npz; the deploy Quantizer (deploy/nn/quantization.py:13-36) as O.rowquant."""
import numpy as np
import pytest
import torch

from oracle import fq_oracle as O
from tests.conftest import hadk_matrix

pytestmark = pytest.mark.gpu

SHAPES = [(14336, 28), (6144, 12), (10240, 20), (28672, 28), (12288, 12), (20480, 20)]   # K * 512 and K * 1024


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


def exact_rotation(x16, K):
    """The rotation in float64: hadK @ H_P over x.view(rows, K, P), P = n / K, / sqrt(n) — no rounding anywhere."""
    rows, n = x16.shape
    P = n // K
    h = np.ones((1, 1))
    while h.shape[0] < P:
        h = np.block([[h, h], [h, -h]])
    v = x16.astype(np.float64).reshape(rows, K, P) @ h
    v = np.einsum("jk,rkp->rjp", hadk_matrix(K).astype(np.float64), v)
    return v.reshape(rows, n) / np.sqrt(n)


def make_x(rows, n, seed, outliers=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, n, generator=g).half()
    if outliers:
        x[:, ::61] *= 9
    return x


@pytest.mark.parametrize("n,K", SHAPES)
@pytest.mark.parametrize("rows", [1, 2, 3, 37, 1000])
def test_rotation_within_tolerance_of_the_exact_rotation(ops, n, K, rows):
    """north_star tolerance: 1e-3 of the row maximum against the exact rotation (what had_A.npz's y64 holds)."""
    x = make_x(rows, n, n + rows)
    if rows > 2:
        x[2] = 0
    hk = torch.from_numpy(hadk_matrix(K)).cuda()
    y, _, _ = ops.hadamard_mfma(x.cuda(), K, hk)
    ref = exact_rotation(x.numpy(), K)
    den = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1e-30)
    err = np.abs(y.cpu().numpy().astype(np.float64) - ref) / den
    assert np.max(err) <= 1e-3, (n, K, rows, float(np.max(err)))
    if rows > 2:
        assert not y[2].any()   # a zero token stays zero


@pytest.mark.parametrize("n,K", SHAPES)
def test_quantizer_stage_bit_exact_on_the_launch_own_rotation(ops, n, K):
    """Packed digits and scales == the deploy Quantizer's fp16 arithmetic applied to the fp16 rotation the SAME launch returns;
    the packed-only and the rotation-only instantiations return the same bytes; clamp and no-clamp routes."""
    rows = 53
    x = make_x(rows, n, n + 7)
    x[5] = 0
    x[6, :] = x[6, :].abs()          # single-signed row
    hk = torch.from_numpy(hadk_matrix(K)).cuda()
    for sig in [(0.9820137619972229, 0.9820137619972229), (0.7, 0.9), (1.0, 1.0), (0.55, 0.5)]:
        y, q, s = ops.hadamard_mfma(x.cuda(), K, hk, sig)
        rq = O.rowquant(y.cpu().numpy(), *sig, clamp0=True, quant_f16=True, sig_f16=True)
        assert np.array_equal(q.cpu().numpy(), rq["packed"]), (n, K, sig)
        assert np.array_equal(s.cpu().numpy(), rq["scale16"]), (n, K, sig)
        _, q2, s2 = ops.hadamard_mfma(x.cuda(), K, hk, sig, want_y=False)
        assert torch.equal(q2, q) and torch.equal(s2, s)
        y2, _, _ = ops.hadamard_mfma(x.cuda(), K, hk)
        assert torch.equal(y2, y)
        qh, sh = ops.hadamard_quant(x.cuda(), K, hk, sig)          # the default route of these shapes
        assert torch.equal(qh, q) and torch.equal(sh.reshape(-1), s)


@pytest.mark.parametrize("n,K", SHAPES)
@pytest.mark.parametrize("rows", [1, 5, 130, 777])
def test_silu_mul_input_equals_the_two_launch_sequence(ops, n, K, rows):
    """fq_silu_mul_hadamard_quant_mfma_f16: x_gate and up in, the rotation's input fp16(up * fp16(silu(gate))) formed inside the launch
    (modeling_llama.py:277-279 in front of down_proj) — the same bytes as the structured launch on fq_silu_mul_f16's output."""
    g = torch.Generator().manual_seed(n + rows)
    gate = (torch.randn(rows, n, generator=g) * 2).half().cuda()
    up = torch.randn(rows, n, generator=g).half().cuda()
    gate[:, ::97] *= 6
    if rows > 2:
        gate[2] = 0
    hk = torch.from_numpy(hadk_matrix(K)).cuda()
    x = ops.silu_mul(gate, up)
    for sig in [(0.9820137619972229, 0.9820137619972229), (0.7, 0.9)]:
        _, q, s = ops.hadamard_mfma(gate, K, hk, sig, want_y=False, up=up)
        _, q2, s2 = ops.hadamard_mfma(x, K, hk, sig, want_y=False)
        assert torch.equal(q, q2) and torch.equal(s, s2), (n, K, rows, sig)
        qd, sd = ops.hadamard_quant(gate, K, hk, sig, up=up)        # the default route of these shapes
        assert torch.equal(qd, q) and torch.equal(sd.reshape(-1), s)
    for _ in range(5):                                                # (claims and meetings are timing-dependent, results are not)
        _, q3, s3 = ops.hadamard_mfma(gate, K, hk, (0.7, 0.9), want_y=False, up=up)
        assert torch.equal(q3, q) and torch.equal(s3, s)


@pytest.mark.parametrize("n,K", SHAPES)
def test_plain_quantizer_in_the_same_launch(ops, n, K):
    """fq_hadamard_quantizer_mfma_f16: the rotation in front of deploy.nn.Quantizer(input_clip_ratio, lac=False) — the pair the
    reference's options.trans == "had" model builds (modeling_llama.py:244-252, quantization.py:30) — as one launch: digits and scales
    == the Quantizer module (fq_rowquant_f16 with FQ_RATIO_POST, pinned against torch in tests/test_gpu_quant.py) applied to the
    rotation the SAME launch returns; an all-zero token keeps scale 0; the SiLU.mul input; the module route and its shapes."""
    import flatquant_amd.deploy as deploy
    rows = 53
    x = make_x(rows, n, n + 3).cuda()
    x[5] = 0
    hk = torch.from_numpy(hadk_matrix(K)).cuda()
    y0 = ops.hadamard_mfma(x, K, hk)[0]
    g = torch.Generator().manual_seed(n)
    gate, up = (torch.randn(rows, n, generator=g) * 2).half().cuda(), torch.randn(rows, n, generator=g).half().cuda()
    for ratio in (1.0, 0.9, 0.83):
        q, s, y = ops.hadamard_quantizer_mfma(x, K, hk, ratio, want_y=True)
        assert torch.equal(y, y0)
        p = deploy.nn.Quantizer(input_clip_ratio=ratio).cuda()(y)
        assert torch.equal(p.quantized_x, q) and torch.equal(p.scales_x.reshape(-1), s), (n, K, ratio)
        assert float(s[5]) == 0.0 and not q[5].any()
        pk, so = O.quantizer_plain(y.cpu().numpy(), ratio)               # the oracle's restatement of quantization.py:30 + quant.cu
        assert np.array_equal(q.cpu().numpy(), pk) and np.array_equal(s.cpu().numpy(), so), (n, K, ratio)
        q2, s2, _ = ops.hadamard_quantizer_mfma(x, K, hk, ratio)          # the packed-only instantiation
        assert torch.equal(q2, q) and torch.equal(s2, s)
        q3, s3, _ = ops.hadamard_quantizer_mfma(gate, K, hk, ratio, up=up)
        q4, s4, _ = ops.hadamard_quantizer_mfma(ops.silu_mul(gate, up), K, hk, ratio)
        assert torch.equal(q3, q4) and torch.equal(s3, s4)
    t = deploy.nn.OnlineTrans(n, trans="had").cuda()
    qz = deploy.nn.Quantizer(lac=False).cuda()
    x3 = x[:52].reshape(2, 26, n)
    fused = t(x3, quantizer=qz)
    if t.rem_dim != K:     # get_hadK prefers another factor for this width (10240 = 40 x 256): not a shape of the structured kernel,
        assert torch.is_tensor(fused)                       # the module returns the rotation and the caller's Quantizer follows
        two = deploy.nn.FusedSequential(t, qz)(x3)
        assert isinstance(two, deploy.PackedQuantizedTensor) and two.scales_x.shape == (2, 1, 26)
        return
    assert isinstance(fused, deploy.PackedQuantizedTensor)
    assert fused.scales_x.shape == (2, 1, 26) and fused.quantized_x.shape == (2, 26, n // 2)
    q, s, _ = ops.hadamard_quantizer_mfma(x[:52].contiguous(), K, hk, 1.0)
    assert torch.equal(fused.quantized_x.reshape(52, -1), q) and torch.equal(fused.scales_x.reshape(-1), s)
    seq = deploy.nn.FusedSequential(t, qz)
    again = seq(x3)
    assert torch.equal(again.quantized_x, fused.quantized_x) and torch.equal(again.scales_x, fused.scales_x)


@pytest.mark.parametrize("n,K", SHAPES)
def test_agrees_with_the_fwht_route_to_rounding_noise(ops, n, K):
    """Against the bit-identical route (register FWHT + K-factor, then the Quantizer): digits within +-1 on <= 1e-3 of the elements
    (SURVEY section 7's bar — round 4 allowed 2e-3; MEASURED on exactly this data, profiles/r05_flip_rates.txt: 4.8e-4 .. 8.4e-4 per
    width, 0 where the structured route is not taken), scales within one fp16 step, the rotation within 1e-3 of the row maximum
    (measured 7.2e-4 .. 9.5e-4). 2048 rows: the data of tools/flip_rates_had.py, so the bound is the measurement, not a sample of it."""
    rows = 2048
    g = torch.Generator().manual_seed(n)
    x = torch.randn(rows, n, generator=g).half()
    x[:, ::61] *= 9
    hk = torch.from_numpy(hadk_matrix(K)).cuda()
    for sig in [(0.9820137619972229, 0.9820137619972229), (0.7, 0.9)]:
        q, s = ops.hadamard_quant(x.cuda(), K, hk, sig)
        qf, sf = ops.hadamard_quant(x.cuda(), K, hk, sig, fwht_route=True)
        qa, qb = O.unpack_i4(q.cpu().numpy().reshape(rows, -1)), O.unpack_i4(qf.cpu().numpy().reshape(rows, -1))
        assert np.mean(qa != qb) <= 1e-3 and np.max(np.abs(qa - qb)) <= 1, (n, K, sig, float(np.mean(qa != qb)))
        sa, sb = s.float().cpu().numpy().reshape(-1), sf.float().cpu().numpy().reshape(-1)
        assert np.all(np.abs(sa - sb) <= 2e-3 * np.maximum(np.abs(sb), 1e-6))          # (one fp16 step of the scale)
    y = ops.hadamard_mfma(x.cuda(), K, hk)[0]
    yf = ops.hadamard(x.cuda(), K, hk, fwht_route=True)
    den = yf.float().abs().amax(dim=1, keepdim=True)
    assert float(((y.float() - yf.float()).abs() / den).max()) <= 1e-3


@pytest.mark.parametrize("n", [14336, 28672])
def test_reference_fixture(ops, golden, n):
    """matmul_hadU's own output on the reference's fixture (tests/golden/had_A.npz, written by tools/gen_golden.py)."""
    g = golden("had_A")
    x = torch.from_numpy(g[f"x_{n}"]).cuda()
    hk = torch.from_numpy(hadk_matrix(28)).cuda()
    y = ops.hadamard_mfma(x, 28, hk)[0].cpu().numpy()
    y64 = g[f"y64_{n}"]
    den = np.abs(y64).max(axis=1, keepdims=True)
    assert np.max(np.abs(y.astype(np.float64) - y64) / den) <= 1e-3


@pytest.mark.parametrize("n,rows", [(14336, 4099), (28672, 2051)])
def test_in_place_partition_and_repeatability(ops, n, rows):
    """y_out == x is allowed; any split of the rows over launches returns the same bytes (rows are independent); 20 repeated
    full-size launches are bit-identical (the token claims and meetings are timing-dependent, the results must not be)."""
    K = 28
    x = make_x(rows, n, 5, outliers=False).cuda()
    hk = torch.from_numpy(hadk_matrix(K)).cuda()
    sig = (0.83, 0.64)
    y, q, s = ops.hadamard_mfma(x, K, hk, sig)
    for _ in range(20):
        y2, q2, s2 = ops.hadamard_mfma(x, K, hk, sig)
        assert torch.equal(y2, y) and torch.equal(q2, q) and torch.equal(s2, s)
    parts = [ops.hadamard_mfma(x[a:b].contiguous(), K, hk, sig) for a, b in [(0, 1), (1, 770), (770, rows)]]
    assert torch.equal(torch.cat([p[0] for p in parts]), y) and torch.equal(torch.cat([p[1] for p in parts]), q)
    z = x.clone()
    from flatquant_amd._lib import check, lib
    import ctypes
    check(lib.fq_hadamard_quant_mfma_f16(z.data_ptr(), rows, n, K, hk.data_ptr(), ctypes.c_float(float(1.0 / torch.tensor(n).sqrt())),
                                         ctypes.c_float(1.0), ctypes.c_float(1.0), None, None, z.data_ptr(),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert torch.equal(z, y)


def test_unsupported_shapes_are_refused(ops):
    from flatquant_amd import _lib
    hk = torch.from_numpy(hadk_matrix(28)).cuda()
    x = torch.zeros(4, 57344, dtype=torch.float16, device="cuda")      # 28 * 2048: neither K * 512 nor K * 1024
    q = torch.empty(4, 28672, dtype=torch.uint8, device="cuda")
    s = torch.empty(4, dtype=torch.float16, device="cuda")
    import ctypes
    rc = _lib.lib.fq_hadamard_quant_mfma_f16(x.data_ptr(), 4, 57344, 28, hk.data_ptr(), ctypes.c_float(1.0), ctypes.c_float(1.0),
                                            ctypes.c_float(1.0), q.data_ptr(), s.data_ptr(), None, None)
    assert rc == _lib.FQ_EUNSUPPORTED
    rc = _lib.lib.fq_hadamard_quant_mfma_f16(x.data_ptr(), 4, 14336, 28, hk.data_ptr(), ctypes.c_float(1.0), ctypes.c_float(1.0),
                                            ctypes.c_float(1.0), None, None, None, None)
    assert rc == _lib.FQ_EINVAL
