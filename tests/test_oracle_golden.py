"""The CPU oracle against golden vectors produced by the reference itself (tools/gen_golden.py).

These tests pin the oracle (oracle/fq_oracle.py) — they run without a GPU.  What "match" means per stage:
  * quantise/pack stage, given the reference's transformed activation: BIT-EXACT (q, scale, fake-quant).
  * transform stage vs reference path A (torch CPU fp16 matmul): fp32 accumulation ORDER differs (BLAS
    blocking), which moves isolated fp16 roundings by one ulp -> <= 1% of elements off by one fp16 ulp.
  * reference path B, non-split Triton kernel, evaluated with its own association (left first): BIT-EXACT.
  * exact-arithmetic (dyadic) fixtures: BIT-EXACT against both reference paths, either association.
"""
import numpy as np
import pytest

from oracle import fq_oracle as O
from tests.conftest import hadk_matrix, same_bits

KRON_A_SHAPES = ["64x64", "64x128", "112x128", "128x224", "86x128", "64x112", "32x64", "56x64"]


def mismatch(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.mean(a.reshape(-1) != b.reshape(-1)))


def test_decompose_dim(golden):
    g = golden("decompose_dim")
    for n, dims in zip(g["n"], g["dims"]):
        assert O.get_decompose_dim(int(n)) == tuple(int(v) for v in dims)


def test_pack_roundtrip_all_bytes(golden):
    g = golden("pack_roundtrip")
    assert np.array_equal(O.pack_i4(g["q"]), g["packed"])
    assert np.array_equal(O.unpack_i4(g["packed"]), g["unpacked"])
    assert np.array_equal(O.unpack_i4(O.pack_i4(g["q"])), g["q"].astype(np.int32))
    assert len(np.unique(g["packed"])) == 256


@pytest.mark.parametrize("shape", KRON_A_SHAPES)
def test_quant_stage_bit_exact_vs_path_a(golden, shape):
    g = golden(f"kron_A_{shape}")
    rows = g["x"].shape[0]
    for ci in range(2):
        y = g[f"a16_lac{ci}_y"].reshape(rows, -1).astype(np.float32)
        o = O.quant_outputs(y, g["sig"][ci][0], g["sig"][ci][1])       # lac: fp32 arithmetic
        assert np.array_equal(o["scale"], g[f"a16_lac{ci}_scale"])
        assert np.array_equal(o["q"], g[f"a16_lac{ci}_q"].reshape(rows, -1))
        assert same_bits(o["fq"], g[f"a16_lac{ci}_fq"].reshape(rows, -1))
    y = g["a16_lac0_y"].reshape(rows, -1).astype(np.float32)
    o = O.quant_outputs(y, 1.0, 1.0, quant_f16=True)                   # no lac: fp16 arithmetic
    assert np.array_equal(o["scale"], g["a16_nolac_scale"].astype(np.float32))
    assert np.array_equal(o["q"], g["a16_nolac_q"].reshape(rows, -1))
    assert same_bits(o["fq"], g["a16_nolac_fq"].reshape(rows, -1))


@pytest.mark.parametrize("shape", KRON_A_SHAPES)
def test_transform_vs_path_a(golden, shape):
    g = golden(f"kron_A_{shape}")
    rows = g["x"].shape[0]
    y32 = O.kron_transform(g["x"], g["L"], g["R"]).reshape(rows, -1)
    ref16 = g["a16_lac0_y"].reshape(rows, -1)
    ref32 = g["a32_lac0_y"].reshape(rows, -1)                          # path A run in fp32 end to end
    y16 = y32.astype(np.float16)
    bad = y16 != ref16
    assert bad.mean() <= 0.01                                          # isolated 1-ulp fp16 flips only
    den = np.abs(ref32).max(axis=1, keepdims=True)
    assert np.max(np.abs(y16.astype(np.float32) - ref16.astype(np.float32)) / den) <= 1e-3
    assert np.max(np.abs(y32 - ref32) / den) <= 1e-3                   # north-star tolerance vs fp32 path A
    for ci in range(2):
        o = O.kron_quant(g["x"], g["L"], g["R"], g["sig"][ci][0], g["sig"][ci][1], round_y_f16=True)
        qa = g[f"a16_lac{ci}_q"].reshape(rows, -1)
        assert mismatch(o["q"], qa) <= 1e-3
        assert np.max(np.abs(o["q"].astype(int) - qa)) <= 1
        assert np.max(np.abs(o["scale"] - g[f"a16_lac{ci}_scale"]) / g[f"a16_lac{ci}_scale"]) <= 1e-3


def test_diag_scale_vs_path_a(golden):
    g = golden("kron_A_diag_64x64")
    o = O.kron_quant(g["x"], g["L"], g["R"], g["sig"][0], g["sig"][1], diag16=g["diag"], round_y_f16=True)
    assert mismatch(o["y16"], g["y"]) <= 0.01
    assert mismatch(o["q"], g["q"]) <= 1e-3
    o2 = O.quant_outputs(g["y"].reshape(4, -1).astype(np.float32), g["sig"][0], g["sig"][1])
    assert np.array_equal(o2["q"], g["q"].reshape(4, -1)) and np.array_equal(o2["scale"], g["scale"])


@pytest.mark.parametrize("shape", ["64x64", "64x128", "32x64"])
def test_exact_fixtures_bit_exact_both_paths(golden, shape):
    g = golden(f"exact_{shape}")
    rows = g["x"].shape[0]
    for ci in range(2):
        s = g[f"sig{ci}"]
        for left_first in (False, True):
            a = O.kron_quant(g["x"], g["L"], g["R"], s[0], s[1], round_y_f16=True, left_first=left_first)
            assert np.array_equal(a["y16"], g[f"a_y{ci}"].reshape(rows, -1))
            assert np.array_equal(a["q"], g[f"a_q{ci}"].reshape(rows, -1))
            assert np.array_equal(a["scale"], g[f"a_scale{ci}"])
            assert same_bits(a["fq"], g[f"a_fq{ci}"].reshape(rows, -1))
            b = O.kron_quant(g["x"], g["L"], g["R"], s[0], s[1], clamp0=False, left_first=left_first)
            assert np.array_equal(b["packed"], g[f"b_packed{ci}"])
            assert np.array_equal(b["scale16"], g[f"b_scale{ci}"])


@pytest.mark.parametrize("shape", ["64x64", "64x128", "32x64"])
def test_path_b_nonsplit_bit_exact_with_its_association(golden, shape):
    g = golden(f"kron_B_{shape}")
    for ci in range(3):
        s = g["sig"][ci]
        o = O.kron_quant(g["x"], g["L"], g["R"], s[0], s[1], clamp0=False, left_first=True)
        assert np.array_equal(o["packed"], g[f"b_packed{ci}"])
        assert np.array_equal(o["scale16"], g[f"b_scale{ci}"])


def test_path_b_split_kernel(golden):
    """M > 64: the Triton path stores fp16 and re-quantises with zero padding (kron_matmul.py:73-88,133-189)
    == left-first association + fp16 rounding of Y + zero joining the min/max."""
    g = golden("kron_B_112x128")
    for ci in range(3):
        s = g["sig"][ci]
        o = O.kron_quant(g["x"], g["L"], g["R"], s[0], s[1], clamp0=True, round_y_f16=True, left_first=True)
        assert np.array_equal(o["packed"], g[f"b_packed{ci}"])
        assert np.array_equal(o["scale16"], g[f"b_scale{ci}"])


@pytest.mark.parametrize("shape", ["64x64", "64x128", "32x64", "112x128"])
def test_pinned_association_vs_path_b_within_tolerance(golden, shape):
    """The shipped arithmetic (right factor first, as reference path A) vs path B's packed output."""
    g = golden(f"kron_B_{shape}")
    for ci in range(3):
        s = g["sig"][ci]
        o = O.kron_quant(g["x"], g["L"], g["R"], s[0], s[1], clamp0=False)
        qb = O.unpack_i4(g[f"b_packed{ci}"])
        assert mismatch(o["q"], qb) <= 1e-3
        assert np.max(np.abs(o["q"].astype(int) - qb)) <= 1
        sb = g[f"b_scale{ci}"].astype(np.float32)
        assert np.max(np.abs(o["scale16"].astype(np.float32) - sb) / sb) <= 1e-3


@pytest.mark.parametrize("shape", ["128x32", "128x64"])
def test_block_b_bit_exact(golden, shape):
    g = golden(f"block_B_{shape}")
    x = g["x"]
    T = x.shape[0] * x.shape[1]
    for ci in range(2):
        s = g[f"sig{ci}"]
        o = O.block_quant(x.reshape(T, x.shape[2], x.shape[3]), g["P"], s[0], s[1], transpose_out=True,
                          clamp0=False)
        assert np.array_equal(o["packed"], g[f"b_packed{ci}"])
        assert np.array_equal(o["scale16"], g[f"b_scale{ci}"])


@pytest.mark.parametrize("n", [4096, 8192, 14336, 28672, 11008, 1024, 512, 5120])
def test_hadamard_vs_matmul_hadU(golden, n):
    g = golden("had_A")
    K = int(g[f"K_{n}"])
    y = O.hadamard(g[f"x_{n}"], K, hadk_matrix(K) if K > 1 else None)
    y64 = g[f"y64_{n}"]
    den = np.abs(y64).max(axis=1, keepdims=True)
    assert np.max(np.abs(y.astype(np.float64) - y64) / den) <= 1e-3
    # the reference's own fp16 evaluation (one rounding per butterfly stage) is no closer to fp64 than we are
    assert np.max(np.abs(y.astype(np.float64) - y64)) <= np.max(np.abs(g[f"y16_{n}"].astype(np.float64) - y64)) * 1.5


def test_hadamard_is_orthogonal_and_involutive():
    rng = np.random.RandomState(0)
    for n, K in [(512, 1), (14336, 28), (5120, 40)]:
        x = rng.randn(2, n).astype(np.float16)
        hk = hadk_matrix(K) if K > 1 else None
        y = O.hadamard(x, K, hk)
        nx, ny = np.linalg.norm(x.astype(np.float64), axis=1), np.linalg.norm(y.astype(np.float64), axis=1)
        assert np.allclose(nx, ny, rtol=2e-3)


def test_edge_rows(golden):
    g = golden("edge_64x64")
    for tag in ("rand", "eye"):
        o = O.kron_quant(g["x"], g[tag + "_L"], g[tag + "_R"], g["sig"][0], g["sig"][1], round_y_f16=True)
        assert mismatch(o["q"], g[tag + "_q"]) <= 1e-3
        assert np.allclose(o["scale"], g[tag + "_scale"], rtol=1e-3)
        assert o["scale"][0] == 1.0 and not o["q"][0].any()            # all-zero token -> scale 1, q 0
    e = O.kron_quant(g["x"], g["eye_L"], g["eye_R"], g["sig"][0], g["sig"][1], round_y_f16=True)
    assert np.array_equal(e["q"], g["eye_q"].reshape(8, -1))             # identity factors: exact
    assert np.array_equal(e["q"][5][:8], np.array([7, 0, 2, 2, 0, -2, 4, -4], dtype=np.int8))  # ties -> even


def test_sym_quant_restatement_properties():
    rng = np.random.RandomState(1)
    x = (rng.randn(16, 64) * 3).astype(np.float16)
    s = (np.abs(x).max(axis=1) / 7).astype(np.float16)
    p = O.sym_quant(x, s)
    q = O.unpack_i4(p)
    assert q.min() >= -8 and q.max() <= 7
    assert np.max(np.abs(q * s[:, None].astype(np.float32) - x.astype(np.float32)) /
                  s[:, None].astype(np.float32)) <= 0.5 + 2e-3
    odd = O.sym_quant(x[:, :63], s)
    assert odd.shape == (16, 32) and np.all(odd[:, -1] >> 4 == 0)


def test_sym_dequant_restatement():
    q = np.array([[0, 10, -10, 12345, -99999, 70000 * 10]], dtype=np.int32)
    out = O.sym_dequant(q, np.array([0.5], np.float16), np.ones(6, np.float16))
    assert out[0, 0] == 0 and out[0, 1] == 5 and out[0, 2] == -5
    assert out[0, 3] == np.float16(np.float16(np.float16(0.5) * np.float16(1234)) * np.float16(10))


def test_path_a_torch_port_matches_reference_goldens(golden):
    """oracle/path_a_torch.py (the cpu_baseline 'port') reproduces the reference's path A bit for bit: same
    torch ops in the same order on the same CPU BLAS."""
    import torch
    from oracle import path_a_torch
    g = golden("kron_A_64x64")
    x, L, R = (torch.from_numpy(g[k]) for k in ("x", "L", "R"))
    for ci in range(2):
        fq = path_a_torch.kron_fakequant(x, L, R, (float(g["sig"][ci][0]), float(g["sig"][ci][1])))
        assert same_bits(fq.numpy(), g[f"a16_lac{ci}_fq"])
    y = path_a_torch.kronecker_matmul(x, L, R)
    assert np.array_equal(y.numpy(), g["a16_lac0_y"])


ASYM_CASES = [("lac32", False), ("lac32b", False), ("plain", True), ("ratio", True), ("lac16", True)]


@pytest.mark.parametrize("name,f16", ASYM_CASES)
@pytest.mark.parametrize("cols", [128, 64, 1000, 4096, 10240])
def test_asymmetric_quantizer_bit_exact_vs_reference(golden, name, f16, cols):
    """ActivationQuantizer(sym=False) of the reference (quant_utils.py:33-46,109-117), run by tools/gen_golden.py on
    fp16 activations in five configurations (fp32-promoted lac, plain, clip_ratio, half()'ed lac): every output bit."""
    g = golden("act_asym")
    x, y = g[f"{name}_{cols}_x"], g[f"{name}_{cols}_y"]
    smax, smin = (g[f"{name}_sig"] if f"{name}_sig" in g else ((0.83, 0.83) if name == "ratio" else (1.0, 1.0)))
    o = O.rowquant_asym(x, smax, smin, quant_f16=f16)
    assert np.array_equal(o.view(np.uint16), y.view(np.uint16))


@pytest.mark.parametrize("cols", [128, 4096])
def test_symmetric_quantizer_with_clip_ratio_vs_reference(golden, cols):
    """clip_ratio without lac: the fp16 extremum times the python float is an fp16 tensor (sig_f16), then / 7 in fp16."""
    g = golden("act_asym")
    x, y = g[f"symratio_{cols}_x"], g[f"symratio_{cols}_y"]
    o = O.rowquant(x, 0.83, 0.83, clamp0=True, quant_f16=True, sig_f16=True)["fq"]
    assert np.array_equal(o.view(np.uint16), y.view(np.uint16))
