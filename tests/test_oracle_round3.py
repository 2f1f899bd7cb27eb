"""CPU: the round-3 oracle pieces against fixtures the REFERENCE wrote (tools/gen_golden.py r3): path A on bfloat16
activations (the dtype flatquant/model_utils.py:20 / main_dpskv3.py:395 feed it), the DeepSeek routed-expert flow under
torch.set_default_dtype(bfloat16), and o_proj head transforms for head counts other than 32 / 64."""
import numpy as np
import pytest

from oracle import fq_oracle as O

BF16_PAIRS = ["64x64", "64x112", "32x64", "112x128", "56x64", "128x148"]
MODES = {"lac32": dict(quant_f16=False, sig_f16=False), "lac16": dict(quant_f16=True, sig_f16=True),
         "nolac": dict(quant_f16=True, sig_f16=False)}


def bf(bits):
    return O.bf16_from_bits(bits)


def ulps_bf16(a, b):
    """|a - b| in bf16 steps (both bf16-representable fp32 arrays of equal sign pattern where it matters)."""
    ia = (a.view(np.uint32) >> 16).astype(np.int64)
    ib = (b.view(np.uint32) >> 16).astype(np.int64)
    ia = np.where(ia & 0x8000, 0x8000 - ia, ia)
    ib = np.where(ib & 0x8000, 0x8000 - ib, ib)
    return np.abs(ia - ib)


def close_bf16(y, want, frac=2e-2, tol=4e-3):
    """bf16 results of two summation orders: few elements differ, each by at most one bf16 step or tol of the largest value."""
    d = np.abs(y - want)
    return np.mean(y != want) < frac and bool(np.all((ulps_bf16(y, want) <= 1) | (d <= tol * np.abs(want).max())))


def test_bf16_helpers_round_to_nearest_even():
    x = np.array([1.0, 1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, 1.0 + 2.0 ** -9, -2.5, 3.0e38, 1e-40], dtype=np.float32)
    r = O.bf16_round(x)
    assert r[0] == 1.0 and r[1] == 1.0 and r[2] == 1.0 + 2.0 ** -6 and r[3] == 1.0   # ties to even, below half rounds down
    assert np.array_equal(O.bf16_from_bits(O.bf16_bits(r)), r)
    import torch
    t = torch.randn(4096, generator=torch.Generator().manual_seed(0)) * 37.0
    assert np.array_equal(O.bf16_round(t.numpy()), t.to(torch.bfloat16).float().numpy())


@pytest.mark.parametrize("tag", BF16_PAIRS)
def test_bf16_transform_and_quantiser_match_reference(golden, tag):
    g = golden("bf16_path_a")
    k = "k" + tag
    x, L, R = bf(g[k + "_x_bits"]), bf(g[k + "_L_bits"]), bf(g[k + "_R_bits"])
    rows = x.shape[0]
    y_ref = bf(g[k + "_lac32_y_bits"])
    # the transform: bf16 operands, fp32 accumulation, U and Y rounded to bf16 (flat_utils.py:15-16 on bf16 tensors). The
    # oracle sums each dot product exactly; torch's CPU GEMM has an order of its own: a few results land on the other side
    # of a bf16 rounding boundary (one step)
    y = O.bf16_round(O.kron_transform(x, L, R, lowp="bf16")).reshape(rows, -1)
    # (an element near zero next to large ones can move several of ITS steps when one bf16 rounding of U flips: bound the
    #  absolute difference by the row scale, as the fp16 tests do)
    assert close_bf16(y, y_ref), (ulps_bf16(y, y_ref).max(), np.mean(y != y_ref))
    # the quantiser stage on the reference's own transformed rows: bit for bit, every promotion route
    for mode, kw in MODES.items():
        smax, smin = (float(v) for v in g[f"{k}_{mode}_sig"]) if mode != "nolac" else (1.0, 1.0)
        yr = bf(g[f"{k}_{mode}_y_bits"])
        o = O.quant_outputs(yr, smax, smin, round_y_f16=True, lowp="bf16", **kw)
        assert np.array_equal(o["scale"], g[f"{k}_{mode}_scale"]), mode
        assert np.array_equal(o["q"], g[f"{k}_{mode}_q"]), mode
        assert np.array_equal(O.bf16_bits(o["fq"]), g[f"{k}_{mode}_fq_bits"]), mode


def test_bf16_activation_quantizer_matches_reference(golden):
    g = golden("bf16_path_a")
    routes = {"lac32": (False, False), "lac32b": (False, False), "plain": (True, False), "ratio": (True, True),
              "lac16": (True, True)}
    for name, (qf, sf) in routes.items():
        if name == "plain":
            sig = (1.0, 1.0)
        elif name == "ratio":
            sig = (0.83, 0.83)
        else:
            sig = tuple(float(v) for v in g[f"aq_{name}_sig"])
        for cols in (128, 4096, 7168):
            x = bf(g[f"aq_sym_{name}_{cols}_x_bits"])
            o = O.rowquant(x, *sig, clamp0=True, quant_f16=qf, sig_f16=sf, lowp="bf16")
            assert np.array_equal(O.bf16_bits(o["fq"]), g[f"aq_sym_{name}_{cols}_y_bits"]), (name, cols)
            xa = bf(g[f"aq_asym_{name}_{cols}_x_bits"])
            ya = O.rowquant_asym(xa, *sig, quant_f16=qf, lowp="bf16")
            assert np.array_equal(O.bf16_bits(ya), g[f"aq_asym_{name}_{cols}_y_bits"]), ("asym", name, cols)


def test_bf16_flat_quantized_linear_quantiser_matches_reference(golden):
    g = golden("bf16_path_a")
    o = O.rowquant(bf(g["fql_x_bits"]), float(g["fql_sig"][0]), float(g["fql_sig"][1]), lowp="bf16")
    assert np.array_equal(O.bf16_bits(o["fq"]), g["fql_fq_bits"])


def test_bf16_moe_flow_matches_reference(golden):
    g = golden("moe_bf16")
    offs = g["offsets"]
    E = len(offs) - 1
    s1, s2 = tuple(float(v) for v in g["sig1"]), tuple(float(v) for v in g["sig2"])
    # under set_default_dtype(bfloat16) the clip parameters are bf16: the sigmoids are bf16 values, everything stays bf16
    assert np.array_equal(O.bf16_round(np.array(s1 + s2, np.float32)), np.array(s1 + s2, np.float32))
    o1 = O.quant_outputs(bf(g["xt_bits"]), *s1, round_y_f16=True, quant_f16=True, sig_f16=True, lowp="bf16")
    assert np.array_equal(O.bf16_bits(o1["fq"][g["rows_tok"]]), g["fq1_bits"])
    y2 = bf(g["y2_bits"])
    parts = [O.quant_outputs(y2[offs[i]:offs[i + 1]], *s2, round_y_f16=True, quant_f16=True, sig_f16=True, lowp="bf16")["fq"]
             for i in range(E) if offs[i + 1] > offs[i]]
    assert np.array_equal(O.bf16_bits(np.concatenate(parts)), g["fq2_bits"])
    # end to end through the grouped entry: the oracle's own GEMM order moves a few bf16 roundings of the transform
    o = O.kron_quant_grouped(bf(g["h_bits"]), bf(g["L2_bits"]), bf(g["R2_bits"]), offs, [s2[0]] * E, [s2[1]] * E,
                             round_y_f16=True, quant_f16=True, sig_f16=True, lowp="bf16")
    assert np.mean(O.bf16_bits(o["fq"]) != g["fq2_bits"]) < 2e-2


HEADS = [(28, 128), (40, 128), (48, 64), (12, 128), (16, 128), (14, 64)]


@pytest.mark.parametrize("H,hd", HEADS)
def test_single_transform_any_head_count(golden, H, hd):
    g = golden("heads_any")
    tag = f"h{H}x{hd}"
    x = g[tag + "_x"]                                              # [T, hd, H] fp16
    for key, mat in (("_y16", "_matrix"), ("_y16_inv_t", "_matrix_inv_t")):
        P = g[tag + mat].astype(np.float16)
        y = O.single_transform(x, P).astype(np.float16)
        want = g[tag + key]
        d = np.abs(y.astype(np.float32) - want.astype(np.float32))
        assert np.mean(y != want) < 2e-2 and d.max() <= 2e-3 * np.abs(want.astype(np.float32)).max()
    # bf16
    xb = O.bf16_round(x.astype(np.float32))
    Pb = O.bf16_round(g[tag + "_matrix"])
    yb = O.bf16_round(O.single_transform(xb, Pb, lowp="bf16"))
    assert close_bf16(yb, bf(g[tag + "_ybf_bits"]))


def test_reference_path_b_takes_power_of_two_head_counts_only(golden):
    """The reference's Triton block_matmul (deploy/kernels/block_matmul.py:56-66: `tl.arange(0, np2_N)` with np2_N = N)
    cannot be traced for 28 / 40 / 48 / 12 / 14 heads — recorded by the generator. Where it runs (16 heads) the oracle's
    transposed pack reproduces its bytes; the other head counts are pinned on path A (above) and on this restatement."""
    g = golden("heads_any")
    ok = dict(zip(HEADS, g["pathb_ok"]))
    assert ok[(16, 128)] == 1 and ok[(28, 128)] == 0 and ok[(40, 128)] == 0
    tag = "h16x128"
    x = g[tag + "_x"]
    o = O.block_quant(x, g[tag + "_P"], float(g[tag + "_b_sig"][0]), float(g[tag + "_b_sig"][1]), transpose_out=True,
                      clamp0=False)
    want = g[tag + "_b_packed"]
    flips = np.mean(O.unpack_i4(o["packed"]) != O.unpack_i4(want))
    assert flips < 1e-3, flips
    assert np.max(np.abs(o["scale16"].astype(np.float32) - want_scale(g, tag))) <= 2e-3 * np.max(want_scale(g, tag))


def want_scale(g, tag):
    return g[tag + "_b_scale"].astype(np.float32).reshape(-1)
