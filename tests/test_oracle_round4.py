"""CPU tests, round 4: the oracle against the reference-written fixtures added this round (tools/gen_golden.py r4)."""
import numpy as np

from oracle import fq_oracle as O


def bf(bits):
    return (np.asarray(bits, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def test_single_trans_128_vs_reference(golden):
    """{SVD,Inv}SingleTransMatrix(128).forward (trans_utils.py:21-25, 136-151) over [tokens, heads, 128], inv_t too, fp16 and
    bf16 — restated as O.single_transform — and the per-head asymmetric quantiser of the K cache on the transformed keys."""
    g = golden("single128")
    for tag in ("svd", "inv"):
        x = g[tag + "_x"]                                   # [T, H, 128] fp16
        for key, mat in (("_y16", "_matrix"), ("_y16_inv_t", "_matrix_inv_t")):
            P = g[tag + mat].astype(np.float16)
            y = O.single_transform(x, P).astype(np.float16)
            want = g[tag + key]
            d = np.abs(y.astype(np.float32) - want.astype(np.float32))
            assert np.mean(y != want) < 2e-2 and d.max() <= 2e-3 * np.abs(want.astype(np.float32)).max(), (tag, key)
        xb = O.bf16_round(x.astype(np.float32))
        for key, mat in (("_ybf_bits", "_matrix"), ("_ybf_inv_t_bits", "_matrix_inv_t")):
            Pb = O.bf16_round(g[tag + mat])
            yb = O.bf16_round(O.single_transform(xb, Pb, lowp="bf16"))
            want = bf(g[tag + key])
            ulp = np.maximum(np.abs(want), 1e-30) * 2.0 ** -7
            assert np.all((np.abs(yb - want) <= ulp) | (np.abs(yb - want) <= 4e-3 * np.abs(want).max())), (tag, key)
        # ActivationQuantizer(bits=4, sym=False, lac=True) on the REFERENCE's transformed keys: bit for bit
        y16 = g[tag + "_y16"]
        kq = O.rowquant_asym(y16.reshape(-1, 128), float(g["kq_sig"][0]), float(g["kq_sig"][1]))
        assert np.array_equal(kq.reshape(y16.shape).view(np.uint16), g[tag + "_kq16"].view(np.uint16)), tag


def _bits_cases(g):
    """(key, bits, sym, route, dtag, cols) of every case of tests/golden/act_bits.npz."""
    for nb in (8, 6, 3):
        for sym in (True, False):
            for name in ("lac32", "lac32b", "plain", "ratio", "lac16"):
                for dtag in ("f16", "bf16"):
                    for cols in (128, 520):
                        yield f"b{nb}_{'sym' if sym else 'asym'}_{name}_{dtag}_{cols}", nb, sym, name, dtag, cols


def bits_case_args(g, name, dtag):
    """-> (sig_max, sig_min, quant_f16, sig_f16): how the reference module of that case evaluates (see rowquant / rowquant_asym)."""
    if name in ("lac32", "lac32b"):
        s = g[name + "_sig"]
        return float(s[0]), float(s[1]), False, False           # fp32 (1,)-shaped parameters promote everything to fp32
    if name == "lac16":
        s = g[f"lac16_{dtag}_sig"]
        return float(s[0]), float(s[1]), True, True             # a module in the activation dtype: every operation rounds to it
    if name == "ratio":
        return 0.83, 0.83, True, True                           # 16-bit extremum x python float: a 16-bit product
    return 1.0, 1.0, True, False


def test_activation_quantizer_other_bit_widths_vs_reference(golden):
    """ActivationQuantizer(bits = 8 / 6 / 3) — get_qmin_qmax (quant_utils.py:10-16), sym and asym, every clip route, fp16 and bf16 —
    restated by O.rowquant(bits=) / O.rowquant_asym(bits=): bit for bit against the reference's outputs."""
    g = golden("act_bits")
    n = 0
    for key, nb, sym, name, dtag, cols in _bits_cases(g):
        lowp = "f16" if dtag == "f16" else "bf16"
        x = g[key + "_x"] if dtag == "f16" else bf(g[key + "_x"])
        smax, smin, qf16, sf16 = bits_case_args(g, name, dtag)
        if sym:
            got = O.rowquant(x, smax, smin, clamp0=True, quant_f16=qf16, sig_f16=sf16, lowp=lowp, bits=nb)["fq"]
        else:
            got = O.rowquant_asym(x, smax, smin, quant_f16=qf16, lowp=lowp, bits=nb)
        if dtag == "f16":
            assert np.array_equal(np.asarray(got, dtype=np.float16).view(np.uint16), g[key + "_y"].view(np.uint16)), key
        else:
            assert np.array_equal(O.bf16_bits(got), g[key + "_y"]), key
        n += 1
    assert n == 120


def test_plain_quantizer_restatement_matches_torch_cpu():
    """oracle.quantizer_plain (deploy/nn/quantization.py:30: Quantizer(input_clip_ratio, lac=False)) against the same expression evaluated by
    torch on the CPU, and its digits against the quant.cu restatement the reference-pinned sym_quant tests use; an all-zero row keeps
    scale 0 and packs zeros (the reference has no guard on this branch)."""
    import torch
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(64, 512, generator=g) * 3).half()
    x[7] = 0
    x[9, 3] = 60000.0
    for ratio in (1.0, 0.9, 0.83):
        packed, s = O.quantizer_plain(x.numpy(), ratio)
        want = (torch.max(torch.abs(x), dim=-1)[0].unsqueeze(1) / 7).to(torch.float16) * ratio
        assert np.array_equal(s, want.reshape(-1).numpy())
        assert s[7] == 0 and not packed[7].any()
        live = np.arange(64) != 7
        assert np.array_equal(packed[live], O.sym_quant(x.numpy()[live], s[live]))

