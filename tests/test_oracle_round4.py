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
