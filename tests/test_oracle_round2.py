"""CPU: the round-2 oracle pieces against fixtures the REFERENCE wrote (tools/gen_golden.py r2):
per-128-element scales (vLLM ActivationQuantizer(groupsize=128)), the routed-expert flow of
flatquant/model_tools/deepseekv3_utils.py:427-452, deploy.nn.Quantizer(lac=True)."""
import numpy as np
import pytest

from oracle import fq_oracle as O


@pytest.mark.parametrize("tag", ["32x64", "64x64", "64x112", "56x64"])
def test_group128_quantiser_matches_reference(golden, tag):
    g = golden("group128")
    y16 = g[f"{tag}_y"]
    for ci in range(2):
        smax, smin = (float(v) for v in g[f"{tag}_sig{ci}"])
        o = O.quant_outputs(y16.astype(np.float32), smax, smin, groupsize=128)
        assert np.array_equal(o["scale"], g[f"{tag}_scale{ci}"])          # fp32 scales [rows, d/128], bit for bit
        assert np.array_equal(o["fq"], g[f"{tag}_fq{ci}"])
    # the transform in front of it: path A rounds Y to fp16 (flat_utils.py:15-16); association right-first
    M, N = (int(v) for v in tag.split("x"))
    y = O.kron_transform(g[f"{tag}_x"], g[f"{tag}_L"], g[f"{tag}_R"]).reshape(y16.shape).astype(np.float16)
    assert np.mean(y != y16) < 2e-2 and np.max(np.abs(y.astype(np.float32) - y16.astype(np.float32))) <= \
        2e-3 * np.max(np.abs(y16.astype(np.float32)))


def test_moe_grouped_flow_matches_reference(golden):
    g = golden("moe_grouped")
    offs = g["offsets"]
    E = len(offs) - 1
    assert offs[-1] == g["h"].shape[0] == g["rows_tok"].shape[0]
    # stage 1: w1_trans once, shared routed quantiser per token == quantise all tokens, then gather (x[idx])
    s1 = (float(g["sig1"][0]), float(g["sig1"][1]))
    o1 = O.quant_outputs(g["xt"].astype(np.float32), *s1)
    assert np.array_equal(o1["fq"][g["rows_tok"]], g["fq1"])
    # stage 2, shared routed_w2_trans: quantiser stage bit-exact on the reference's own transformed rows
    s2 = np.tile(g["sig2"][None, :], (E, 1))
    parts = [O.quant_outputs(g["y2_shared"][offs[i]:offs[i + 1]].astype(np.float32), float(s2[i, 0]), float(s2[i, 1]))["fq"]
             for i in range(E) if offs[i + 1] > offs[i]]
    assert np.array_equal(np.concatenate(parts), g["fq2_shared"])
    # end to end through the oracle's grouped entry (its own GEMM order): INT4 steps differ in < 1e-3 of the elements
    for L, R, sg, want in ((g["L2"], g["R2"], s2, g["fq2_shared"]), (g["L2e"], g["R2e"], g["sig2e"], g["fq2_indep"])):
        o = O.kron_quant_grouped(g["h"], L, R, offs, sg[:, 0], sg[:, 1], round_y_f16=True)
        assert o["fq"].shape == want.shape
        assert np.mean(o["fq"] != want) < 2e-3


def test_grouped_oracle_edge_cases():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((5, 2048)).astype(np.float16)
    L = (rng.standard_normal((32, 32)) / 6).astype(np.float16)
    R = (rng.standard_normal((64, 64)) / 8).astype(np.float16)
    one = O.kron_quant(x, L, R, 0.9, 0.8)
    # empty groups in front, in the middle and at the end; a 1-row group
    o = O.kron_quant_grouped(x, L, R, [0, 0, 1, 1, 5, 5], [0.1, 0.9, 0.2, 0.9, 0.3], [0.1, 0.8, 0.2, 0.8, 0.3])
    for k in ("packed", "scale16", "fq"):
        assert np.array_equal(o[k], one[k])
    e = O.kron_quant_grouped(x[:0], L, R, [0, 0], [1.0], [1.0])
    assert e["packed"].shape == (0, 1024)


def test_deploy_quantizer_lac_matches_reference(golden):
    g = golden("quantizer_lac")
    for ci in range(3):
        smax, smin = (float(v) for v in g[f"sig{ci}"])
        o = O.rowquant(g[f"x{ci}"], smax, smin, clamp0=True, quant_f16=True, sig_f16=True)
        assert np.array_equal(o["scale16"], g[f"scales{ci}"])             # the reference module's own scales
        assert np.array_equal(o["packed"], g[f"packed{ci}"])
        # the fp32-sigmoid form (quant_utils.py's (1,)-shaped parameters) is a different function: it must NOT be used here
        o32 = O.rowquant(g[f"x{ci}"], smax, smin, clamp0=True, quant_f16=True, sig_f16=False)
        assert not np.array_equal(o32["scale16"], g[f"scales{ci}"])
