"""A reference-written real-quant export (tests/golden/ckpt/) loaded into flatquant_amd's deploy modules and run
on the GPU, against the reference's own fake-quant evaluation of the same exported model (fp32 on CPU,
tests/golden/ckpt_io.npz). This composes the whole path — key map, Kronecker transform + INT4 quantiser, Linear4bit
GEMM + dequant epilogue, SiLU.mul — so the tolerance is the model-level one: the two evaluations differ in fp16 vs
fp32 activations and in the INT4 indices that sit on rounding ties, not in structure. A wrong key,
a transposed factor or a swapped clip factor produces errors of order 1, not 1e-2.

One reference quirk matters at this toy width: sym_dequant truncates the int32 accumulator to a multiple of 10 before
the fp16 conversion (quant.cu:83-84, an overflow guard). With K = 256 the accumulators are of order 100, so that
truncation alone is ~6 % of the output norm (~2-3 % at K = 2048: the accumulators only grow like sqrt(K)). The module path (which reproduces the quirk
bit for bit, tests/test_gpu_gemm_i4.py) is therefore held to 1e-1 here, and the same accumulators de-quantised
without the truncation to 2e-2.
"""
import os

import numpy as np
import pytest
import torch

from ckpt_models import deploy_model
from flatquant_amd import checkpoint as C

pytestmark = pytest.mark.gpu
CKPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt")


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.fixture(scope="module")
def loaded():
    torch.set_default_dtype(torch.float16)          # the reference loader builds the deploy model in fp16
    try:
        model = deploy_model()
    finally:
        torch.set_default_dtype(torch.float32)
    C.load_deploy_checkpoint(model, CKPT)
    return model.cuda()


def test_qkv_projections_match_the_reference_evaluation(loaded, golden):
    g = golden("ckpt_io")
    x = torch.from_numpy(g["x"]).cuda()
    attn = loaded.model.layers[0].self_attn
    from flatquant_amd import ops
    for n in ("q", "k", "v"):
        lin = getattr(attn, f"{n}_proj")
        p = getattr(attn, f"inp_trans_{n}")(x)
        y = lin(p).float().cpu().numpy()
        assert y.shape == g[n].shape
        assert rel_l2(y, g[n]) <= 1e-1, n
        acc = ops.int4_matmul(p.quantized_x.reshape(-1, p.quantized_x.shape[-1]), lin.weight).double()
        exact = acc * p.scales_x.reshape(-1, 1).double() * lin.weight_scales.reshape(1, -1).double()
        assert rel_l2(exact.cpu().numpy().reshape(g[n].shape), g[n]) <= 2e-2, n


def _exact_linear(ops, lin, p):
    """the Linear4bit's integer accumulators de-quantised in fp32, without sym_dequant's multiple-of-10 truncation"""
    acc = ops.int4_matmul(p.quantized_x.reshape(-1, p.quantized_x.shape[-1]), lin.weight).float()
    y = acc * p.scales_x.reshape(-1, 1).float() * lin.weight_scales.reshape(1, -1).float()
    return y.reshape(*p.quantized_x.shape[:-1], -1)


def test_mlp_block_matches_the_reference_evaluation(loaded, golden):
    from flatquant_amd import ops
    g = golden("ckpt_io")
    x = torch.from_numpy(g["x"]).cuda()
    mlp = loaded.model.layers[0].mlp
    up = mlp.up_proj(mlp.inp_trans_u(x))
    gate = mlp.gate_proj(mlp.inp_trans_g(x))
    y_ref_order = mlp.down_proj(ops.silu_mul(gate, up))                       # modeling_llama.py:277-279 as written
    y_fused = mlp.down_proj[2](mlp.down_proj[0](gate, up=up))                 # SiLU.mul inside the transform launch
    assert torch.equal(y_ref_order, y_fused)
    # module path: three truncating linears (see the header) -> loose; un-truncated accumulators -> tight
    e_mod = rel_l2(y_fused.float().cpu().numpy(), g["mlp_out"])
    up_e = _exact_linear(ops, mlp.up_proj, mlp.inp_trans_u(x)).half()
    gate_e = _exact_linear(ops, mlp.gate_proj, mlp.inp_trans_g(x)).half()
    y_e = _exact_linear(ops, mlp.down_proj[2], mlp.down_proj[0](gate_e, up=up_e))
    e_exact = rel_l2(y_e.cpu().numpy(), g["mlp_out"])
    print("mlp rel l2: module path", e_mod, "un-truncated", e_exact)
    # a ~1 % perturbation of the down_proj input (fp16 factors, fp16 scales) moves ~3 % of its INT4 indices by one
    # step, i.e. ~6 % of the output norm at 4 bits: measured 5.7e-2 un-truncated, 2.1e-1 through the module path
    assert e_exact <= 8e-2
    assert e_mod <= 2.5e-1


# ---- the same composition at K = 2048 (tests/golden/ckpt2k/, `python tools/gen_golden.py ckpt2k`): hidden = ffn = 2048, 16 heads,
# one layer. The SHARP checks are on the quantised activations themselves, in front of every linear: per-token scales to 2e-3
# and INT4 digits equal to the reference's except for the ties the fp16 transform moves (measured: 0.06 % of them, by one) —
# with clip factors that differ by several per cent between the quantisers, so that an exchanged pair fails (shown). The
# linears on top: un-truncated accumulators within 2e-2 of the reference's fp32 evaluation; the module path (which reproduces
# sym_dequant's multiple-of-10 truncation, ~2-3 % of the output norm at this K) within 5e-2.
CKPT2K = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt2k")


@pytest.fixture(scope="module")
def loaded2k():
    torch.set_default_dtype(torch.float16)
    try:
        model = deploy_model(n_layers=1, hidden=2048, ffn=2048, heads=16, kv_heads=2)
    finally:
        torch.set_default_dtype(torch.float32)
    C.load_deploy_checkpoint(model, CKPT2K)
    return model.cuda()


def _deq(p):
    """PackedQuantizedTensor -> (integer digits [rows, K], scales [rows]) on the host"""
    from oracle import fq_oracle as O
    q = O.unpack_i4(p.quantized_x.reshape(-1, p.quantized_x.shape[-1]).cpu().numpy()).astype(np.float64)
    return q, p.scales_x.reshape(-1).float().cpu().numpy().astype(np.float64)


def _check_quantised_activation(p, ref_fq, what):
    """ours (fp16 transform, real INT4) against the reference's fake-quantised activation (fp32 evaluation) in front of the
    same linear: per-token scales within 2e-3 — the clip factors of this fixture differ by several per cent between the
    quantisers, so an exchanged pair fails here — and digits equal except for ties moved by the fp16 transform (<= 3 %, by one)."""
    q, s = _deq(p)
    ref = np.asarray(ref_fq, dtype=np.float64).reshape(q.shape)
    s_ref = np.abs(ref).max(axis=1) / np.abs(np.round(ref / s[:, None])).max(axis=1)   # the reference's step of each token
    assert np.max(np.abs(s - s_ref) / s_ref) <= 2e-3, what
    q_ref = np.round(ref / s_ref[:, None])
    diff = np.abs(q - q_ref)
    print(what, "digits moved", float(np.mean(diff != 0)), "max", diff.max())
    assert diff.max() <= 1 and np.mean(diff != 0) <= 3e-2, what


def test_k2048_quantised_activations_match_the_reference(loaded2k, golden):
    from flatquant_amd import ops
    g = golden("ckpt2k_io")
    x = torch.from_numpy(g["x"]).cuda()
    layer = loaded2k.model.layers[0]
    attn, mlp = layer.self_attn, layer.mlp
    for n in ("q", "k", "v"):
        _check_quantised_activation(getattr(attn, f"inp_trans_{n}")(x), g[f"aq_{n}"], n)
    pu, pg = mlp.inp_trans_u(x), mlp.inp_trans_g(x)
    _check_quantised_activation(pu, g["aq_up"], "up")
    _check_quantised_activation(pg, g["aq_gate"], "gate")
    # the clip pairs really differ in this fixture: up's digits against gate's reference must NOT pass
    with pytest.raises(AssertionError):
        _check_quantised_activation(pu, g["aq_gate"], "up vs gate (must differ)")
    # the down_proj stage on the REFERENCE's own x_up * silu(x_gate) (fixture `act`, rounded to fp16): transform + Quantizer
    act = torch.from_numpy(g["act"]).half().cuda()
    pd = mlp.down_proj[1](mlp.down_proj[0](act))
    _check_quantised_activation(pd, g["aq_down"], "down")
    y = mlp.down_proj[2](pd).float().cpu().numpy()
    e = rel_l2(y, g["mlp_out"])
    print("down_proj stage on the reference's activation: rel l2", e)
    assert e <= 5e-2


def test_k2048_qkv_module_path(loaded2k, golden):
    from flatquant_amd import ops
    g = golden("ckpt2k_io")
    x = torch.from_numpy(g["x"]).cuda()
    attn = loaded2k.model.layers[0].self_attn
    for n in ("q", "k", "v"):
        lin, p = getattr(attn, f"{n}_proj"), getattr(attn, f"inp_trans_{n}")(x)
        e = rel_l2(lin(p).float().cpu().numpy(), g[n])
        e_exact = rel_l2(_exact_linear(ops, lin, p).cpu().numpy(), g[n])
        print(n, "rel l2: module path", e, "un-truncated accumulators", e_exact)
        # the module path reproduces sym_dequant's truncation of the accumulator to a multiple of 10 (quant.cu:83-84): with
        # accumulators of rms ~270 at K = 2048 that alone is ~2-3 % of the output norm; without it the two evaluations differ
        # by the 0.06 % of activation digits the fp16 transform moves
        assert e_exact <= 2e-2, n
        assert e <= 5e-2, n


def test_k2048_mlp_module_path(loaded2k, golden):
    from flatquant_amd import ops
    g = golden("ckpt2k_io")
    x = torch.from_numpy(g["x"]).cuda()
    mlp = loaded2k.model.layers[0].mlp
    up = mlp.up_proj(mlp.inp_trans_u(x))
    gate = mlp.gate_proj(mlp.inp_trans_g(x))
    y = mlp.down_proj[2](mlp.down_proj[0](gate, up=up))
    assert torch.equal(y, mlp.down_proj(ops.silu_mul(gate, up)))
    e = rel_l2(y.float().cpu().numpy(), g["mlp_out"])
    print("mlp rel l2 (module path, K = 2048)", e)
    # two chained INT4 stages: the ~1 % perturbation of the down_proj input (fp16 vs fp32 activations) moves a few per cent of
    # its INT4 indices by one step; a swapped clip factor or matrix gives errors of order 1
    assert e <= 2e-1     # measured 1.3e-1 (the error of the first stage moves ~10 % of the second stage's digits); each stage alone: 5e-2 above
