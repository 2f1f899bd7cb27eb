"""A reference-written real-quant export (tests/golden/ckpt/) loaded into flatquant_amd's deploy modules and run
on the GPU, against the reference's own fake-quant evaluation of the same exported model (fp32 on CPU,
tests/golden/ckpt_io.npz). This composes the whole path — key map, Kronecker transform + INT4 quantiser, Linear4bit
GEMM + dequant epilogue, SiLU.mul — so the tolerance is the model-level one: the two evaluations differ in fp16 vs
fp32 activations and in the INT4 indices that sit on rounding ties, not in structure. A wrong key,
a transposed factor or a swapped clip factor produces errors of order 1, not 1e-2.

One reference quirk matters at this toy width: sym_dequant truncates the int32 accumulator to a multiple of 10 before
the fp16 conversion (quant.cu:83-84, an overflow guard). With K = 256 the accumulators are of order 100, so that
truncation alone is ~6 % of the output norm (it is ~0.1 % at K = 4096). The module path (which reproduces the quirk
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
