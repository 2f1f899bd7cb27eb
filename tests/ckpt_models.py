"""Containers with the reference's module names, built from flatquant_amd modules, for the checkpoint tests.

Fake-quant side (flatquant/model_tools/llama_utils.py:20-45,112-160): ``self_attn.{q,k,v,o}_proj`` FlatQuantizedLinear,
``ln_trans / o_trans / kcache_trans / vcache_trans``, ``k/v_cache_quantizer``; ``mlp.{up,gate,down}_proj``,
``up_gate_trans / down_trans``. Deploy side (deploy/transformers/modeling_llama.py:156-197,238-266).
"""
import types

import torch
from torch import nn


def fake_quant_layer(hidden=256, ffn=512, heads=4, kv_heads=2):
    from flatquant_amd.flatquant.flat_linear import FlatQuantizedLinear
    from flatquant_amd.flatquant.function_utils import get_decompose_dim
    from flatquant_amd.flatquant.quant_utils import ActivationQuantizer
    from flatquant_amd.flatquant.trans_utils import SVDDecomposeTransMatrix, SVDSingleTransMatrix
    args = types.SimpleNamespace(w_bits=4, a_bits=4, lac=True, lwc=True, a_groupsize=-1, a_asym=False, w_asym=False)
    hd = hidden // heads
    attn = nn.Module()
    attn.q_proj = FlatQuantizedLinear(args, nn.Linear(hidden, hidden, bias=False))
    attn.k_proj = FlatQuantizedLinear(args, nn.Linear(hidden, kv_heads * hd, bias=False))
    attn.v_proj = FlatQuantizedLinear(args, nn.Linear(hidden, kv_heads * hd, bias=False))
    attn.o_proj = FlatQuantizedLinear(args, nn.Linear(hidden, hidden, bias=False))
    attn.ln_trans = SVDDecomposeTransMatrix(*get_decompose_dim(hidden), add_diag=True)
    attn.o_trans = SVDSingleTransMatrix(heads)
    attn.kcache_trans = SVDSingleTransMatrix(hd)
    attn.vcache_trans = SVDSingleTransMatrix(hd)
    attn.k_cache_quantizer = ActivationQuantizer(bits=4, sym=False, lac=True, groupsize=-1)
    attn.v_cache_quantizer = ActivationQuantizer(bits=4, sym=False, lac=True, groupsize=-1)
    mlp = nn.Module()
    mlp.gate_proj = FlatQuantizedLinear(args, nn.Linear(hidden, ffn, bias=False))   # LlamaMLP's order
    mlp.up_proj = FlatQuantizedLinear(args, nn.Linear(hidden, ffn, bias=False))
    mlp.down_proj = FlatQuantizedLinear(args, nn.Linear(ffn, hidden, bias=False))
    mlp.up_gate_trans = SVDDecomposeTransMatrix(*get_decompose_dim(hidden), add_diag=True)
    mlp.down_trans = SVDDecomposeTransMatrix(*get_decompose_dim(ffn), add_diag=True)
    layer = nn.Module()
    layer.self_attn, layer.mlp = attn, mlp
    return layer


def deploy_layer(hidden=256, ffn=512, heads=4, kv_heads=2):
    import flatquant_amd.deploy as deploy
    from flatquant_amd.flatquant.function_utils import get_decompose_dim
    hd = hidden // heads
    attn = nn.Module()
    attn.q_proj = deploy.nn.Linear4bit(hidden, hidden)
    attn.k_proj = deploy.nn.Linear4bit(hidden, kv_heads * hd)
    attn.v_proj = deploy.nn.Linear4bit(hidden, kv_heads * hd)
    attn.o_proj_trans = deploy.nn.OnlineTrans(heads, trans="matmul", decompose=False)
    attn.o_proj = nn.Sequential(deploy.nn.Quantizer(lac=True), deploy.nn.Linear4bit(hidden, hidden))
    for n in ("q", "k", "v"):
        setattr(attn, f"inp_trans_{n}", deploy.nn.OnlineTrans(hidden, trans="matmul"))
        setattr(attn, f"quantizer_{n}", deploy.nn.Quantizer(lac=True))
    attn.register_buffer("trans_matrix_k", torch.zeros(hd, hd))
    attn.register_buffer("trans_matrix_k_inv_t", torch.zeros(hd, hd))
    attn.register_buffer("trans_matrix_v", torch.zeros(hd, hd))
    for n in ("kclip_factor_a_max", "kclip_factor_a_min", "vclip_factor_a_max", "vclip_factor_a_min"):
        attn.register_buffer(n, torch.tensor(4.0))
    left, right = get_decompose_dim(hidden)
    attn.register_buffer("left_matrix", torch.zeros(left, left))
    attn.register_buffer("right_matrix", torch.zeros(right, right))
    mlp = nn.Module()
    mlp.up_proj = deploy.nn.Linear4bit(hidden, ffn)
    mlp.gate_proj = deploy.nn.Linear4bit(hidden, ffn)
    mlp.down_proj = nn.Sequential(deploy.nn.OnlineTrans(ffn, trans="matmul"), deploy.nn.Quantizer(lac=True),
                                  deploy.nn.Linear4bit(ffn, hidden))
    mlp.inp_trans_u = deploy.nn.OnlineTrans(hidden, trans="matmul")
    mlp.inp_trans_g = deploy.nn.OnlineTrans(hidden, trans="matmul")
    mlp.register_buffer("left_matrix", torch.zeros(left, left))
    mlp.register_buffer("right_matrix", torch.zeros(right, right))
    layer = nn.Module()
    layer.self_attn, layer.mlp = attn, mlp
    return layer


def deploy_model(n_layers=2, **kw):
    model = nn.Module()
    model.model = nn.Module()
    model.model.layers = nn.ModuleList([deploy_layer(**kw) for _ in range(n_layers)])
    return model
