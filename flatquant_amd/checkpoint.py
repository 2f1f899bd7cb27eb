"""The two wire formats on either side of the hot path, host side only.

1. ``flat_matrices.pth`` — the calibrated transform matrices and clipping factors, ``{layer_index: {name: tensor}}``
   with names relative to the decoder layer (``self_attn.ln_trans.matrix_left`` ...). Reference:
   flatquant/flat_utils.py:65-93 (save_flat_matrices / load_flat_matrices), function_utils.py:51-58.
2. The real-quant export — ``model.safetensors`` (or shards + ``model.safetensors.index.json``) holding packed INT4
   weights (uint8, two nibbles per byte), fp16 everything else, ``quantizer.<layer>.scale`` tensors and a
   ``quantization_config`` metadata entry; ``quantization_config.json`` beside it. Reference:
   flatquant/flat_utils.py:97-204 (writer), deploy/transformers/modeling_llama.py:388-538 (reader + the key-rename
   table that maps the fake-quant model's names onto the deploy modules).

Nothing here touches a GPU except ``pack_quantized_weight`` when handed CUDA tensors; file IO uses torch.save/load and
safetensors exactly like the reference so that files written by either side load on the other.
"""
import json
import os
from collections import OrderedDict
from typing import Dict, Iterable, Mapping, Optional, Tuple

import torch

# ---------------------------------------------------------------------------------------------------------------
# 1. flat_matrices.pth
# ---------------------------------------------------------------------------------------------------------------
FLAT_PARAM_NAMES = ("trans.matrix", "trans.diag_scale", "clip_factor_w", "clip_factor_a")  # flat_utils.py:71


def get_paras_dict_by_name(module: torch.nn.Module, required_names: Iterable[str] = FLAT_PARAM_NAMES,
                           destination: Optional[OrderedDict] = None, prefix: str = "") -> OrderedDict:
    """Parameters whose name contains one of ``required_names``, grouped in that order (function_utils.py:51-58)."""
    if destination is None:
        destination = OrderedDict()
    for r_name in required_names:
        for name, param in module.named_parameters():
            if name.find(r_name) > -1:
                destination[prefix + name] = param.detach()
    return destination


def collect_flat_matrices(layers: Iterable[torch.nn.Module]) -> Dict[int, OrderedDict]:
    """{i: parameters of layer i that make up its FlatQuant state}. The reference first collapses the training
    parametrisation (``rep_matrix_only``); modules that have one are asked to, eval-mode mirrors have nothing to do."""
    out = {}
    for i, layer in enumerate(layers):
        for sub in ("self_attn", "mlp"):
            m = getattr(layer, sub, None)
            if m is not None and hasattr(m, "rep_matrix_only"):
                m.rep_matrix_only()
        out[i] = get_paras_dict_by_name(layer)
    return out


def save_flat_matrices(layers: Iterable[torch.nn.Module], exp_dir: str, rank: Optional[int] = None) -> str:
    """flat_utils.py:65-80: ``<exp_dir>/flat_matrices.pth`` (``flat_matrices_<rank>.pth`` when a rank is given)."""
    path = os.path.join(exp_dir, "flat_matrices.pth" if rank is None else f"flat_matrices_{rank}.pth")
    torch.save(collect_flat_matrices(layers), path)
    return path


def read_flat_matrices(path: str) -> Dict[int, Mapping[str, torch.Tensor]]:
    """``path`` is the file or the directory that holds ``flat_matrices.pth``. Tensors only (weights_only load)."""
    if os.path.isdir(path):
        path = os.path.join(path, "flat_matrices.pth")
    flat = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(flat, dict) or not all(isinstance(k, int) for k in flat):
        raise ValueError(f"{path}: expected {{layer_index: {{name: tensor}}}}")
    return flat


def load_flat_matrices(layers, path: str):
    """flat_utils.py:83-93: layer i takes entry i with ``strict=False`` (a layer keeps what the file does not name).
    Returns {i: (missing_keys, unexpected_keys)} so that a caller can see what did not land — the reference drops that."""
    flat = read_flat_matrices(path)
    report = {}
    for i in range(len(flat)):
        layer = layers[i]
        for sub in ("self_attn", "mlp"):
            m = getattr(layer, sub, None)
            if m is not None and hasattr(m, "rep_matrix_only"):
                m.rep_matrix_only()
        res = layer.load_state_dict(flat[i], strict=False)
        report[i] = (list(res.missing_keys), list(res.unexpected_keys))
    return report


# ---------------------------------------------------------------------------------------------------------------
# 2. real-quant safetensors
# ---------------------------------------------------------------------------------------------------------------
# modeling_llama.py:480-509. ORDER MATTERS: each pair is applied to the result of the previous one (str.replace
# chain), e.g. "ln_trans.matrix_left" must be consumed before the bare "ln_trans" rule.
DEPLOY_KEY_TABLE: Tuple[Tuple[str, str], ...] = (
    ("q_proj.linear", "q_proj"),
    ("q_proj.act_quantizer", "inp_trans_q"),
    ("k_proj.linear", "k_proj"),
    ("k_proj.act_quantizer", "inp_trans_k"),
    ("v_proj.linear", "v_proj"),
    ("v_proj.act_quantizer", "inp_trans_v"),
    ("o_proj.linear", "o_proj.1"),
    ("o_proj.act_quantizer", "o_proj_trans"),
    ("ln_trans.matrix_left", "left_matrix"),
    ("ln_trans.matrix_right", "right_matrix"),
    ("ln_trans", "inp_trans_k"),
    ("o_trans.matrix", "o_proj_trans.right_matrix"),
    ("gate_proj.linear", "gate_proj"),
    ("gate_proj.act_quantizer", "inp_trans_g"),
    ("up_proj.linear", "up_proj"),
    ("up_proj.act_quantizer", "inp_trans_u"),
    ("down_proj.linear", "down_proj.2"),
    ("down_proj.act_quantizer", "down_proj.0"),
    ("down_trans.matrix_left", "down_proj.0.left_matrix"),
    ("down_trans.matrix_right", "down_proj.0.right_matrix"),
    ("down_trans", "down_proj.0"),
    ("up_gate_trans.matrix_left", "left_matrix"),
    ("up_gate_trans.matrix_right", "right_matrix"),
    ("up_gate_trans", "inp_trans_g"),
    ("k_cache_quantizer.clip", "kclip"),
    ("v_cache_quantizer.clip", "vclip"),
    ("kcache_trans.matrix", "trans_matrix_k"),
    ("vcache_trans.matrix", "trans_matrix_v"),
)
# modeling_llama.py:511-516, applied to the layer names under "quantizer."
QUANTIZER_KEY_TABLE: Tuple[Tuple[str, str], ...] = (
    ("linear", "weight_scales"),
    ("mlp.down_proj.weight_scales", "mlp.down_proj.2.weight_scales"),
    ("self_attn.o_proj.weight_scales", "self_attn.o_proj.1.weight_scales"),
)


def _chain(k: str, table) -> str:
    for old, new in table:
        k = k.replace(old, new)
    return k


def deploy_key(k: str) -> str:
    """Name of export tensor ``k`` in the deploy model (modeling_llama.py:480-509)."""
    return _chain(k, DEPLOY_KEY_TABLE)


def deploy_scale_key(layer_name: str) -> str:
    """``quantizer.<layer_name>.scale`` -> the ``weight_scales`` buffer it fills (modeling_llama.py:511-516)."""
    return _chain(layer_name, QUANTIZER_KEY_TABLE)


def read_safetensors_dir(path: str) -> Tuple[Dict[str, torch.Tensor], Dict[str, str]]:
    """All tensors of a real-quant export directory (sharded through ``model.safetensors.index.json`` or a single
    ``model.safetensors``, modeling_llama.py:405-427) and the metadata of the file that carries it."""
    from safetensors import safe_open
    state, meta = {}, {}
    index_path = os.path.join(path, "model.safetensors.index.json")
    if os.path.exists(index_path):
        with open(index_path) as f:
            weight_map = json.load(f)["weight_map"]
        files = sorted(set(weight_map.values()))
    else:
        weight_map, files = None, ["model.safetensors"]
    for fn in files:
        with safe_open(os.path.join(path, fn), framework="pt") as f:
            meta.update(f.metadata() or {})
            for key in f.keys():
                if weight_map is None or weight_map.get(key) == fn:
                    state[key] = f.get_tensor(key)
    return state, meta


def quantization_config(meta: Mapping[str, str]) -> dict:
    """The JSON document under the ``quantization_config`` metadata key (flat_utils.py:154-159)."""
    return json.loads(meta["quantization_config"]) if "quantization_config" in meta else {}


def deploy_state_dicts(state: Mapping[str, torch.Tensor]) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """Split an export state dict the way the reference loader does (modeling_llama.py:455-516):
    (renamed module tensors, renamed weight scales). ``quantizer.*.zero`` / ``.maxq`` are dropped like there."""
    model_sd, scales = {}, {}
    for k, v in state.items():
        if k.startswith("quantizer."):
            parts = k.split(".")
            if parts[-1] == "scale":
                scales[deploy_scale_key(".".join(parts[1:-1]))] = v
        else:
            model_sd[deploy_key(k)] = v
    return model_sd, scales


def load_deploy_checkpoint(model: torch.nn.Module, path_or_state, share_matrices: bool = True):
    """Fill a deploy model (modules named like the reference's FlatQuantLlamaForCausalLM) from an export directory or
    an already read state dict: rename, ``load_state_dict(strict=False)`` twice (tensors, then scales), and hand each
    attention's / MLP's shared Kronecker factors to its per-projection OnlineTrans modules (modeling_llama.py:518-529).
    Clipping factors stay one-element buffers: flatquant_amd reads them to the host once (ops.host_scalar), which is
    what the reference's conversion to Python floats (:531-537) is for. Returns the two load reports."""
    state = read_safetensors_dir(path_or_state)[0] if isinstance(path_or_state, (str, os.PathLike)) else path_or_state
    model_sd, scales = deploy_state_dicts(state)
    r1 = model.load_state_dict(model_sd, strict=False)
    r2 = model.load_state_dict(scales, strict=False)
    if share_matrices:
        for m in model.modules():
            for holder, users in ((m, ("inp_trans_q", "inp_trans_k", "inp_trans_v")), (m, ("inp_trans_u", "inp_trans_g"))):
                if not all(hasattr(holder, u) for u in users):
                    continue
                if not (hasattr(holder, "left_matrix") and hasattr(holder, "right_matrix")):
                    continue
                for u in users:
                    sub = getattr(holder, u)
                    for name in ("left_matrix", "right_matrix"):
                        if name in sub._buffers:
                            del sub._buffers[name]
                        sub.register_buffer(name, getattr(holder, name))
    return r1, r2


def pack_quantized_weight(param: torch.Tensor, scale: torch.Tensor, maxq: torch.Tensor, zero: Optional[torch.Tensor] = None,
                          sym: bool = True) -> torch.Tensor:
    """flat_utils.py:118-131: integer weight = clamp(round(w / scale) [+ zero]), packed two per byte (low nibble =
    even column) -> uint8 [out, in/2]."""
    from .deploy.functional.quantization import pack_i4
    scale, maxq = scale.to(param.device), maxq.to(param.device)
    if sym:
        q = torch.clamp((param / scale).round(), -(maxq + 1), maxq)
    else:
        q = torch.clamp((param / scale).round() + zero.to(param.device), 0, maxq)
    return pack_i4(q.to(torch.int8)).contiguous()


def export_state_dict(named_parameters, quantizers: Mapping[str, object], sym: bool = True) -> Dict[str, torch.Tensor]:
    """The tensors save_quantized_weights_with_safetensors writes (flat_utils.py:106-141): packed uint8 for the weights
    of quantised layers, fp16 for every other parameter, then ``quantizer.<layer>.{scale,zero,maxq}``."""
    state = {}
    for name, param in named_parameters:
        layer_name = name.rsplit(".", 1)[0] if (name.endswith(".weight") or name.endswith(".bias")) else name
        if layer_name in quantizers and "weight" in name:
            qz = quantizers[layer_name]
            state[name] = pack_quantized_weight(param.detach(), qz.scale, qz.maxq, getattr(qz, "zero", None), sym)
        else:
            state[name] = param.detach().to(torch.half).contiguous()
    for layer_name, qz in quantizers.items():
        state[f"quantizer.{layer_name}.scale"] = qz.scale.contiguous()
        if getattr(qz, "zero", None) is not None:
            state[f"quantizer.{layer_name}.zero"] = qz.zero.contiguous()
        if getattr(qz, "maxq", None) is not None:
            state[f"quantizer.{layer_name}.maxq"] = qz.maxq.contiguous()
    return state


def save_quantized_weights_with_safetensors(exp_dir: str, named_parameters, quantizers, w_bits: int, model_name: str,
                                            sym: bool = True, max_shard_size: str = "5GB") -> None:
    """flat_utils.py:97-204: shards named ``model{suffix}.safetensors`` (metadata on the first), the index file when
    sharded, and ``quantization_config.json``."""
    from huggingface_hub import split_torch_state_dict_into_shards
    from safetensors.torch import save_file
    state = export_state_dict(named_parameters, quantizers, sym)
    split = split_torch_state_dict_into_shards(state, max_shard_size=max_shard_size,
                                               filename_pattern="model{suffix}.safetensors")
    os.makedirs(exp_dir, exist_ok=True)
    cfg = {"w_bits": w_bits, "model_name": model_name, "symmetric": sym, "format": "packed_int4"}
    first = True
    for filename, names in split.filename_to_tensors.items():
        shard = {n: state[n] for n in names}
        save_file(shard, os.path.join(exp_dir, filename), metadata={"quantization_config": json.dumps(cfg)} if first else None)
        first = False
    if split.is_sharded:
        with open(os.path.join(exp_dir, "model.safetensors.index.json"), "w") as f:
            json.dump({"metadata": getattr(split, "metadata", {}) or {}, "weight_map": split.tensor_to_filename}, f, indent=2)
    with open(os.path.join(exp_dir, "quantization_config.json"), "w") as f:
        json.dump(dict(cfg, sharded=split.is_sharded), f, indent=2)
