import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from oracle import fq_oracle as O
from flatquant_amd import checkpoint as C, ops
from flatquant_amd._lib import FQ_OUT_PACKED, FQ_OUT_TRANSFORM, FQ_NO_CLAMP0
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
state, _ = C.read_safetensors_dir(os.path.join(root, "tests/golden/ckpt"))
g = np.load(os.path.join(root, "tests/golden/ckpt_io.npz"))
x = g["x"].reshape(-1, 256)
L = state["model.layers.0.self_attn.ln_trans.matrix_left"].numpy(); R = state["model.layers.0.self_attn.ln_trans.matrix_right"].numpy()
cm = float(state["model.layers.0.self_attn.q_proj.act_quantizer.clip_factor_a_max"]); cn = float(state["model.layers.0.self_attn.q_proj.act_quantizer.clip_factor_a_min"])
sig = (float(1 / (1 + np.exp(-cm))), float(1 / (1 + np.exp(-cn))))
o = ops.kron_quant(torch.from_numpy(x).cuda(), torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda(), [sig], FQ_OUT_PACKED | FQ_OUT_TRANSFORM | FQ_NO_CLAMP0)
y = o.y.cpu().numpy(); yo = O.kron_transform(x, L, R).reshape(y.shape)
print("transform max err", np.abs(y - yo).max(), np.abs(yo).max())
ref = O.kron_quant(x, L, R, sig[0], sig[1], clamp0=False)
q = O.unpack_i4(o.q[0].cpu().numpy())
print("q mismatch", np.mean(q != ref["q"]), "scale", np.abs(o.scale[0].cpu().numpy().astype(np.float32) - ref["scale"]).max())
wq = state["model.layers.0.self_attn.q_proj.linear.weight"]
ws = state["quantizer.model.layers.0.self_attn.q_proj.linear.scale"]
yl = ops.int4_linear(o.q[0].reshape(-1, 128).contiguous(), o.scale[0].reshape(-1).contiguous(), wq.cuda(), ws.reshape(-1).half().cuda(), None)
rl = O.linear4bit(o.q[0].cpu().numpy().reshape(-1, 128), o.scale[0].cpu().numpy().reshape(-1), wq.numpy(), ws.numpy().reshape(-1).astype(np.float16), None)
print("linear mismatch", np.mean(yl.cpu().numpy() != rl), np.abs(yl.cpu().numpy().astype(np.float32) - rl.astype(np.float32)).max())
refq = g["q"].reshape(-1, 256)
print("rel l2 vs reference", np.linalg.norm(yl.cpu().numpy().astype(np.float64) - refq) / np.linalg.norm(refq))
