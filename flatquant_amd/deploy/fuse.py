"""``flatquant_amd.deploy.fuse(model)`` — the fused launches for a model built by the reference's UNCHANGED deploy code.

The reference's decoder layer (deploy/transformers/modeling_llama.py) calls one module per projection:

    attention.forward :66-78     inp_trans_q(h), inp_trans_k(h), inp_trans_v(h)      three transforms of the SAME hidden state
                                 quantizer_q/k/v(...)                                  (pass packed inputs through, quantization.py:14)
                                 q_proj(hq), k_proj(hk), v_proj(hv)                    three GEMMs
    mlp.forward :268-280         up_proj(inp_trans_u(x)), gate_proj(inp_trans_g(x))    two transforms of the same x, two GEMMs
                                 down_proj = Sequential(OnlineTrans, Quantizer, Linear4bit)(x_up * act(x_gate))

and its loader hands the three (two) transforms of a group the SAME matrices (:518-529), so they differ only in their clip factors.
The library has the one-launch forms (deploy.nn.fused_forward: one transform, one quantisation per clip pair; linear4bit_multi: the
projections of a group as one GEMM launch; FusedSequential: rotation + Quantizer as one launch) — but until round 5 a maintainer had
to edit the two ``forward`` functions to reach them. ``fuse`` reaches them from the outside: it walks the model and

  * joins the transforms of a group in a ``TransformGroup``: the FIRST member called with a tensor runs the fused launch for all
    members, the others — called with that same tensor object, unmodified — take their result from it;
  * joins the projections behind them in a ``LinearGroup``: the first projection called with a group's packed tensor runs the
    multi-problem GEMM for all members (their inputs are the transform group's results), the others take theirs from it;
  * replaces ``down_proj`` by ``FusedSequential`` with the same children (same state-dict keys).

Results are bit for bit those of the unfused modules (each fused form is; tests/test_gpu_fuse.py runs a layer both ways). Nothing is
assumed about the ``forward`` that calls the modules except what the hooks check at run time: a member called with a different
tensor, a modified tensor, keyword arguments, or a second time simply runs on its own. Group results are dropped as soon as every
member has taken its own (nothing activation-sized is kept between layers).
"""
import weakref

import torch

from .. import ops
from .graphed import GraphedDecode
from .nn.linear import Linear4bit, _fused_decode_problems, fused_transform_linear, linear4bit_multi
from .nn.online_trans import FusedSequential, OnlineTrans, fused_forward
from .nn.quantization import Quantizer


DECODE_ROWS = 128   # at or below this many tokens an EAGER module call is Python-bound (tools/ref_layer.py --seq 1): the groups step aside
                    # — except while a HIP graph is being captured, where fewer launches are what counts


class TransformGroup:
    """The OnlineTrans modules of one attention (q, k, v) or MLP (up, gate): one launch per distinct input tensor."""

    def __init__(self, members):
        self.members = list(members)
        self.index = {id(m): i for i, m in enumerate(self.members)}
        self.linear_group = None
        self._real = None
        self._ref = None          # weakref of the tensor the cached results belong to
        self._xver = -1
        self._outs = None
        self._taken = 0           # bit mask of the members that have taken their result
        self._lazy_x = None       # the input of lazy results (the transform has not run: the projections' launch may carry it as its prologue)
        self.launches = 0         # (counters for tests / reports)
        self.served = 0
        self.lazy = 0

    def shareable(self):
        f = self.members[0]
        for t in self.members:
            if not (t.trans == "matmul" and t.decompose and "left_matrix" in t._buffers and "right_matrix" in t._buffers):
                return False
            if t._buffers["left_matrix"].data_ptr() != f._buffers["left_matrix"].data_ptr():
                return False
            if t._buffers["right_matrix"].data_ptr() != f._buffers["right_matrix"].data_ptr():
                return False
        return True

    def drop(self):
        self._ref, self._outs, self._taken, self._lazy_x = None, None, 0, None

    def materialise(self):
        """The real results behind lazy ones (somebody other than the group's projections asked for the packed bytes)."""
        x = self._lazy_x
        if x is None:
            raise RuntimeError("a lazy PackedQuantizedTensor outlived its group's call")
        if self._real is None:
            self._real = fused_forward(x, self.members)
            self.launches += 1
        return self._real

    def get(self, member, x):
        if x.dim() != 3:
            return None
        if x.shape[0] * x.shape[1] <= DECODE_ROWS and not torch.cuda.is_current_stream_capturing():
            return None                                      # decode-sized and eager: the members' own prepared calls are the faster route
        # (decode-sized under stream CAPTURE: the host cost is paid once, the replayed graph keeps one transform launch and one
        #  weight-streaming launch per group instead of three — a small launch costs ~4 us on the device whatever it does)
        i = self.index[id(member)]
        bit = 1 << i
        if self._outs is not None and self._ref is not None and self._ref() is x and self._xver == ops.ver(x) and not (self._taken & bit):
            out = self._outs[i]
            self._taken |= bit
            self.served += 1
        else:
            if not (x.dim() == 3 and x.is_cuda and self.shareable()):
                return None                                  # (the caller runs on its own)
            lg = self.linear_group
            if lg is not None and _fused_decode_problems(x, self.members, lg.members) is not None:
                # (round 6) decode-sized, and the projections behind this group can run the transform as their launch's prologue: hand
                # out lazy results — the bytes exist only if somebody else asks for them
                from . import LazyPackedQuantizedTensor
                shape = tuple(x.shape[:-1]) + (x.shape[-1] // 2,)
                outs = [LazyPackedQuantizedTensor(self, k, shape) for k in range(len(self.members))]
                self._lazy_x, self._real = x, None
                self.lazy += 1
            else:
                outs = fused_forward(x, self.members)
                self.launches += 1
            self._ref, self._xver, self._outs, self._taken = weakref.ref(x), ops.ver(x), outs, bit
            out = outs[i]
        if self._taken == (1 << len(self.members)) - 1 and self.linear_group is None:
            self.drop()                                      # everybody has theirs: keep nothing activation-sized
        return out

    def outputs(self):
        return self._outs


class LinearGroup:
    """The Linear4bit modules behind a TransformGroup (q_proj / k_proj / v_proj, up_proj / gate_proj): one GEMM launch per fused
    transform launch (linear4bit_multi falls back to the members' own forward where the multi-problem route does not apply)."""

    def __init__(self, members, transform_group):
        self.members = list(members)
        self.index = {id(m): i for i, m in enumerate(self.members)}
        self.tg = transform_group
        transform_group.linear_group = self
        self._ins = None
        self._ys = None
        self._taken = 0
        self._busy = False        # inside linear4bit_multi (whose fall-back calls the members' own forward)
        self.launches = 0
        self.served = 0
        self.fused = 0            # launches that carried the group's transform as their prologue

    def get(self, member, x):
        if self._busy or (self._ys is None and self.tg.outputs() is None):
            return None                                      # (no fused transform launch behind this call, e.g. decode-sized: on its own)
        i = self.index[id(member)]
        bit = 1 << i
        if self._ys is not None and self._ins[i] is x and not (self._taken & bit):
            y = self._ys[i]
            self._taken |= bit
            self.served += 1
        else:
            outs = self.tg.outputs()
            if outs is None or len(outs) != len(self.members) or outs[i] is not x:
                return None                                  # not this group's packed tensor: the member runs on its own
            self._ins = list(outs)
            self._busy = True
            try:
                lazy_x = self.tg._lazy_x
                if lazy_x is not None and self.tg._real is None:
                    self._ys = fused_transform_linear(lazy_x, self.tg.members, self.members)     # ONE launch: transform + projections
                    self.fused += 1
                else:
                    self._ys = linear4bit_multi(self.members, self.tg.materialise() if lazy_x is not None else self._ins)
            finally:
                self._busy = False
            self.launches += 1
            self._taken = bit
            y = self._ys[i]
        if self._taken == (1 << len(self.members)) - 1:
            self._ins, self._ys, self._taken = None, None, 0
            self.tg.drop()
        return y


def _is_trans(m):
    return isinstance(m, OnlineTrans) and m.trans == "matmul" and m.decompose


def fuse(model: torch.nn.Module, down_proj: bool = True, linears: bool = True, static_outputs: bool = False,
         fp6_image=None, fp6_image_budget_bytes=None, capture: bool = False) -> dict:
    """Walk ``model`` (built by the reference's deploy/transformers/modeling_llama.py with ``import flatquant_amd.deploy as deploy``)
    and install the fused launches described in this module's docstring. Idempotent. ``static_outputs=True`` additionally sets the
    modules' opt-in static output plans (results rewritten by the next call of the same module — safe for the reference's forward,
    which consumes every result before the module is called again). Returns what was installed:
    ``{"transform_groups": n, "linear_groups": n, "down_proj": n}``.
    ``fp6_image`` (True / False; None leaves the modules' policy alone) and ``fp6_image_budget_bytes`` set, on every Linear4bit of ``model``,
    whether the layer keeps its FP6 operand image (+0.75 B/param, the prefill GEMM 1.6x faster) and the cap on what all images may hold
    together — the memory side of the deployment in the one call that configures it (INTEGRATION.md "Memory").
    ``capture=True`` (round 6): every decoder layer (a module with ``self_attn`` and ``mlp`` children, the reference's LlamaDecoderLayer
    shape) serves its decode-sized calls from captured HIP graphs, transparently — deploy/graphed.py; reported as ``"graphed_layers"``."""
    report = {"transform_groups": 0, "linear_groups": 0, "down_proj": 0}
    if capture:
        report["graphed_layers"] = 0
        for mod in model.modules():
            if (isinstance(getattr(mod, "self_attn", None), torch.nn.Module) and isinstance(getattr(mod, "mlp", None), torch.nn.Module)
                    and not isinstance(mod.__dict__.get("forward"), GraphedDecode)):
                mod.forward = GraphedDecode(mod.forward)      # (an instance attribute: nn.Module.__call__ finds it before the class's)
                report["graphed_layers"] += 1
    if fp6_image is not None or fp6_image_budget_bytes is not None:
        for mod in model.modules():
            if isinstance(mod, Linear4bit):
                if fp6_image is not None:
                    mod.fp6_image = bool(fp6_image)
                if fp6_image_budget_bytes is not None:
                    mod.fp6_image_budget_bytes = int(fp6_image_budget_bytes)
    for mod in list(model.modules()):
        for tnames, lnames in ((("inp_trans_q", "inp_trans_k", "inp_trans_v"), ("q_proj", "k_proj", "v_proj")),
                               (("inp_trans_u", "inp_trans_g"), ("up_proj", "gate_proj"))):
            ts = [getattr(mod, n, None) for n in tnames]
            if not all(_is_trans(t) for t in ts) or any("_group" in t.__dict__ for t in ts):
                continue
            tg = TransformGroup(ts)
            for t in ts:
                t.__dict__["_group"] = tg
            report["transform_groups"] += 1
            ls = [getattr(mod, n, None) for n in lnames]
            if linears and all(isinstance(l, Linear4bit) for l in ls) and not any("_group" in l.__dict__ for l in ls):
                lg = LinearGroup(ls, tg)
                for l in ls:
                    l.__dict__["_group"] = lg
                report["linear_groups"] += 1
        dp = getattr(mod, "down_proj", None)
        if (down_proj and type(dp) is torch.nn.Sequential and len(dp) >= 2 and isinstance(dp[0], OnlineTrans)
                and isinstance(dp[1], Quantizer)):
            mod.down_proj = FusedSequential(*list(dp))
            report["down_proj"] += 1
    if static_outputs:
        for mod in model.modules():
            if isinstance(mod, (OnlineTrans, Quantizer, Linear4bit)):
                mod.static_outputs = True
    return report


def unfuse(model: torch.nn.Module) -> None:
    """Remove the groups ``fuse`` installed (``down_proj`` stays a FusedSequential: same children, same results)."""
    for mod in model.modules():
        if isinstance(mod.__dict__.get("forward"), GraphedDecode):
            mod.__dict__.pop("forward").release()
        if isinstance(mod, (OnlineTrans, Linear4bit)):
            mod.__dict__.pop("_group", None)
