"""Counterpart of deploy/nn/online_trans.py."""
import torch

from ... import ops
from .. import functional
from ...flatquant.function_utils import get_decompose_dim  # noqa: F401


class OnlineTrans(torch.nn.Module):
    """Online transform in front of a 4-bit linear.  Reference: deploy/nn/online_trans.py:18-67 — same
    constructor, same buffers (``had_rem_dim`` | ``left_matrix right_matrix diag_scale``;
    ``clip_factor_a_max/min`` default 1.0) so reference state dicts load unchanged.

      trans="had":    fp16 tensor out (QuaRot baseline), one HIP launch instead of FWHT + bmm.
      trans="matmul": PackedQuantizedTensor out (transform + INT4 quant fused), one HIP launch.
    """

    def __init__(self, trans_dim, force_fp32=False, trans="had", decompose=True, lac=False):
        super().__init__()
        self.fp32_trans = force_fp32
        self.trans = trans
        self.decompose = decompose
        self.trans_dim = trans_dim
        if trans == "had":
            had_rem_dim, self.rem_dim = functional.online_trans.get_hadK(trans_dim)
            if had_rem_dim is not None:
                self.register_buffer("had_rem_dim", had_rem_dim)
                if not self.fp32_trans:
                    self.had_rem_dim = self.had_rem_dim.to(torch.float16)
            else:
                self.had_rem_dim = None
        elif trans == "matmul":
            if decompose:
                left_size, right_size = get_decompose_dim(trans_dim)
                self.register_buffer("left_matrix", torch.randn([left_size, left_size], dtype=torch.float16))
                self.register_buffer("right_matrix", torch.randn([right_size, right_size], dtype=torch.float16))
                self.register_buffer("diag_scale", torch.randn([trans_dim], dtype=torch.float16))
            else:
                self.register_buffer("right_matrix", torch.randn([trans_dim, trans_dim], dtype=torch.float16))
        self.lac = lac
        ops.invalidate_on_load(self)   # matrices / clip factors loaded later: no stale fragment images or host scalars
        self.register_buffer("clip_factor_a_max", torch.tensor(1.0))
        self.register_buffer("clip_factor_a_min", torch.tensor(1.0))

    static_outputs = False   # (round 4, opt-in, class or instance attribute) decomposed matmul transform: a prepared launch with STATIC
                             # output buffers (ops.LaunchPlan) — the returned PackedQuantizedTensor is rewritten by the next call, as under
                             # a captured graph. ~6 us of Python per call instead of ~20 (tools/host_overhead.py); same bits out.

    def _planned(self, x):
        # (buffers straight from the module's dict: nn.Module.__getattr__ costs ~0.5 us per name, four names per call)
        bf = self._buffers
        L, R, cmax, cmin = bf["left_matrix"], bf["right_matrix"], bf["clip_factor_a_max"], bf["clip_factor_a_min"]
        st = self.__dict__.get("_plan_state")
        if (st is None or st[0] is not L or st[1] != ops.ver(L) or st[2] is not R or st[3] != ops.ver(R) or st[4] != ops.ver(cmax)
                or st[5] != ops.ver(cmin) or st[6] != ops.cache_epoch() or st[7].shape != x.shape or st[7].dtype != x.dtype
                or st[7].device != x.device):
            from .. import PackedQuantizedTensor
            bsz, seq_len, _ = x.shape
            sig = ops.sigmoid_pair(cmax, cmin)
            plan = ops.kron_plan(x.contiguous(), L.contiguous(), R.contiguous(), [sig],
                                 functional.online_trans.deploy_kron_flags(L.shape[0], R.shape[0]))
            o = plan.outputs
            plan.result = PackedQuantizedTensor(o.q[0].reshape(bsz, seq_len, -1), o.scale[0].reshape(bsz, 1, seq_len))
            st = (L, ops.ver(L), R, ops.ver(R), ops.ver(cmax), ops.ver(cmin), ops.cache_epoch(), plan)
            self.__dict__["_plan_state"] = st
        return st[7].run(x)

    fast_path = True         # (round 5, default) the decomposed matmul transform as a C-side prepared call (ops.FreshPlan): FRESH outputs every
                             # call — nothing aliases, unlike static_outputs — and ~8 us of Python instead of ~20. False: the general entry point.

    def _fresh(self, x):
        bf = self._buffers
        L, R = bf["left_matrix"], bf["right_matrix"]
        # (the reference's loader replaces the clip buffers by Python floats: modeling_llama.py:532-538 — an attribute then, not a buffer)
        cmax, cmin = bf.get("clip_factor_a_max"), bf.get("clip_factor_a_min")
        if cmax is None or cmin is None:
            cmax, cmin = self.clip_factor_a_max, self.clip_factor_a_min
        kmax = (id(cmax), ops.ver(cmax)) if isinstance(cmax, torch.Tensor) else cmax
        kmin = (id(cmin), ops.ver(cmin)) if isinstance(cmin, torch.Tensor) else cmin
        st = self.__dict__.get("_fresh_state")
        if st is None:
            st = self.__dict__["_fresh_state"] = ops.FreshPlanSet()
        plan = st.lookup((id(L), ops.ver(L), id(R), ops.ver(R), kmax, kmin, ops.cache_epoch()), x)
        if plan is None:
            bsz, seq_len, d = x.shape
            plan = st.add(x, ops.kron_fresh_plan(x, L.contiguous(), R.contiguous(), ops.sigmoid_pair(cmax, cmin),
                                                 functional.online_trans.deploy_kron_flags(L.shape[0], R.shape[0]),
                                                 (bsz, seq_len, d // 2), (bsz, 1, seq_len)), (L, R, cmax, cmin))
        from .. import PackedQuantizedTensor
        q, sc = plan.run(x)
        return PackedQuantizedTensor(q, sc)

    def _static_ok(self, x):
        """the prepared launch takes exactly what its plan was built from: a contiguous 3-D CUDA tensor, outside stream capture (a plan
        built inside a capture would leave its outputs and workspace in the capture's pool while later eager calls reuse it)"""
        return x.dim() == 3 and x.is_contiguous() and x.is_cuda and not torch.cuda.is_current_stream_capturing()

    def forward(self, x, quantizer=None, norm=None, up=None):
        """``up`` (extension, optional): ``x`` is then x_gate and the transform consumes x_up * silu(x_gate)
        (FlatQuantLlamaMLP.forward, modeling_llama.py:277-279) formed inside the launch: trans="matmul" (decomposed)
        returns the PackedQuantizedTensor as always, trans="had" needs ``quantizer`` as well.
        ``norm`` (extension, optional): the deploy.nn.RMSNorm whose output this transform consumes; pass its INPUT as
        ``x`` and the normalisation runs inside the transform + quantisation launch (trans="matmul", decompose).
        ``quantizer`` (extension, optional): the deploy.nn.Quantizer that consumes this transform's output. For
        trans="had" the two then run as ONE launch and a PackedQuantizedTensor comes back (the Quantizer passes packed
        inputs through, quantization.py:14), bit-identical to calling them one after the other on the FWHT route, to rounding noise
        on the matrix-pipe routes. Quantizer(lac=True): every width; Quantizer(lac=False) (the reference's options.trans == "had"
        model): the widths of the structured kernel and of the tall Kronecker kernel (ops.hadamard_quantizer), two launches elsewhere."""
        grp = self.__dict__.get("_group")     # (deploy.fuse: the transforms of one attention / MLP run as one launch)
        if grp is not None and quantizer is None and norm is None and up is None:
            out = grp.get(self, x)
            if out is not None:
                return out
        if up is not None and norm is not None:
            raise RuntimeError("OnlineTrans: up= and norm= are exclusive")
        if self.trans == "had":
            if quantizer is not None and getattr(quantizer, "lac", False) and not self.fp32_trans:
                from ... import ops
                from .. import PackedQuantizedTensor
                sig = ops.sigmoid_pair_f16(quantizer.clip_factor_a_max, quantizer.clip_factor_a_min)   # (device semantics)
                q, s = ops.hadamard_quant(x.contiguous(), self.rem_dim, self.had_rem_dim, sig,
                                          up=None if up is None else up.contiguous())
                return PackedQuantizedTensor(q, s.reshape(-1, 1))   # (the lac Quantizer's [rows, 1] scales, quantization.py:16-28)
            if (quantizer is not None and not getattr(quantizer, "lac", False) and not self.fp32_trans and x.dtype == torch.float16
                    and x.is_cuda and self.had_rem_dim is not None and float(quantizer.input_clip_ratio) > 0.0):
                # Quantizer(lac=False): the pair the reference's options.trans == "had" model builds (modeling_llama.py:244-252). One launch
                # where a fused route covers the width (the structured kernel: 14336, 28672, ...; the tall Kronecker kernel: 11008, 8960, ...):
                # fp16(max|y| / 7) * ratio, no zero guard, scales of the reference's shape [..., 1, seq] (quantization.py:30); the other
                # widths run the two launches below
                from ... import ops
                from .. import PackedQuantizedTensor
                qs = ops.hadamard_quantizer(x.contiguous(), self.rem_dim, self.had_rem_dim, float(quantizer.input_clip_ratio),
                                            up=None if up is None else up.contiguous())
                if qs is not None:
                    return PackedQuantizedTensor(qs[0], qs[1].reshape(x.shape[:-1]).unsqueeze(1))
            if up is not None:
                from ... import ops
                x = ops.silu_mul(x.contiguous(), up.contiguous())
            if self.fp32_trans:
                # the reference up-casts and returns the fp32 transform (online_trans.py:55-59): the butterflies and the scaling in
                # fp32 with NO rounding to fp16 (fq_fwht_f32_f16), the fp32 K x K factor as the reference's own GEMM (round 4)
                from ... import ops
                if x.dtype not in (torch.float16, torch.bfloat16):
                    raise TypeError("OnlineTrans(force_fp32=True): fp16 / bf16 activations (their up-cast is exact)")
                return ops.hadamard_fp32(x.contiguous(), self.rem_dim, self.had_rem_dim)
            return functional.matmul_hadU_cuda(x, self.had_rem_dim, self.rem_dim)
        if self.trans == "matmul" and norm is not None:
            if not (self.decompose and hasattr(self, "left_matrix")):
                raise RuntimeError("OnlineTrans: norm= is fused for the decomposed (Kronecker) transform only")
            from ... import ops
            from ..._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED
            from .. import PackedQuantizedTensor
            bsz, seq_len, _ = x.shape
            sig = ops.sigmoid_pair(self.clip_factor_a_max, self.clip_factor_a_min)
            o = ops.rmsnorm_kron_quant(x.contiguous(), float(norm.eps), self.left_matrix.contiguous(),
                                       self.right_matrix.contiguous(), [sig],
                                       functional.online_trans.deploy_kron_flags(self.left_matrix.shape[0], self.right_matrix.shape[0]))
            return PackedQuantizedTensor(o.q[0].reshape(bsz, seq_len, -1), o.scale[0].reshape(bsz, 1, seq_len))
        if self.trans == "matmul" and up is not None:
            from ... import ops
            if not (self.decompose and hasattr(self, "left_matrix") and hasattr(self, "right_matrix")):
                x, up = ops.silu_mul(x.contiguous(), up.contiguous()), None
            else:
                from ..._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED
                from .. import PackedQuantizedTensor
                bsz, seq_len, _ = x.shape
                sig = ops.sigmoid_pair(self.clip_factor_a_max, self.clip_factor_a_min)
                o = ops.silu_mul_kron_quant(x.contiguous(), up.contiguous(), self.left_matrix.contiguous(),
                                            self.right_matrix.contiguous(), [sig],
                                            functional.online_trans.deploy_kron_flags(self.left_matrix.shape[0], self.right_matrix.shape[0]))
                return PackedQuantizedTensor(o.q[0].reshape(bsz, seq_len, -1), o.scale[0].reshape(bsz, 1, seq_len))
        # (quantizer=...: a matmul transform returns the packed tensor whatever Quantizer follows — it passes packed inputs through)
        if self.trans == "matmul" and self.decompose and "left_matrix" in self._buffers and "right_matrix" in self._buffers \
                and self._static_ok(x):
            if self.static_outputs:
                return self._planned(x)
            if self.fast_path:
                return self._fresh(x)
        if self.trans == "matmul":
            invs = []
            if hasattr(self, "left_matrix"):
                invs.append(self.left_matrix)
            if hasattr(self, "right_matrix"):
                invs.append(self.right_matrix)
            return functional.online_trans.kronecker_matmul(x, invs, self.clip_factor_a_max, self.clip_factor_a_min)
        return x


def fused_forward(x, transforms, norm=None):
    """Several decomposed ``OnlineTrans`` modules that share one Kronecker factor pair and differ only in their clip
    factors — inp_trans_q / inp_trans_k / inp_trans_v, inp_trans_u / inp_trans_g after the reference's loader handed
    them the attention's / MLP's matrices (modeling_llama.py:518-529) — applied to the same ``x`` in ONE launch:
    the token is read and transformed once and quantised once per clip pair (the reference runs the whole transform
    per projection, modeling_llama.py:66-77, 270-271). ``norm``: see ``OnlineTrans.forward``.
    Returns one PackedQuantizedTensor per module, each bit-identical to ``t(x)``."""
    from ... import ops
    from ..._lib import FQ_MAX_CLIPS, FQ_NO_CLAMP0, FQ_OUT_PACKED
    from .. import PackedQuantizedTensor
    first = transforms[0]
    if not 1 <= len(transforms) <= FQ_MAX_CLIPS:
        raise ValueError(f"fused_forward: 1..{FQ_MAX_CLIPS} transforms")
    for t in transforms:
        if not (t.trans == "matmul" and t.decompose and hasattr(t, "left_matrix") and hasattr(t, "right_matrix")):
            raise RuntimeError("fused_forward: decomposed matmul transforms only")
        if t.left_matrix.data_ptr() != first.left_matrix.data_ptr() or t.right_matrix.data_ptr() != first.right_matrix.data_ptr():
            raise RuntimeError("fused_forward: the transforms must share their left/right matrices")
    bsz, seq_len, _ = x.shape
    sigs = [ops.sigmoid_pair(t.clip_factor_a_max, t.clip_factor_a_min) for t in transforms]
    left, right = first.left_matrix.contiguous(), first.right_matrix.contiguous()
    flags = functional.online_trans.deploy_kron_flags(left.shape[0], right.shape[0])   # (as each module's own forward)
    if norm is not None:
        o = ops.rmsnorm_kron_quant(x.contiguous(), float(norm.eps), left, right, sigs, flags)
    else:
        o = ops.kron_quant(x.contiguous(), left, right, sigs, flags)
    return [PackedQuantizedTensor(o.q[i].reshape(bsz, seq_len, -1), o.scale[i].reshape(bsz, 1, seq_len))
            for i in range(len(transforms))]


class FusedSequential(torch.nn.Sequential):
    """``Sequential(OnlineTrans, Quantizer, ...)`` — the reference's down_proj (deploy/transformers/modeling_llama.py:248-253: OnlineTrans(had),
    Quantizer, Linear4bit) — with the first two modules run as ONE launch: ``self[0](x, quantizer=self[1])`` returns the
    PackedQuantizedTensor, the Quantizer passes packed inputs through (quantization.py:14), the remaining modules follow. Same children,
    same state-dict keys as the ``nn.Sequential`` it replaces: ``model.mlp.down_proj = FusedSequential(*model.mlp.down_proj)`` is the
    whole change for a model built by the reference's code. ``forward(x, up=None)``: with ``up``, x is x_gate and the transform
    consumes x_up * silu(x_gate) inside the launch (modeling_llama.py:277-279)."""

    def forward(self, x, up=None):
        mods = list(self)
        if len(mods) >= 2 and isinstance(mods[0], OnlineTrans):
            x = mods[0](x, quantizer=mods[1], up=up)
            rest = mods[1:]
        else:
            if up is not None:
                from ... import ops
                x = ops.silu_mul(x.contiguous(), up.contiguous())
            rest = mods
        for m in rest:
            x = m(x)
        return x

