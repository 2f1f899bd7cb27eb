"""Counterpart of deploy/nn/linear.py."""
import torch

from ... import ops
from .. import PackedQuantizedTensor
from ..functional.quantization import pack_i4


class Linear4bit(torch.nn.Module):
    """Symmetric 4-bit linear layer.  Reference: deploy/nn/linear.py:22-83 — same constructor, same buffers
    (``weight`` uint8 [out, in/2] packed nibbles, ``weight_scales`` [out, 1], optional ``bias``), so reference state
    dicts load unchanged.  ``forward`` takes the PackedQuantizedTensor the online transform / Quantizer produced and
    returns fp16: the reference runs deploy.matmul (CUTLASS int4 GEMM -> int32 in HBM) and deploy.sym_dequant as two
    launches (+ a torch add for the bias); here the dequantisation and the bias sit in the GEMM's epilogue
    (fq_int4_linear_f16), bit-identical to the two-step form.

    Memory. ``weight`` is 0.5 byte per parameter. Two optional operand images trade memory for speed, each built once
    per buffer version and kept until ``release_images()``:
      * the decode image (M <= 128 weight-streaming kernel, 4-9 us per linear instead of 46-144): +0.5 B/param, built on
        the first decode-sized call — ON by default (``Linear4bit.decode_image``, a class / instance attribute);
      * the FP6 image (BF6 operands on the FP6 matrix path for K % 128 == 0, N % 16 == 0: same bits out, the GEMM 1.6x
        faster than on the int8 path — 16384 x 4096 x 4096: 159 us against 285, profiles/r03_gemm_bf6_pipeline.txt):
        +0.75 B/param — kept by DEFAULT since round 5 (``Linear4bit.fp6_image``), built on the first prefill-sized call of a
        layer of >= ``fp6_min_out_features`` outputs, and only while ``fp6_image_min_free`` (10 %) of the device memory would stay
        free afterwards: a layer used for prefill AND decode then sits at 1.75 B/param, 3.5x the INT4 footprint — 122 GB for a
        70 B-parameter model on a 288 GB part; ``fp6_image = False`` (class or instance) returns to 0.5 - 1.0 B/param.
    Without the kept image a call of >= ``fp6_transient_rows`` (129: above the decode kernel's range) tokens still takes the FP6 path: the weights
    are converted INTO A TRANSIENT image for the call (12.6 MB for 4096 x 4096: ~8 us, freed with the call — the caching
    allocator hands the same block to the next layer), which costs 463 / rows of the GEMM's own time and leaves the
    resident footprint at 0.5 - 1.0 B/param. Both FP6 routes apply to layers of at least ``fp6_min_out_features`` outputs (2048:
    25-30 % faster than the int8 path there, slower for the 1024-wide k/v projections, where a call would also re-convert the
    whole weight); ``fp6_gemm = False`` turns every FP6 route off (int8 matrix path only). The routes are policy ATTRIBUTES of
    the class (or of an instance) — no environment variable is read (round 4)."""

    static_outputs = False   # (round 4, opt-in) decode-sized calls: a prepared launch with a static output buffer, ~6 us of Python instead of ~13
    fast_path = True         # (round 5, default) decode-sized calls as a C-side prepared call with a FRESH output: ~8 us of Python, nothing aliases
    decode_image = True   # class-wide policy switches (set on the class or on an instance)
    fp6_gemm = True       # False: no FP6 route at all
    fp6_image = True      # (round 5) kept by default for layers of >= fp6_min_out_features outputs while the device has room for it
    fp6_image_budget_bytes = None   # explicit cap on the bytes ALL layers' kept FP6 images may hold together (None: the free-memory rule below)
    _fp6_image_bytes_held = 0       # (class-wide account of the kept images)
    fp6_image_min_free = 0.10   # ... i.e. while at least this fraction of the device memory would stay free after building the image;
                                # otherwise (and with fp6_image = False) a prefill call converts the weights for the call (transient route)
    fp6_min_out_features = 2048   # narrower layers stay on the int8 matrix path (kept image AND transient route)
    fp6_transient_rows = 129    # calls with at least this many tokens convert the weights for the call when no image is kept (0: never).
                                # 129 = everything above the decode kernel's range: measured with both conversions inside the call
                                # (tools/microbench/gemm_small_m.py), 129 tokens x 4096 x 4096: 34.6 us against 45.6 on the int8 path,
                                # K = 11008: 55 against 111

    def __init__(self, in_features, out_features, bias=False, dtype=torch.float16):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.register_buffer("weight_scales", torch.zeros((self.out_features, 1), requires_grad=False))
        self.register_buffer("weight", torch.randint(1, 7, (self.out_features, self.in_features // 2),
                                                     dtype=torch.uint8, requires_grad=False))
        if bias:
            self.register_buffer("bias", torch.zeros((self.out_features), dtype=dtype))
        else:
            self.bias = None

    def _weight_image(self):
        """BF6 operand image of ``weight`` for the FP6 matrix path (csrc/fq_gemm_bf6.hip), rebuilt when the buffer is
        replaced or written in place. Used when the shape is covered and wide enough to amortise the per-call conversion
        of the activations (``fp6_min_out_features``)."""
        if not (self.fp6_gemm and self.fp6_image) or not ops.bf6_supported(self.out_features, self.in_features):
            return None
        if self.out_features < self.fp6_min_out_features:
            return None
        key = (self.weight.data_ptr(), ops.ver(self.weight), self.weight.device)
        if getattr(self, "_wimg_key", None) != key:
            self._drop_wimg()
            need = int(ops.lib.fq_bf6_blob_bytes(self.out_features, self.in_features))   # 0.75 B/param (rows padded to 32) on top of the 0.5 B/param of `weight`
            if self._image_room(need):
                self._wimg = ops.int4_to_bf6(self.weight, weights=True)
                Linear4bit._fp6_image_bytes_held += need
                self._wimg_need = need
                self._wimg_key = key
            # (a refusal is NOT remembered: the next prefill call asks again — memory may have been released, a budget raised)
        return self._wimg

    def _drop_wimg(self):
        if getattr(self, "_wimg", None) is not None:
            Linear4bit._fp6_image_bytes_held -= getattr(self, "_wimg_need", 0)
        self._wimg = None
        self._wimg_key = None

    def _image_room(self, need: int) -> bool:
        """May this layer keep ``need`` more bytes of FP6 image? With an explicit ``fp6_image_budget_bytes`` (class or instance attribute;
        deploy.fuse(model, fp6_image_budget_bytes=...)): while the images of ALL layers stay within it — the deployment's own number, which
        knows about KV-page growth and other processes. Without one: while ``fp6_image_min_free`` of the device would stay free, counting
        what torch's caching allocator holds but has not handed out as free (mem_get_info alone does not see it)."""
        budget = self.fp6_image_budget_bytes
        if budget is not None:
            return Linear4bit._fp6_image_bytes_held + need <= budget
        dev = self.weight.device
        free, total = torch.cuda.mem_get_info(dev)
        free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        return free - need >= self.fp6_image_min_free * total

    def _scales16(self):
        """(weight_scales as a flat fp16 vector, bias as fp16 or None), converted once per buffer version: the buffers are
        fp32 unless the model was built under a fp16 default dtype, and a conversion launch per call costs more than the
        decode-sized GEMM itself."""
        b = self.bias
        key = (self.weight_scales.data_ptr(), ops.ver(self.weight_scales), None if b is None else (b.data_ptr(), ops.ver(b)))
        if getattr(self, "_s16_key", None) != key:
            self._s16 = (self.weight_scales.reshape(-1).to(torch.float16).contiguous(),
                         None if b is None else b.to(torch.float16).contiguous())
            self._s16_key = key
        return self._s16

    def _decode_image(self):
        """``weight`` in MFMA fragment order for the decode-sized (M <= 128) weight-streaming kernel; cached like the
        FP6 image. ``decode_image = False`` turns the path off."""
        if not self.decode_image or self.in_features % 64:
            return None
        key = (self.weight.data_ptr(), ops.ver(self.weight), self.weight.device)
        if getattr(self, "_dimg_key", None) != key:
            self._dimg = ops.int4_to_frag(self.weight)
            self._dimg_key = key
        return self._dimg

    def release_images(self):
        """Drop the cached operand images and fp16 scale / bias copies (they are rebuilt on demand)."""
        self._drop_wimg()
        for name in ("_wimg", "_wimg_key", "_wimg_need", "_dimg", "_dimg_key", "_s16", "_s16_key"):
            if hasattr(self, name):
                delattr(self, name)

    def image_bytes(self):
        """Bytes currently held by the cached operand images (on top of ``weight``)."""
        return sum(t.numel() * t.element_size() for t in (getattr(self, "_wimg", None), getattr(self, "_dimg", None))
                   if t is not None)

    def forward(self, x):
        assert isinstance(x, PackedQuantizedTensor)  # quantized input is given (linear.py:45; a fused group's lazy form is one too)
        grp = self.__dict__.get("_group")      # (deploy.fuse: the projections of one attention / MLP run as one GEMM launch)
        if grp is not None:
            y = grp.get(self, x)
            if y is not None:
                return y
        q, scales_x = x.quantized_x, x.scales_x
        if self.static_outputs:
            st = self.__dict__.get("_plan_state")
            if st is not None:
                bf = self._buffers
                w = bf["weight"]
                if (st[0] is w and st[1] == ops.ver(w) and st[2] == ops.ver(bf["weight_scales"])
                        and st[3] == (-1 if self.bias is None else ops.ver(self.bias)) and st[4] == ops.cache_epoch()
                        and scales_x.dtype == torch.float16 and scales_x.is_contiguous() and scales_x.numel() == st[5].args[5]):
                    return st[5].run2(q, scales_x)      # (run() checks q's shape, dtype, device and contiguity)
                del self.__dict__["_plan_state"]
        lead = q.shape[:-1]
        rows = q.numel() // q.shape[-1]
        if q.is_cuda and ops.skinny_supported(rows, self.in_features):
            dimg = self._decode_image()
            if dimg is not None and self.static_outputs and q.is_contiguous() and scales_x.is_contiguous() and scales_x.dtype == torch.float16:
                # (round 4, opt-in) a prepared launch with a STATIC output buffer (ops.LaunchPlan): rewritten by the next call
                ws16, b16 = self._scales16()
                plan = ops.skinny_linear_plan(q.reshape(rows, -1), dimg, ws16, b16, self.out_features)
                plan.shape = q.shape                         # (run() compares the caller's own shape: no reshape per call)
                plan.result = plan.outputs.view(*lead, self.out_features)
                bf = self._buffers
                self.__dict__["_plan_state"] = (bf["weight"], ops.ver(bf["weight"]), ops.ver(bf["weight_scales"]),
                                                -1 if self.bias is None else ops.ver(self.bias), ops.cache_epoch(), plan)
                return plan.run2(q, scales_x)
            if (dimg is not None and self.fast_path and q.is_contiguous() and scales_x.is_contiguous() and scales_x.dtype == torch.float16
                    and scales_x.numel() == rows and not torch.cuda.is_current_stream_capturing()):
                # (round 5, default) the decode-sized call as a C-side prepared call with a FRESH output (ops.FreshPlan)
                bf = self._buffers
                w = bf["weight"]
                st = self.__dict__.get("_fresh_state")
                if st is None:
                    st = self.__dict__["_fresh_state"] = ops.FreshPlanSet()
                plan = st.lookup((id(w), ops.ver(w), ops.ver(bf["weight_scales"]), -1 if self.bias is None else ops.ver(self.bias),
                                  ops.cache_epoch()), q)
                if plan is None:
                    ws16, b16 = self._scales16()
                    plan = st.add(q, ops.skinny_linear_fresh_plan(q, dimg, ws16, b16, self.out_features, lead + (self.out_features,)), (w,))
                if rows >= 33 or self.in_features >= 8192:   # (a captured step of this size takes the split launch: its zeroed workspace must exist ON THIS STREAM before the capture)
                    ops.skinny_split_workspace(rows, self.out_features, self.in_features, q.device)
                return plan.run_linear(q, scales_x)
            if dimg is not None:
                ws16, b16 = self._scales16()
                y = ops.int4_skinny_linear(q.reshape(rows, -1).contiguous(), scales_x.reshape(-1).contiguous(), dimg,
                                           ws16, b16, self.out_features)
                return y.view(*lead, self.out_features)
        if q.is_cuda and self.fp6_gemm and ops.bf6_supported(self.out_features, self.in_features):
            wimg = self._weight_image()                     # the kept image, or None: converted for this call (transient)
            if wimg is not None or (self.fp6_transient_rows and rows >= self.fp6_transient_rows
                                    and self.out_features >= self.fp6_min_out_features):
                ws16, b16 = self._scales16()
                y = ops.int4_linear_fp6(q.reshape(-1, q.shape[-1]).contiguous(), scales_x.reshape(-1).contiguous(), self.weight, wimg,
                                        ws16, b16)
                return y.view(*lead, self.out_features)
        ws16, b16 = self._scales16()
        y = ops.int4_linear(q.reshape(-1, q.shape[-1]).contiguous(), scales_x.reshape(-1).contiguous(),
                            self.weight, ws16, b16)
        return y.view(*lead, self.out_features)

    @staticmethod
    def from_float(module: torch.nn.Linear, weight_scales=None):
        """linear.py:58-83: round(weight / weight_scales) packed two per byte."""
        weight_matrix = module.weight.data
        int_module = Linear4bit(module.in_features, module.out_features, bias=module.bias is not None,
                                dtype=weight_matrix.dtype).to(weight_matrix.dtype)
        if weight_scales is not None:
            assert weight_scales.shape == (module.out_features, 1), "weight_scales should have shape (out_features, 1)"
            weight_matrix = weight_matrix.cuda()
            int_module.weight_scales.copy_(weight_scales.to(weight_matrix.dtype))
            int_rounded_weight = (weight_matrix / weight_scales.cuda()).round()
            int_module.weight.copy_(pack_i4(int_rounded_weight.to(torch.int8)).cpu())
            if module.bias is not None:
                int_module.bias.copy_(module.bias.data)
        return int_module


WIDE_DECODE_ROWS = 33          # linear4bit_multi: from this many rows a group of >= WIDE_DECODE_FEATURES output features takes the FP6 tile kernel
WIDE_DECODE_FEATURES = 16384


def linear4bit_multi(modules, inputs):
    """Several ``Linear4bit`` modules of one decoder layer that see the same token count and input width — q_proj / k_proj / v_proj, or
    up_proj / gate_proj (deploy/transformers/modeling_llama.py:66-78, 268-276: one GEMM + dequant per projection in the reference) — each
    on its own PackedQuantizedTensor, as ONE GEMM launch on the FP6 matrix path (round 4, fq_int4_linear_fp6_multi_f16): at prefill sizes
    of a few thousand tokens a single projection does not fill the chip. Returns one fp16 tensor per module, bit-identical to
    ``m(x)``. Decode-sized inputs (<= 128 rows) take ONE launch of the weight-streaming kernel instead (round 5). Falls back to the modules' own
    forward when neither route applies to every one of them (shapes the FP6 path does not cover, fp6_gemm off, layers narrower than
    fp6_min_out_features without a kept image; no decode image)."""
    assert len(modules) == len(inputs) and len(modules) >= 1
    q0 = inputs[0].quantized_x
    rows = q0.numel() // q0.shape[-1]
    # (round 6) 33 .. 128 rows on a WIDE group (up + gate of a decoder layer: >= 16384 features together): the weight-streaming kernel is
    # compute-bound there (int8 MFMAs behind a nibble unpack: 31 / 44 / 54 us at 64 / 96 / 128 rows of Llama-3-8B's up + gate), the FP6 tile
    # kernel on its 128 x 128 decode tile is not (26.8 / 27.9 / 29.9 us, activation conversion included; profiles/r06_decode_gemm_routes.txt).
    # Only with KEPT FP6 images; every narrower group and every lone projection stays on the weight-streaming kernel (faster there).
    wide = (2 <= len(modules) <= 4 and q0.is_cuda and WIDE_DECODE_ROWS <= rows <= 128 and sum(m.out_features for m in modules) >= WIDE_DECODE_FEATURES
            and all(m.fp6_gemm and m.fp6_image and ops.bf6_supported(m.out_features, m.in_features) and m.in_features == modules[0].in_features
                    for m in modules)
            and all(m._weight_image() is not None for m in modules))
    if not wide and 2 <= len(modules) <= 4 and q0.is_cuda and ops.skinny_supported(rows, modules[0].in_features):
        # (round 5) decode-sized: ONE launch of the weight-streaming kernel over the members' weight images (a decode-sized launch
        # costs ~4 us whatever it streams — fq_int4_skinny_linear_multi_f16)
        problems = []
        for m, x in zip(modules, inputs):
            assert type(x) == PackedQuantizedTensor
            dimg = m._decode_image() if m.in_features == modules[0].in_features and x.quantized_x.shape == q0.shape else None
            if dimg is None or x.scales_x.dtype != torch.float16 or x.scales_x.numel() != rows:
                problems = None
                break
            ws16, b16 = m._scales16()
            q = x.quantized_x
            problems.append((q.reshape(rows, -1).contiguous(), x.scales_x.reshape(-1).contiguous(), dimg, ws16, b16))
        if problems is not None:
            ys = ops.int4_skinny_linear_multi(problems)
            lead = q0.shape[:-1]
            return [y.view(*lead, m.out_features) for y, m in zip(ys, modules)]
    ok = 1 <= len(modules) <= 4 and q0.is_cuda and (wide or not ops.skinny_supported(rows, modules[0].in_features))
    for m, x in zip(modules, inputs):
        assert type(x) == PackedQuantizedTensor
        ok = ok and m.fp6_gemm and ops.bf6_supported(m.out_features, m.in_features) and x.quantized_x.shape == q0.shape
        ok = ok and m.in_features == modules[0].in_features
        # (a narrow projection — the 1024-wide k / v of a GQA model — rides along in the launch of a wide one: fp6_min_out_features is
        #  asked of the widest module of the group, not of each)
        ok = ok and (m._weight_image() is not None or bool(m.fp6_transient_rows and rows >= m.fp6_transient_rows))
    ok = ok and max(m.out_features for m in modules) >= min(m.fp6_min_out_features for m in modules)
    if not ok:
        return [m(x) for m, x in zip(modules, inputs)]
    problems = []
    for m, x in zip(modules, inputs):
        ws16, b16 = m._scales16()
        q = x.quantized_x
        problems.append((q.reshape(-1, q.shape[-1]).contiguous(), x.scales_x.reshape(-1).contiguous(), m.weight, m._weight_image(), ws16, b16))
    ys = ops.int4_linear_fp6_multi(problems)
    lead = q0.shape[:-1]
    return [y.view(*lead, m.out_features) for y, m in zip(ys, modules)]


FUSED_DECODE_ROWS = 8          # fused_transform_linear: up to this many tokens the transform runs as the GEMM launch's prologue (measured,
                               # profiles/r06_fused_decode.txt: q / k / v 11.0 -> 7.0 us at 1 token, 12.8 -> 10.3 at 8; level at 16)


def _fused_decode_problems(x, transforms, linears):
    """The (w_image, w_scale, bias) triples of ops.kron64_linear_multi when the fused decode launch covers this call, else None."""
    from ... import ops as _ops
    if not (x.is_cuda and x.dtype == torch.float16 and x.shape[-1] == 4096 and 1 <= len(linears) <= 4 and len(linears) == len(transforms)):
        return None
    rows = x.numel() // 4096
    if not 1 <= rows <= min(FUSED_DECODE_ROWS, _ops.FUSED_DECODE_MAX_ROWS):
        return None
    first = transforms[0]
    if not (first.trans == "matmul" and first.decompose and tuple(first.left_matrix.shape) == (64, 64) and tuple(first.right_matrix.shape) == (64, 64)
            and first.left_matrix.dtype == torch.float16):
        return None
    problems = []
    for m in linears:
        dimg = m._decode_image() if (m.in_features == 4096 and m.out_features % 32 == 0) else None
        if dimg is None:
            return None
        ws16, b16 = m._scales16()
        problems.append((dimg, ws16, b16))
    return problems


def fused_transform_linear(x, transforms, linears, norm=None):
    """``[m(t(norm(x))) for t, m in zip(transforms, linears)]`` — the OnlineTrans modules of one attention / MLP (sharing their Kronecker
    pair, their own clip factors: fused_forward's contract) and the Linear4bit projections behind them — with THE TRANSFORM AS THE GEMM'S
    PROLOGUE where that pays (round 6): up to FUSED_DECODE_ROWS tokens of d = 4096 run as ONE launch (fq_kron64_linear_multi_f16: every
    workgroup transforms and quantises the tokens itself while its weights are in flight; the packed activations never reach memory);
    anything else as fused_forward + linear4bit_multi. Bit-identical either way (tests/test_gpu_fused_decode.py)."""
    from ... import ops as _ops
    from ..._lib import FQ_NO_CLAMP0, FQ_ROUND_Y_F16
    from ..functional.online_trans import deploy_kron_flags
    from .online_trans import fused_forward
    problems = _fused_decode_problems(x, transforms, linears)
    if problems is not None:
        first = transforms[0]
        for t in transforms:
            if not (t.trans == "matmul" and t.decompose and t.left_matrix.data_ptr() == first.left_matrix.data_ptr()
                    and t.right_matrix.data_ptr() == first.right_matrix.data_ptr()):
                raise RuntimeError("fused_transform_linear: the transforms must share their left/right matrices")
        sigs = [_ops.sigmoid_pair(t.clip_factor_a_max, t.clip_factor_a_min) for t in transforms]
        flags = deploy_kron_flags(64, 64) & (FQ_NO_CLAMP0 | FQ_ROUND_Y_F16)
        ys = _ops.kron64_linear_multi(x.contiguous(), first.left_matrix.contiguous(), first.right_matrix.contiguous(), sigs, problems,
                                      eps=None if norm is None else float(norm.eps), flags=flags)
        if ys is not None:
            lead = x.shape[:-1]
            return [y.view(*lead, m.out_features) for y, m in zip(ys, linears)]
    return linear4bit_multi(linears, fused_forward(x, transforms, norm=norm))
