"""CPU oracle for the FlatQuant online-transform + INT4-quantisation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``flatquant_amd/`` imports this module; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may.  It is the checker, never
the thing measured or shipped.

It restates, in numpy with explicitly pinned IEEE arithmetic, what the reference computes on the path
(citations are file:line in the FlatQuant repository):

  get_decompose_dim      flatquant/function_utils.py:11-21
  pack_i4 / unpack_i4    deploy/functional/quantization.py:49-56 / :60-82
  kron_transform         flatquant/flat_utils.py:6-17          (x @ hadR -> fp16, hadL.T @ . -> fp32)
  token_scale/quantize   flatquant/quant_utils.py:85-107, :19-22 ; deploy/kernels/kron_matmul.py:91-123
  rowquant               deploy/nn/quantization.py:13-36 (Quantizer) ; quant_utils.py:71-119
  sym_quant              deploy/kernels/quant.cu:13-47   (fp16 division, rn, clamp, low nibble = even col)
  sym_dequant            deploy/kernels/quant.cu:5-10, :66-85
  hadamard               flatquant/hadamard_utils.py:89-110 (matmul_hadU), :132-141 (matmul_hadU_cuda)
  block_quant            deploy/kernels/block_matmul.py:29-104

Parity pinning: ``tests/golden/*.npz`` are produced by ``tools/gen_golden.py`` from the imported
reference (path A on CPU; path-B Triton kernels under TRITON_INTERPRET=1) and
``tests/test_oracle_golden.py`` checks this module against them.  Two pieces have no reference artefact
that can run here and are therefore "parity unpinned" beyond their published algorithm:
``sym_quant``/``sym_dequant`` (quant.cu needs nvcc + CUTLASS) and the FWHT of the un-vendored third-party
``fast_hadamard_transform`` (Dao-AILab/fast-hadamard-transform, pin unknown: empty submodule) — for the
latter ``matmul_hadU`` (in-repo, same mathematics) is the anchor.

Arithmetic pinned (DESIGN.md "Pinned arithmetic"):
  * GEMMs take fp16 operands, multiply exactly, accumulate in fp32.  The accumulation ORDER is a
    parameter (`groups`): each group of contraction indices is summed exactly and added to the running
    fp32 accumulator with one rounding.  ``None`` = one group (exact dot product, rounded once).
  * intermediate U = x @ hadR is rounded to fp16 (flat_utils.py:15 returns the input dtype).
  * statistics, scale = m/7 and y/scale in fp32 with correctly rounded division; rint = half-to-even.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np

F16 = np.float16
F32 = np.float32

# --------------------------------------------------------------------------------------------------
# bfloat16 (round 3). The reference's path A is dtype-generic and its pipeline feeds bf16 on Llama-3 / Qwen / DeepSeek
# (flatquant/model_utils.py:20 torch_dtype='auto'; main_dpskv3.py:395). numpy has no bf16: a bf16 tensor is held here as a
# float32 array whose values are bf16-representable (tests/golden stores the 16-bit patterns: bf16_from_bits / bf16_bits).
# Every function below that rounds to "the activation's dtype" takes lowp = "f16" | "bf16" and rounds through _rnd.
# --------------------------------------------------------------------------------------------------
def bf16_round(a) -> np.ndarray:
    """fp32 -> bf16 (round to nearest even, what torch's .to(bfloat16) and v_cvt_pk_bf16_f32 do) -> fp32."""
    a = np.ascontiguousarray(np.asarray(a, dtype=F32))
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    out = u.astype(np.uint32).view(F32).reshape(a.shape)
    return np.where(np.isnan(a), a, out).astype(F32)


def bf16_from_bits(bits) -> np.ndarray:
    return (np.asarray(bits).astype(np.uint32) << 16).view(F32)


def bf16_bits(a) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(a, dtype=F32))
    assert np.array_equal(bf16_round(a), a, equal_nan=True), "not bf16-representable"
    return (a.view(np.uint32) >> 16).astype(np.uint16)


def _rnd(a, lowp="f16") -> np.ndarray:
    """Round an fp32 array to the 16-bit activation dtype and return it as fp32."""
    if lowp == "bf16":
        return bf16_round(a)
    with np.errstate(over="ignore"):
        return np.asarray(a, dtype=F32).astype(F16).astype(F32)


def _as_lowp(a, lowp="f16") -> np.ndarray:
    """An input tensor of the activation dtype as fp32 values (fp16 arrays convert exactly; bf16 inputs already are
    bf16-representable fp32 arrays — asserted)."""
    if lowp == "bf16":
        a = np.asarray(a, dtype=F32)
        assert np.array_equal(bf16_round(a), a, equal_nan=True), "bf16 input holds values that are not bf16-representable"
        return a
    return np.asarray(a, dtype=F16).astype(F32)


# --------------------------------------------------------------------------------------------------
# integer helpers
# --------------------------------------------------------------------------------------------------
def get_decompose_dim(n: int):
    """flatquant/function_utils.py:11-21 — closest factor pair (a-b, a+b) with a^2 - n = b^2."""
    a = int(math.sqrt(n))
    if a * a < n:
        a += 1
    while True:
        tmp = a * a - n
        b = int(math.sqrt(tmp))
        if b * b == tmp:
            break
        a += 1
    return a - b, a + b


def pack_i4(q: np.ndarray) -> np.ndarray:
    """deploy/functional/quantization.py:49-56: two's complement nibbles, even column -> low nibble."""
    q = np.asarray(q)
    assert np.issubdtype(q.dtype, np.signedinteger)
    assert q.min(initial=0) >= -8 and q.max(initial=0) <= 7
    u = (q.astype(np.int16) & 0xF).astype(np.uint8)
    return (u[..., 0::2] | (u[..., 1::2] << 4)).astype(np.uint8)


def unpack_i4(p: np.ndarray) -> np.ndarray:
    """deploy/functional/quantization.py:60-82 -> int32, interleaved low/high."""
    p = np.asarray(p, dtype=np.uint8)
    lo = (p & 0x0F).astype(np.int32)
    hi = ((p & 0xF0) >> 4).astype(np.int32)
    lo[lo >= 8] -= 16
    hi[hi >= 8] -= 16
    out = np.empty(p.shape[:-1] + (p.shape[-1] * 2,), dtype=np.int32)
    out[..., 0::2] = lo
    out[..., 1::2] = hi
    return out


# --------------------------------------------------------------------------------------------------
# fp16-operand / fp32-accumulate GEMM with a pinned accumulation order
# --------------------------------------------------------------------------------------------------
def _grouped_matmul(a16: np.ndarray, b16: np.ndarray, groups: Optional[Sequence[Sequence[int]]]):
    """a16 [..., I, K] @ b16 [K, J] -> fp32 [..., I, J].

    Products of fp16 values are exact in fp32 (22-bit significands) and a sum of <= a few hundred of them is
    exact in float64 for inputs of moderate dynamic range, so "sum the group exactly" is a float64 sum.
    """
    a = a16.astype(np.float64)
    b = b16.astype(np.float64)
    K = a.shape[-1]
    if groups is None:
        groups = [list(range(K))]
    acc = np.zeros(a.shape[:-1] + (b.shape[-1],), dtype=F32)
    for g in groups:
        g = list(g)
        part = a[..., g] @ b[g, :]                      # exact group sum in float64
        acc = (acc.astype(np.float64) + part).astype(F32)   # one fp32 rounding per group
    return acc


def kron_transform(x16, left16, right16, diag16=None, groups1=None, groups2=None, left_first=False, lowp="f16"):
    """Y = left^T . fp16(X . right) for every token; returns fp32 [T, M, N].

    left_first=True evaluates the deploy Triton kernel's association instead (kron_matmul.py:63-71):
    T = fp16(left^T . X), Y = T . right.  The two orders differ by the fp16 rounding of a different
    intermediate (about 5e-4 of INT4 indices flip by one step on LLM-like data).

    flatquant/flat_utils.py:13-16: ``x = x.reshape(-1, L, R); x = x @ hadR; x = hadL.T @ x`` with fp16
    tensors (each torch.matmul accumulates in fp32 and returns fp16).  The second rounding to fp16 is the
    caller's choice here (FQ_ROUND_Y_F16 / round_y_f16) because the deploy kernels quantise the fp32
    accumulator (kron_matmul.py:71-107).
    """
    x16 = _as_lowp(x16, lowp)          # (fp32 arrays of fp16- / bf16-representable values from here on)
    left16 = _as_lowp(left16, lowp)
    right16 = _as_lowp(right16, lowp)
    M, N = left16.shape[0], right16.shape[0]
    X = x16.reshape(-1, M, N)
    if diag16 is not None:
        d = _as_lowp(diag16, lowp).reshape(M, N)
        X = _rnd((X * d).astype(F32), lowp)      # one fp16 / bf16 multiply (the product of two such values is exact in fp32
                                                 # for bf16 and rounded once to fp32 for fp16: a multiply in fp32 opmath)
    if left_first:
        # T[t, m', n] = sum_m L[m, m'] X[t, m, n]   ->  (X^T)[t, n, m] @ L
        Tt = _rnd(_grouped_matmul(np.swapaxes(X, -1, -2), left16, groups1), lowp)   # [T, N, M']
        return _grouped_matmul(np.swapaxes(Tt, -1, -2), right16, groups2)           # [T, M', N']
    U = _rnd(_grouped_matmul(X, right16, groups1), lowp)       # [T, M, N'] rounded to the activation dtype
    # Y[t, m', n'] = sum_m L[m, m'] U[t, m, n']  ->  (U^T)[t, n', m] @ L[m, m']
    Yt = _grouped_matmul(np.swapaxes(U, -1, -2), left16, groups2)   # [T, N', M']
    return np.ascontiguousarray(np.swapaxes(Yt, -1, -2))               # [T, M', N']


def single_transform(x16, P16, groups=None, lowp="f16"):
    """x [T, R, C] fp16 @ P [C, C] -> fp32 [T, R, C]  (block_matmul.py:59-70; trans_utils.py:21-25)."""
    return _grouped_matmul(_as_lowp(x16, lowp), _as_lowp(P16, lowp), groups)


# --------------------------------------------------------------------------------------------------
# per-token symmetric INT4 quantisation
# --------------------------------------------------------------------------------------------------
def token_scale(y32, sig_max=1.0, sig_min=1.0, clamp0=True, quant_f16=False, sig_f16=False, lowp="f16", bits=4):
    """y32 [T, d] fp32 -> scale fp32 [T].

    quant_utils.py:88-107 (clamp to 0, lac sigmoid factors, m = max(|xmin|, xmax), scale = m/q_max,
    scale[m == 0] = 1) evaluated in fp32 — what torch's type promotion yields when ``lac`` multiplies the
    fp16 row extrema by the fp32 sigmoid — and kron_matmul.py:91-104 (no clamp: clamp0=False).
    quant_f16: the scale is rounded to fp16 ((m/7).to(float16), deploy/nn/quantization.py:25-30).
    bits: q_max = 2^(bits-1) - 1 (get_qmin_qmax, quant_utils.py:10-16; every reference script uses 4).
    """
    y32 = np.asarray(y32, dtype=F32)
    xmax = y32.max(axis=-1)
    xmin = y32.min(axis=-1)
    if clamp0:
        xmax = np.maximum(xmax, F32(0))
        xmin = np.minimum(xmin, F32(0))
    if sig_f16:
        # deploy/nn/quantization.py:21-22: fp16 [rows, 1] extrema times a 0-dim fp32 sigmoid tensor -> torch's result is
        # fp16: the product is formed in fp32 (the sigmoid keeps its fp32 value) and rounded to fp16 (checked against the
        # reference module: tests/golden/quantizer_lac.npz)
        xmax = _rnd((xmax * F32(sig_max)).astype(F32), lowp)
        xmin = _rnd((xmin * F32(sig_min)).astype(F32), lowp)
    else:
        xmax = (xmax * F32(sig_max)).astype(F32)
        xmin = (xmin * F32(sig_min)).astype(F32)
    m = np.maximum(np.abs(xmin), xmax).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = (m / F32(2 ** (bits - 1) - 1)).astype(F32)
    if quant_f16:
        scale = _rnd(scale, lowp)
    scale = np.where(m == 0, F32(1.0), scale).astype(F32)
    return scale


def quantize(y32, scale, quant_f16=False, lowp="f16", bits=4):
    """clamp(rint(y/scale), -8, 7) -> int8 (bits != 4: [-2^(bits-1), 2^(bits-1) - 1] -> int16); division fp32 (fp16-rounded
    quotient when quant_f16)."""
    y32 = np.asarray(y32, dtype=F32)
    s = np.asarray(scale, dtype=F32)[..., None]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        t = (y32 / s).astype(F32)
        if quant_f16:
            t = _rnd(t, lowp)
    t = np.rint(t)
    t = np.clip(t, -(2 ** (bits - 1)), 2 ** (bits - 1) - 1)
    t = np.where(np.isnan(t), 0, t)
    return t.astype(np.int8 if bits <= 8 else np.int16)


def dequantize(q, scale, quant_f16=False, lowp="f16"):
    """(scale * q).to(fp16)  (quant_utils.py:25-26,81). lowp="bf16": the bf16 result as an fp32 array."""
    s = np.asarray(scale, dtype=F32)[..., None]
    if quant_f16:
        s = _rnd(s, lowp)
    p = (s * q.astype(F32)).astype(F32)
    return bf16_round(p) if lowp == "bf16" else p.astype(F16)


def quant_outputs(y32, sig_max=1.0, sig_min=1.0, round_y_f16=False, clamp0=True, quant_f16=False, groupsize=-1,
                  sig_f16=False, lowp="f16", bits=4):
    """Everything the fused kernels can emit for one clip set, from the fp32 transformed activation.

    groupsize > 0: ``ActivationQuantizer(groupsize=g)`` of vllm_custom/model_executor/layers/quantization/utils/
    fake_quant_utils.py:72-78 — ``x.reshape(-1, groupsize)`` before the extrema, i.e. one scale per `groupsize`
    consecutive elements of a token; "scale" is then [T, d / groupsize] (deepseek_v3/kernel.py:10-30 uses the same
    128-element blocks)."""
    y32 = np.asarray(y32, dtype=F32)
    T = y32.shape[0]
    y = y32.reshape(T, -1)
    d = y.shape[1]
    # "y16" / "fq" / "scale16": fp16 arrays, or (lowp="bf16") fp32 arrays of bf16-representable values
    y16 = bf16_round(y) if lowp == "bf16" else y.astype(F16)
    if round_y_f16:
        y = y16.astype(F32)
    if groupsize > 0:
        assert d % groupsize == 0
        yg = y.reshape(T * (d // groupsize), groupsize)
        scale = token_scale(yg, sig_max, sig_min, clamp0, quant_f16, sig_f16, lowp, bits)
        q = quantize(yg, scale, quant_f16, lowp, bits).reshape(T, d)
        fq = dequantize(q.reshape(yg.shape), scale, quant_f16, lowp).reshape(T, d)
        scale = scale.reshape(T, d // groupsize)
    else:
        scale = token_scale(y, sig_max, sig_min, clamp0, quant_f16, sig_f16, lowp, bits)
        q = quantize(y, scale, quant_f16, lowp, bits)
        fq = dequantize(q, scale, quant_f16, lowp)
    return {
        "y16": y16,
        "scale": scale,
        "scale16": bf16_round(scale) if lowp == "bf16" else scale.astype(F16),
        "q": q,
        "packed": pack_i4(q) if bits == 4 else None,   # (the packed format is the INT4 one)
        "fq": fq,
    }


def kron_quant(x16, left16, right16, sig_max=1.0, sig_min=1.0, diag16=None, round_y_f16=False,
               clamp0=True, quant_f16=False, groups1=None, groups2=None, left_first=False, groupsize=-1, sig_f16=False,
               lowp="f16"):
    y = kron_transform(x16, left16, right16, diag16, groups1, groups2, left_first, lowp)
    return quant_outputs(y, sig_max, sig_min, round_y_f16, clamp0, quant_f16, groupsize, sig_f16, lowp)


def kron_quant_grouped(x16, left16, right16, group_offsets, sig_max_g, sig_min_g, round_y_f16=False, clamp0=True,
                       groupsize=-1, quant_f16=False, sig_f16=False, lowp="f16"):
    """The routed experts of flatquant/model_tools/deepseekv3_utils.py:427-452: rows sorted by expert, expert i owns
    rows [group_offsets[i], group_offsets[i+1]) (``idx, top = torch.where(indices == i)``; ``expert(x[idx], ...)``),
    every expert transforms with its matrices and quantises with its own clip pair. left16 / right16: one shared pair
    ([M, M], [N, N]: ``routed_w2_trans`` at :470, independent_w2_trans = False) or one pair per expert ([G, M, M],
    [G, N, N]: the ``routed_w2_trans[i]`` branch, :446). Returns the same dict as kron_quant over all rows."""
    lp = F32 if lowp == "bf16" else F16
    x16 = np.asarray(x16, dtype=lp)
    left16, right16 = np.asarray(left16, dtype=lp), np.asarray(right16, dtype=lp)
    offs = [int(v) for v in group_offsets]
    parts = []
    for g in range(len(offs) - 1):
        a, b = offs[g], offs[g + 1]
        if b == a:
            continue
        L = left16[g] if left16.ndim == 3 else left16
        R = right16[g] if right16.ndim == 3 else right16
        parts.append(kron_quant(x16[a:b], L, R, float(sig_max_g[g]), float(sig_min_g[g]), None, round_y_f16, clamp0,
                                quant_f16, groupsize=groupsize, sig_f16=sig_f16, lowp=lowp))
    if not parts:                                              # no rows at all: the empty result, shapes intact
        d = left16.shape[-1] * right16.shape[-1]
        ns = d // groupsize if groupsize > 0 else None
        return {"y16": np.zeros((0, d), lp), "scale": np.zeros((0, ns) if ns else (0,), F32),
                "scale16": np.zeros((0, ns) if ns else (0,), lp), "q": np.zeros((0, d), np.int8),
                "packed": np.zeros((0, d // 2), np.uint8), "fq": np.zeros((0, d), lp)}
    return {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0]}


def rowquant(x16, sig_max=1.0, sig_min=1.0, clamp0=True, quant_f16=False, sig_f16=False, lowp="f16", bits=4):
    x = _as_lowp(x16, lowp)
    return quant_outputs(x.reshape(x.shape[0], -1), sig_max, sig_min, False, clamp0, quant_f16,
                         sig_f16=sig_f16, lowp=lowp, bits=bits)


def rowquant_asym(x16, sig_max=1.0, sig_min=1.0, quant_f16=False, lowp="f16", bits=4):
    """ActivationQuantizer(sym=False).fake_quant on an fp16 activation -> fp16 [rows, cols].

    quant_utils.py:86-92 (row extrema through 0), :95-100 (clip factors), :109-113 (both zero -> (-1, +1);
    scale = (xmax - xmin) / q_max, q_max = 15; zero = round(-xmin / scale)), :33-46 (q = clamp(round_ste(x / scale) + zero,
    0, q_max); scale * (q - zero)), :81 (.to(x_dtype)). round_ste's (r - t) + t equals r exactly in both widths.
    quant_f16=False: lac with fp32 (1,)-shaped clip parameters — torch promotes the extrema, scale, zero, quotient and
    product to fp32. quant_f16=True: no lac / clip_ratio / half()'ed module — every operation rounds to fp16 (the
    extremum x factor product is formed in fp32 opmath and then rounded to fp16).
    """
    x = _as_lowp(x16, lowp)
    x = x.reshape(x.shape[0], -1)
    rnd = (lambda a: _rnd(a.astype(F32), lowp)) if quant_f16 else (lambda a: a.astype(F32))
    xmax = np.maximum(x.max(axis=-1), F32(0))
    xmin = np.minimum(x.min(axis=-1), F32(0))
    xmax = rnd((xmax * F32(sig_max)).astype(F32))
    xmin = rnd((xmin * F32(sig_min)).astype(F32))
    both = (xmax == 0) & (xmin == 0)
    xmin = np.where(both, F32(-1), xmin).astype(F32)
    xmax = np.where(both, F32(1), xmax).astype(F32)
    qmax = 2 ** bits - 1   # (get_qmin_qmax, quant_utils.py:14; sums of two integers <= 2 qmax are exact in fp16, and inside [0, qmax] in bf16)
    scale = rnd(rnd(xmax - xmin) / F32(qmax))
    zero = np.rint(rnd((-xmin) / scale)).astype(F32)
    t = rnd(x / scale[:, None])
    q = np.clip(rnd(np.rint(t) + zero[:, None]), 0, qmax).astype(F32)
    out = (scale[:, None] * (q - zero[:, None])).astype(F32)
    return bf16_round(out) if lowp == "bf16" else out.astype(F16)


def block_quant(x16, P16, sig_max=1.0, sig_min=1.0, transpose_out=True, round_y_f16=False, clamp0=False,
                quant_f16=False, groups=None, lowp="f16"):
    """block_matmul.py:29-104: Y = x[t] ([R, C]) @ P; statistics over the whole block; the quantised block
    is transposed before packing (:86-90) when transpose_out."""
    y = single_transform(x16, P16, groups, lowp)         # [T, R, C]
    if transpose_out:
        y = np.ascontiguousarray(np.swapaxes(y, -1, -2))  # [T, C, R]
    return quant_outputs(y, sig_max, sig_min, round_y_f16, clamp0, quant_f16, lowp=lowp)


# --------------------------------------------------------------------------------------------------
# quant.cu restatements
# --------------------------------------------------------------------------------------------------
def sym_quant(x16, scale16):
    """quant.cu:13-47: q = clamp(__half2int_rn(__hdiv(x, scale[row])), -8, 7); odd tail -> high nibble 0."""
    x16 = np.asarray(x16, dtype=F16)
    s = np.asarray(scale16, dtype=F16).reshape(-1, 1)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        d = (x16.astype(F32) / s.astype(F32)).astype(F16).astype(F32)
    t = np.clip(np.rint(d), -8, 7)
    t = np.where(np.isnan(t), 0, t).astype(np.int8)
    if t.shape[1] & 1:
        t = np.concatenate([t, np.zeros((t.shape[0], 1), np.int8)], axis=1)
    return pack_i4(t)


def quantizer_plain(x16, input_clip_ratio=1.0):
    """deploy.nn.Quantizer(input_clip_ratio, lac=False).forward on an fp16 [rows, cols] activation (deploy/nn/quantization.py:30-33):
    scales = (max|x| / 7).to(fp16) * ratio — the division and the product each rounded to fp16 (an fp16 tensor times a python
    scalar: the product formed in fp32 opmath, what the CPU evaluates) — with NO zero guard (an all-zero row keeps scale 0), then
    deploy.sym_quant (quant.cu:13-47: 0 / 0 -> NaN -> digit 0). -> (packed uint8 [rows, cols/2], scales fp16 [rows])."""
    x = np.asarray(x16, dtype=F16)
    m = np.abs(x.astype(F32)).max(axis=-1)
    s = (m / F32(7)).astype(F16)
    s = (s.astype(F32) * F32(input_clip_ratio)).astype(F16)
    return sym_quant(x, s), s


def sym_dequant(q32, scale_row16, scale_col16):
    """quant.cu:5-10,66-85: x = s_row * s_col * half(int(q/10.0f)) * half(10), fp16 products left to right."""
    q32 = np.asarray(q32, dtype=np.int32)
    iv = np.trunc(q32.astype(F32) / F32(10.0)).astype(np.int64)
    iv = np.clip(iv, -65176, 65176)
    with np.errstate(over="ignore"):
        xe = iv.astype(F32).astype(F16)
        sr = np.asarray(scale_row16, dtype=F16).reshape(-1, 1)
        sc = np.asarray(scale_col16, dtype=F16).reshape(1, -1)
        r = (sr.astype(F32) * sc.astype(F32)).astype(F16)
        r = (r.astype(F32) * xe.astype(F32)).astype(F16)
        r = (r.astype(F32) * F32(10.0)).astype(F16)
    return r


def kv_asym_quant(x16, clip_max16=1.0, clip_min16=1.0, lac=False):
    """deploy/transformers/kv_cache.py:11-51 (asym_quantize_and_pack_i4) on fp16 tensors, every torch op rounding to
    fp16 as it does there. Per last-axis row: -> (packed uint8 [..., n/2] low nibble = even column, scale fp16 [..., 1],
    zero fp16 [..., 1], q uint8 [..., n]). The cache's own call sites leave lac=False (kv_cache.py:283-284)."""
    x = np.asarray(x16, dtype=F16)
    xmax = x.max(axis=-1, keepdims=True)
    xmin = x.min(axis=-1, keepdims=True)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        if lac:
            xmax = np.maximum(xmax, F16(0))
            xmin = np.minimum(xmin, F16(0))
            xmax = (xmax.astype(F32) * F32(F16(clip_max16))).astype(F16)
            xmin = (xmin.astype(F32) * F32(F16(clip_min16))).astype(F16)
            both = (xmin == 0) & (xmax == 0)
            xmin = np.where(both, F16(-1), xmin)
            xmax = np.where(both, F16(1), xmax)
            d = (xmax.astype(F32) - xmin.astype(F32)).astype(F16)
            scale = (d.astype(F32) / F32(15)).astype(F16)
            nx = (F32(-1.0) * xmin.astype(F32)).astype(F16)
            zero = np.rint((nx.astype(F32) / scale.astype(F32)).astype(F16)).astype(F16)
            t = np.rint((x.astype(F32) / scale.astype(F32)).astype(F16)).astype(F16)
            q = np.clip((t.astype(F32) + zero.astype(F32)).astype(F16), F16(0), F16(15))
        else:
            d = (xmax.astype(F32) - xmin.astype(F32)).astype(F16)
            d = np.maximum(d, F16(1e-5))
            scale = (d.astype(F32) / F32(15)).astype(F16)
            zero = (-xmin).astype(F16)
            t = (x.astype(F32) + zero.astype(F32)).astype(F16)
            q = np.clip(np.rint((t.astype(F32) / scale.astype(F32)).astype(F16)), F16(0), F16(15))
    q8 = np.nan_to_num(q.astype(F32), nan=0.0).astype(np.uint8)
    packed = (q8[..., 0::2] | (q8[..., 1::2] << 4)).astype(np.uint8)
    return packed, scale, zero, q8


def kv_asym_dequant(packed, scale16, zero16, lac=False):
    """kv_cache.py:54-61 (unpack_i4_and_asym_dequantize): lac: scale * (q - zero); else q * scale - zero (fp16 ops)."""
    p = np.asarray(packed, dtype=np.uint8)
    q = np.stack((p & 0x0F, (p >> 4) & 0x0F), axis=-1).reshape(*p.shape[:-1], p.shape[-1] * 2).astype(F16)
    s, z = np.asarray(scale16, dtype=F16), np.asarray(zero16, dtype=F16)
    with np.errstate(over="ignore", invalid="ignore"):
        if lac:
            return (s.astype(F32) * (q.astype(F32) - z.astype(F32)).astype(F16).astype(F32)).astype(F16)
        return ((q.astype(F32) * s.astype(F32)).astype(F16).astype(F32) - z.astype(F32)).astype(F16)


def kv_transform(x16, trans16):
    """kv_cache.py:268 torch.matmul(key_states.to(fp16), trans_matrix_k): fp32 accumulation, fp16 result."""
    return (np.asarray(x16, dtype=F16).astype(F32) @ np.asarray(trans16, dtype=F16).astype(F32)).astype(F16)


# The paged-cache functions below restate CUDA kernels of the reference (vendored FlashInfer) that cannot be run in this
# container and have no test vectors in the reference: PARITY UNPINNED for these two (formulas cited by file:line).
def kv_cache_append(kv_data, kv_param, indptr, indices, last_page_offset, layer_idx, k, v, k_param, v_param,
                    seqlen_indptr=None):
    """page.cuh:118-214 in numpy (in place): kv_data [pages, layers, 2, heads, page, hd/2] uint8, kv_param [..., page, 2]
    fp16; k, v [tokens, heads, hd/2], params [tokens, heads, 2]. Lengths are the ones AFTER the append."""
    batch = len(last_page_offset)
    page_size = kv_data.shape[4]
    for b in range(batch):
        seq_len = (indptr[b + 1] - indptr[b] - 1) * page_size + last_page_offset[b]
        if seqlen_indptr is None:
            toks = [b]
        else:
            toks = list(range(seqlen_indptr[b], seqlen_indptr[b + 1]))
        start = seq_len - len(toks)
        for j, t in enumerate(toks):
            pos = start + j
            page, entry = indices[indptr[b] + pos // page_size], pos % page_size
            kv_data[page, layer_idx, 0, :, entry, :] = k[t]
            kv_data[page, layer_idx, 1, :, entry, :] = v[t]
            kv_param[page, layer_idx, 0, :, entry, :] = k_param[t]
            kv_param[page, layer_idx, 1, :, entry, :] = v_param[t]


def kv_cache_decode(q16, kv_data, kv_param, indptr, indices, last_page_offset, layer_idx):
    """decode.cuh:492-683 (rotary none) as a plain fp64 softmax attention over the de-quantised rows
    (quantization.cuh:58-80: n * scale - zero with fp32(fp16) parameters): q [batch, heads, hd] -> o fp16."""
    q = np.asarray(q16, dtype=F16).astype(np.float64)
    batch, heads, hd = q.shape
    page_size = kv_data.shape[4]
    out = np.zeros((batch, heads, hd), dtype=np.float64)
    for b in range(batch):
        seq_len = (indptr[b + 1] - indptr[b] - 1) * page_size + last_page_offset[b]
        rows = [(indices[indptr[b] + pos // page_size], pos % page_size) for pos in range(seq_len)]
        for h in range(heads):
            kd, vd = [], []
            for page, entry in rows:
                for which, dst in ((0, kd), (1, vd)):
                    p8 = kv_data[page, layer_idx, which, h, entry]
                    n = np.stack((p8 & 15, p8 >> 4), axis=-1).reshape(-1).astype(np.float64)
                    s, z = kv_param[page, layer_idx, which, h, entry].astype(np.float64)
                    dst.append(n * s - z)
            K, V = np.array(kd), np.array(vd)
            x = K @ q[b, h] / np.sqrt(hd)
            w = np.exp(x - x.max())
            out[b, h] = (w / w.sum()) @ V
    return out.astype(F16)


def silu_mul(gate16, up16):
    """deploy/transformers/modeling_llama.py:277-278 on fp16 tensors: ac = act_fn(x_gate) (SiLU: fp32 g / (1 + exp(-g)),
    rounded to fp16), x = x_up * ac (fp16 product = exact product rounded once)."""
    g = np.asarray(gate16, dtype=F16).astype(F32)
    with np.errstate(over="ignore", invalid="ignore"):
        ac = (g / (F32(1.0) + np.exp(-g, dtype=F32))).astype(F16)
    return (np.asarray(up16, dtype=F16).astype(F32) * ac.astype(F32)).astype(F16)


def rmsnorm(x16, eps=1e-5):
    """deploy/nn/normalization.py:16-23: fp32(x) * rsqrt(sum(x^2) / d + eps) -> fp16 (no weight). The fp32 sum uses
    numpy's pairwise order; any kernel's order differs in the last bits of the variance (tests allow for that)."""
    x = np.asarray(x16, dtype=F16).astype(F32)
    var = (x * x).sum(axis=-1, keepdims=True, dtype=F32) / F32(x.shape[-1])
    r = (F32(1.0) / np.sqrt(var + F32(eps))).astype(F32)
    return (x * r).astype(F16)


def int4_matmul(x_packed, w_packed):
    """deploy/kernels/gemm.cu:8-47 (CUTLASS int4b_t row-major x column-major -> int32): c[m][n] = sum_k x[m][k] w[n][k]
    on the nibbles of pack_i4's layout (even k in the low nibble, two's complement). Exact integer arithmetic."""
    x = unpack_i4(np.asarray(x_packed, dtype=np.uint8)).astype(np.int64)
    w = unpack_i4(np.asarray(w_packed, dtype=np.uint8)).astype(np.int64)
    c = x @ w.T
    assert np.abs(c).max(initial=0) < 2 ** 31
    return c.astype(np.int32)


def linear4bit(x_packed, x_scale16, w_packed, w_scale16, bias16=None):
    """deploy/nn/linear.py:41-56: sym_dequant(matmul(x, w), scales_x, weight_scales) (+ bias, an fp16 add)."""
    y = sym_dequant(int4_matmul(x_packed, w_packed), x_scale16, w_scale16)
    if bias16 is not None:
        with np.errstate(over="ignore"):
            y = (y.astype(F32) + np.asarray(bias16, dtype=F16).reshape(1, -1).astype(F32)).astype(F16)
    return y


# --------------------------------------------------------------------------------------------------
# Hadamard
# --------------------------------------------------------------------------------------------------
def fwht_f32(v32):
    """Unnormalised FWHT over the last axis (power of two), fp32, stages from stride 1 upwards — the order
    of hadamard_utils.py:94-101 (adjacent pairs first) and of the fast_hadamard_transform kernel
    (in-thread, then warp shuffles, then across warps)."""
    v = np.array(v32, dtype=F32, copy=True)
    n = v.shape[-1]
    assert n & (n - 1) == 0
    h = 1
    lead = v.shape[:-1]
    while h < n:
        w = v.reshape(lead + (n // (2 * h), 2, h))
        a = w[..., 0, :].copy()
        b = w[..., 1, :].copy()
        w[..., 0, :] = a + b
        w[..., 1, :] = a - b
        v = w.reshape(lead + (n,))
        h *= 2
    return v


def hadamard(x16, K=1, hadK16=None, scale=None, groupsK=None):
    """matmul_hadU_cuda (hadamard_utils.py:132-141): view [rows, K, n/K]; FWHT over the last axis in fp32
    times `scale`, rounded to fp16 (fast_hadamard_transform returns the input dtype); then
    hadK [K,K] @ . with fp32 accumulation, rounded to fp16."""
    x16 = np.asarray(x16, dtype=F16)
    rows, n = x16.shape
    if scale is None:
        scale = F32(1.0) / np.sqrt(F32(n))
    v = fwht_f32(x16.astype(F32).reshape(rows, K, n // K))
    v = (v * F32(scale)).astype(F16)
    if K == 1:
        return v.reshape(rows, n)
    h = np.asarray(hadK16, dtype=F16)
    # out[r, k', p] = sum_k hadK[k', k] v[r, k, p]   ->  v^T [r, p, k] @ hadK^T [k, k']
    o = _grouped_matmul(np.swapaxes(v, -1, -2), np.ascontiguousarray(h.T), groupsK)
    return np.swapaxes(o, -1, -2).astype(F16).reshape(rows, n)
