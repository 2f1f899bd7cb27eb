// fq_common.hpp — shared device helpers for the gfx950 kernels of libfqhip.
//
// Arithmetic pinned here (and restated on the CPU in oracle/):
//   * per-token statistics, scale = m/7 and x/scale are IEEE fp32 (round-to-nearest-even, correctly
//     rounded division), mirroring flatquant/quant_utils.py:85-107 when `lac` promotes to fp32 and
//     deploy/kernels/kron_matmul.py:91-107;
//   * rint is round-half-to-even (torch.round / llrint / __half2int_rn);
//   * nibble order: even element -> low nibble (deploy/functional/quantization.py:49-56).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fqhip.h"

typedef _Float16 f16;
typedef f16   f16x2  __attribute__((ext_vector_type(2)));
typedef f16   f16x4  __attribute__((ext_vector_type(4)));
typedef f16   f16x8  __attribute__((ext_vector_type(8)));
typedef float f32x4  __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// bfloat16 activations (round 3): the reference's path A is dtype-generic (flat_utils.py:6-17; quant_utils.py:86 casts q_max
// to x's dtype) and its pipeline feeds it bf16 on Llama-3 / Qwen / DeepSeek (model_utils.py:20 torch_dtype='auto',
// main_dpskv3.py:395 set_default_dtype(bfloat16)). Kernels that serve that surface take the element type as a template
// parameter T in {f16, bf16}: v_mfma_f32_32x32x16_bf16, v_cvt_pk_bf16_f32 (round to nearest even) for every rounding the
// torch expression performs, fp32 for everything the type promotion makes fp32. The f16 instantiations are unchanged code.
typedef __bf16 bf16;
typedef bf16  bf16x2 __attribute__((ext_vector_type(2)));
typedef bf16  bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16  bf16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct FqVec;
template <> struct FqVec<f16>  { typedef f16x8  x8; typedef f16x4  x4; typedef f16x2  x2; static constexpr bool is_f16 = true;  };
template <> struct FqVec<bf16> { typedef bf16x8 x8; typedef bf16x4 x4; typedef bf16x2 x2; static constexpr bool is_f16 = false; };

// internal launcher flag (never part of the ABI's `flags`): the tensors are bf16
constexpr int FQ_DT_BF16 = 0x20000000;

template <typename T>
__device__ __forceinline__ f32x16 fq_mfma32(typename FqVec<T>::x8 a, typename FqVec<T>::x8 b, f32x16 c);
template <>
__device__ __forceinline__ f32x16 fq_mfma32<f16>(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 fq_mfma32<bf16>(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// fp32 -> T -> fp32: the rounding a torch op with a T result performs on its fp32 opmath value
template <typename T>
__device__ __forceinline__ float fq_round_to(float v) { return (float)(T)v; }

// Per-launch output description, passed by value as a kernel argument.
struct FqQuantOut {
    float    sig_max[FQ_MAX_CLIPS];
    float    sig_min[FQ_MAX_CLIPS];
    uint8_t* q[FQ_MAX_CLIPS];      // packed INT4, or nullptr
    f16*     scale[FQ_MAX_CLIPS];  // fp16 scales, or nullptr
    f16*     fq[FQ_MAX_CLIPS];     // fake-quant fp16, or nullptr
    f16*     y;                    // transformed fp16, or nullptr
    int      n_clips;
    int      rt_flags;             // run-time flags: FQ_ROUND_Y_F16, FQ_NO_CLAMP0 (wave-uniform branches)
    float    rms_eps;              // FQ_IN_RMSNORM: epsilon of the fused RMSNorm
    const f16* in2;                // FQ_IN_SILU_MUL: the `up` tensor (x is `gate`), same shape as x
    // grouped launch (fq_kron_quant_grouped_f16): rows are sorted by group (expert), group g owns rows
    // [group_offsets[g], group_offsets[g+1]) and quantises with its own clip pair; nullptr = not grouped
    const int64_t* group_offsets;  // device, [n_groups + 1]
    const float*   sig_max_g;      // device, [n_groups]
    const float*   sig_min_g;      // device, [n_groups]
    int            n_groups;
    float          post_scale;     // != 0: the transformed activation is multiplied by it (fp32) before rounding / statistics
                                   // (fq_kron_quant_ex_f16: a normalisation that fp16 factor matrices cannot carry exactly)
    int64_t        ws_group_stride;  // != 0 (fq_kron_quant_grouped_mats_*): every group has its OWN factor pair; the workspace
                                     // holds n_groups fragment images this many 16-byte chunks apart (routed_w2_trans[i],
                                     // deepseekv3_utils.py:446)
};

// Clip pair of token `tok` (wave-uniform): the launch-wide pair `ci`, or the pair of the token's group. A wave walks its
// tokens in increasing order, so it keeps a cursor (g, [begin, end)) and only searches when the token leaves the range:
// a binary search over group_offsets with scalar loads (<= 10 dependent s_loads for 1024 groups), taking the LAST group
// whose offset is <= tok, i.e. skipping empty groups.
struct FqGroupCursor {
    int g = -1;
    int64_t begin = 0, end = 0;
    float smax = 1.0f, smin = 1.0f;
};
// (the cursor is wave-uniform, but what a global load returns lives in VGPRs: without the readfirstlane below the compiler
//  keeps the whole cursor — two 64-bit bounds and the clip pair — in vector registers and shuffles them around every token)
__device__ __forceinline__ int64_t fq_uniform_i64(int64_t v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float fq_uniform_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
// Move the cursor to the group of token `tok` (wave-uniform; a no-op while the token stays inside the cursor's range).
__device__ __forceinline__ void fq_group_locate(const FqQuantOut& out, int64_t tok, FqGroupCursor& cur);

__device__ __forceinline__ void fq_token_sigs(const FqQuantOut& out, int ci, int64_t tok, FqGroupCursor& cur, float& smax,
                                              float& smin) {
    if (out.group_offsets == nullptr) {
        smax = out.sig_max[ci];
        smin = out.sig_min[ci];
        return;
    }
    fq_group_locate(out, tok, cur);
    smax = cur.smax;
    smin = cur.smin;
}

__device__ __forceinline__ void fq_group_locate(const FqQuantOut& out, int64_t tok, FqGroupCursor& cur) {
    if (cur.g < 0 || tok >= cur.end || tok < cur.begin) {
        int lo = 0, hi = out.n_groups;  // invariant: offsets[lo] <= tok < offsets[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (fq_uniform_i64(out.group_offsets[mid]) <= tok) lo = mid;
            else hi = mid;
        }
        cur.g = lo;
        cur.begin = fq_uniform_i64(out.group_offsets[lo]);
        cur.end = fq_uniform_i64(out.group_offsets[lo + 1]);
        cur.smax = out.sig_max_g ? fq_uniform_f32(out.sig_max_g[lo]) : 1.0f;   // (a transform-only launch carries no clip pairs)
        cur.smin = out.sig_min_g ? fq_uniform_f32(out.sig_min_g[lo]) : 1.0f;
    }
}

// Raise a kernel's dynamic-LDS cap once PER DEVICE (hipFuncSetAttribute acts on the current device; one process may
// drive several GPUs). Used inside the int-returning launchers: a failure is returned as the hipError_t.
#define FQ_RAISE_LDS_CAP(kern, bytes)                                                                               \
    do {                                                                                                            \
        static unsigned long long fq_done_ = 0; /* bit per device; a racy double-set stores the same attribute */   \
        int fq_dev_ = 0;                                                                                            \
        (void)hipGetDevice(&fq_dev_);                                                                               \
        if (!((fq_done_ >> (fq_dev_ & 63)) & 1ull)) {                                                               \
            const hipError_t fq_e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                       \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            if (fq_e_ != hipSuccess) return (int)fq_e_;                                                             \
            fq_done_ |= 1ull << (fq_dev_ & 63);                                                                     \
        }                                                                                                           \
    } while (0)

constexpr int FQ_NO_WAVE_KERNEL = 0x40000000;  // internal (launcher to launcher): skip the wave-per-token kernels

// Flags that select a compile-time kernel specialisation; the rest travel in FqQuantOut::rt_flags.
constexpr int FQ_CT_MASK = FQ_OUT_PACKED | FQ_OUT_FAKEQUANT | FQ_OUT_TRANSFORM | FQ_QUANT_F16 | FQ_IN_RMSNORM;

// Wave64 all-reduce in registers: four DPP steps inside each 16-lane row, then v_permlane16_swap and
// v_permlane32_swap (gfx950) across rows. ~6 VALU-latency steps instead of six ds_bpermute round trips
// through the LDS crossbar. Every lane ends up with the result.
template <int CTRL>
__device__ __forceinline__ float fq_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// One instruction per reduction step: v_max/min_f32 with a DPP source. (Through fmaxf + update_dpp hipcc emits a
// v_mov_dpp, a canonicalising v_max x,x and the v_max: 3 VALU per step.) "s_nop 1": a VALU write needs 2 wait
// states before a DPP read of the same register.
#define FQ_DPP_STEP(OPNAME, CTRLSTR)                                                                     \
    asm("s_nop 1\n\t" OPNAME " %0, %1, %1 " CTRLSTR " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v)); \
    v = r;
struct FqMaxOp {
    __device__ __forceinline__ float operator()(float a, float b) const {
        float d;
        asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
        return d;
    }
    __device__ __forceinline__ float row(float v) const {
        float r;
        FQ_DPP_STEP("v_max_f32_dpp", "quad_perm:[1,0,3,2]")
        FQ_DPP_STEP("v_max_f32_dpp", "quad_perm:[2,3,0,1]")
        FQ_DPP_STEP("v_max_f32_dpp", "row_half_mirror")
        FQ_DPP_STEP("v_max_f32_dpp", "row_mirror")
        return v;
    }
};
struct FqMinOp {
    __device__ __forceinline__ float operator()(float a, float b) const {
        float d;
        asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
        return d;
    }
    __device__ __forceinline__ float row(float v) const {
        float r;
        FQ_DPP_STEP("v_min_f32_dpp", "quad_perm:[1,0,3,2]")
        FQ_DPP_STEP("v_min_f32_dpp", "quad_perm:[2,3,0,1]")
        FQ_DPP_STEP("v_min_f32_dpp", "row_half_mirror")
        FQ_DPP_STEP("v_min_f32_dpp", "row_mirror")
        return v;
    }
};
template <class Op>
__device__ __forceinline__ float fq_wave_reduce(float v, Op op) {
    v = op.row(v);  // every lane of a 16-lane row now holds the row's result
    // v_permlane16_swap: odd rows of the first register <-> even rows of the second; v_permlane32_swap: upper
    // half of the first <-> lower half of the second. Written as inline asm on two explicit registers: with
    // the builtin hipcc (ROCm 7.2) folds the two results into one when both inputs hold the same value.
    // "s_nop 1" = the 2 wait states a VALU write needs before v_permlane*_swap reads it.
    {
        float a = v, b = v;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        v = op(a, b);
    }
    {
        float a = v, b = v;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        v = op(a, b);
    }
    return v;
}
struct FqAddOp {
    __device__ __forceinline__ float operator()(float a, float b) const { return a + b; }
    __device__ __forceinline__ float row(float v) const {
        float r;
        FQ_DPP_STEP("v_add_f32_dpp", "quad_perm:[1,0,3,2]")
        FQ_DPP_STEP("v_add_f32_dpp", "quad_perm:[2,3,0,1]")
        FQ_DPP_STEP("v_add_f32_dpp", "row_half_mirror")
        FQ_DPP_STEP("v_add_f32_dpp", "row_mirror")
        return v;
    }
};
__device__ __forceinline__ float fq_wave_sum(float v) { return fq_wave_reduce(v, FqAddOp()); }  // every lane: the total
__device__ __forceinline__ float fq_wave_max(float v) { return fq_wave_reduce(v, FqMaxOp()); }
__device__ __forceinline__ float fq_wave_min(float v) { return fq_wave_reduce(v, FqMinOp()); }

// fp16(fp32(a * b)): the fp32 product is ROUNDED TO fp32 FIRST, then to fp16 — what torch does for
// (scale * q).to(float16) (quant_utils.py:25-26,81). hipcc otherwise selects v_fma_mixlo_f16 for
// fptrunc(fmul), which rounds the exact product once and differs when the fp32 product lands on an fp16 tie
// (seen on the GPU: 7 * 3.184152 -> 22.297 instead of 22.281). The empty asm makes the product opaque.
template <typename T>
__device__ __forceinline__ T fq_mul_to(float a, float b) {
    float p = a * b;
    asm volatile("" : "+v"(p));
    return (T)p;
}
__device__ __forceinline__ f16 fq_mul_to_f16(float a, float b) { return fq_mul_to<f16>(a, b); }
// The same for two values at once: v_pk_mul_f32 (both fp32 products, each rounded to fp32) + v_cvt_pk_f16_f32 — one VALU per element
// where two fq_mul_to_f16 cost two (the fp16 epilogues of the Hadamard-as-Kronecker launches: round 4). Needs f32x2 (defined below).
typedef float fq_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f16x2 fq_mul_to_f16x2(float a0, float a1, fq_f32x2_t b2) {
    fq_f32x2_t p = fq_f32x2_t{a0, a1} * b2;
    asm volatile("" : "+v"(p));
    return f16x2{(f16)p.x, (f16)p.y};
}

// FQ_GROUP128 with N = 64: a 128-element group is two consecutive 64-element rows of the transformed token, held by the
// lanes (h, c) with c in {2j, 2j+1}, h in {0, 1} of one output-row tile: combine a per-lane partial extremum over
// lane ^ 1 (DPP quad_perm) and lane ^ 32 (v_permlane32_swap). Every one of the four lanes ends with the group's value.
template <class Op>
__device__ __forceinline__ float fq_group4_reduce(float v, Op op) {
    {
        const float r = fq_dpp<0xB1>(v);  // quad_perm:[1,0,3,2]: lane ^ 1
        v = op(v, r);
    }
    {
        float a = v, b = v;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        v = op(a, b);
    }
    return v;
}

// scale from (xmax, xmin) of one token and one clip set; see header comment for the pinned arithmetic.
template <int FLAGS, typename T = f16>
__device__ __forceinline__ float fq_token_scale(float xmax, float xmin, float sig_max, float sig_min,
                                                int rt_flags) {
    if (!(rt_flags & FQ_NO_CLAMP0)) {
        xmax = fmaxf(xmax, 0.0f);
        xmin = fminf(xmin, 0.0f);
    }
    if ((FLAGS & FQ_QUANT_F16) && (rt_flags & FQ_RATIO_POST)) {
        // deploy Quantizer(input_clip_ratio) (deploy/nn/quantization.py:30): (max|x| / 7).to(fp16) * ratio, both steps in T
        const float m0 = fmaxf(fabsf(xmin), xmax);
        if (m0 == 0.0f) return 1.0f;   // (quantise with 1: every digit is 0; the caller stores the reference's scale, 0)
        return (float)fq_mul_to<T>((float)(T)(m0 / 7.0f), sig_max);
    }
    if ((FLAGS & FQ_QUANT_F16) && (rt_flags & FQ_SIG_F16)) {
        // deploy.nn.Quantizer(lac=True), deploy/nn/quantization.py:21-22: an fp16 [rows, 1] tensor times a 0-dim fp32
        // sigmoid tensor is an fp16 RESULT under torch's type promotion: the product is formed in fp32 (fp16 extremum x the
        // fp32 sigmoid, rounded to fp32) and then rounded to fp16 — two roundings (run on the reference: golden
        // quantizer_lac.npz). The (1,)-shaped parameters of quant_utils.py:96-97 promote the result to fp32 instead.
        // (T = bf16: a bf16 extremum times a bf16 sigmoid — main_dpskv3.py:395 makes the clip parameters bf16 — is exact in
        //  fp32 and rounds to bf16 once; the caller passes the bf16-rounded sigmoid)
        xmax = (float)fq_mul_to<T>(xmax, sig_max);
        xmin = (float)fq_mul_to<T>(xmin, sig_min);
    } else {
        xmax = xmax * sig_max;
        xmin = xmin * sig_min;
    }
    float m = fmaxf(fabsf(xmin), xmax);
    float scale;
    if (FLAGS & FQ_QUANT_F16) {
        // fp16 scale: (m/7).to(fp16) as deploy/nn/quantization.py:25-30 and quant_utils.py:103 (fp16
        // tensors; with sig == 1 m is itself an fp16 value, so fp16(m/7) is the fp16 division).
        // (T = bf16: the fp32 quotient of two 8-bit significands rounded to bf16 is the correctly rounded bf16 quotient)
        scale = (float)(T)(m / 7.0f);
        if (m == 0.0f) scale = 1.0f;
    } else {
        scale = m / 7.0f;
        if (m == 0.0f) scale = 1.0f;
    }
    return scale;
}

// clamp(rint(y/scale), -8, 7) as an integer in [-8, 7].
template <int FLAGS, typename T = f16>
__device__ __forceinline__ int fq_quant1(float y, float scale) {
    float t = y / scale;                       // correctly rounded fp32 division
    if (FLAGS & FQ_QUANT_F16) t = (float)(T)t;  // fp16 / bf16 quotient (== __hdiv, no double-rounding issue)
    t = __builtin_rintf(t);
    t = fminf(fmaxf(t, -8.0f), 7.0f);
    return (int)t;
}

// The fp16-arithmetic quantiser when y itself is an fp16 value (rowquant / sym_quant inputs; quant.cu:40 __hdiv):
// the native _Float16 division (v_rcp_f32 + two v_fma_mix refinements + v_div_fixup_f16, ~7 VALU with the reciprocal
// of the per-row scale hoisted) instead of the ~11 of an IEEE fp32 division. tools/microbench/h16div.hip checked on
// gfx950 that it equals the correctly rounded quotient for ALL 2^32 pairs of finite fp16 values.
__device__ __forceinline__ int fq_quant1_h(f16 y, f16 s) {
    const f16 t = y / s;
    float r = __builtin_rintf((float)t);
    r = __builtin_amdgcn_fmed3f(r, -8.0f, 7.0f);
    return (int)r;
}
// bf16: torch evaluates x / scale in fp32 opmath and rounds the quotient to bf16 (a quotient of two 8-bit significands is
// never within 2^-24 of a bf16 rounding boundary unless it lies on it: the double rounding is harmless)
__device__ __forceinline__ int fq_quant1_h(bf16 y, bf16 s) {
    const float t = (float)(bf16)((float)y / (float)s);
    float r = __builtin_rintf(t);
    r = __builtin_amdgcn_fmed3f(r, -8.0f, 7.0f);
    return (int)r;
}

// x_up * act_fn(x_gate) in front of a down_proj transform (deploy/transformers/modeling_llama.py:277-278, fp16 tensors):
// ac = fp16( g / (1 + exp(-g)) ) evaluated in fp32 (torch's SiLU opmath), x = fp16(ac * up). v_exp_f32 / v_rcp_f32
// are 1-ulp fp32 approximations: after the fp16 rounding a few results in 10^4 differ from a correctly rounded fp32
// evaluation by one fp16 step (tests/test_gpu_silu.py states the bound).
__device__ __forceinline__ f16x8 fq_silu8(f16x8 g) {
    f16x8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float gf = (float)g[j];
        const float e = __builtin_amdgcn_exp2f(gf * -1.44269504088896340736f);
        a[j] = (f16)(gf * __builtin_amdgcn_rcpf(1.0f + e));
    }
    return a;
}
__device__ __forceinline__ f16x8 fq_silu_mul8(f16x8 g, f16x8 u) { return fq_silu8(g) * u; }

// The fake-quant value fp16(fp32(scale * q)) of an integer-valued float q. A zero product is made +0.0: the reference
// rounds with round_ste (quant_utils.py:3-7: (x.round() - x) + x), which never returns -0.0, while v_rndne_f32 of a small
// negative quotient does (the reference-written fixtures hold no negative zero).
template <typename T>
__device__ __forceinline__ T fq_fake(float scale, float q) {
    float p = scale * q;
    asm volatile("" : "+v"(p));
    p = p + 0.0f;
    return (T)p;
}
__device__ __forceinline__ f16 fq_fake_f16(float scale, float q) { return fq_fake<f16>(scale, q); }

template <int FLAGS, typename T = f16>
__device__ __forceinline__ T fq_dequant1(int q, float scale) {
    // FQ_QUANT_F16: scale is an fp16 (bf16) value and |q| <= 8, so the fp32 product is exact and one rounding remains
    if (FLAGS & FQ_QUANT_F16) return (T)((float)(T)scale * (float)q);
    return fq_mul_to<T>(scale, (float)q);
}

// Extrema of eight 16-bit floats held as four dwords (a 16-byte chunk of an activation row). fp16: packed v_pk_max/min_f16
// on running pairs (see fq_pk_max below); bf16 has no packed max on gfx950 — a bf16 IS the upper half of an fp32, so the two
// halves of a dword become fp32 values with one shift / one and, and the extrema run on v_max3 / v_min3_f32.
__device__ __forceinline__ void fq_bf16_pair(uint32_t w, float& lo, float& hi) {
    lo = __builtin_bit_cast(float, w << 16);
    hi = __builtin_bit_cast(float, w & 0xFFFF0000u);
}

// ---------------------------------------------------------------------------------------------------
// Fast EXACT quantiser (fp32 arithmetic only).
//
// The pinned result is q = clamp(rint(fl(y / s)), -8, 7) with a correctly rounded division, which costs
// ~11 VALU per element. Instead: inv = fl(1/s) once per token, t = fl(y * inv), r = rint(t).
// |t - fl(y/s)| <= 3 * 2^-24 |y/s| (rounding of inv, of the product and of the true quotient), i.e. less
// than 2.9e-6 for |y/s| <= 16; beyond 16 both sides clamp identically. Hence whenever t is further than
// FQ_NEAR from every half-integer, rint(t) == rint(fl(y/s)). Each lane tracks dmax = max |t - rint(t)|;
// if any lane of the wave saw dmax > 0.5 - FQ_NEAR (probability ~3e-2 per 4096-element token) the
// caller recomputes that token with the true division. Results are therefore bit-identical to the slow
// form, always.
// ---------------------------------------------------------------------------------------------------
constexpr float FQ_NEAR = 4e-6f;

// 1/scale for the fast quantisers: v_rcp_f32 (1 ulp) instead of the IEEE division (~12 VALU on every lane for a
// wave-uniform value). The proofs below only need |inv - 1/s| <= 2^-23 / s: with |t| <= 16 the fast quotient then
// differs from fl(y/s) by < 2.9e-6 + 1e-6 < FQ_NEAR. (The scale itself, an output, stays a correctly rounded m / 7.)
#ifndef FQ_IEEE_INV
#define FQ_IEEE_INV 0  // 1 (A/B builds): the correctly rounded reciprocal
#endif
// Issued as asm with one wait state behind it: the consumers are often the first instruction of an inline-asm quantiser
// block, where the compiler's hazard recogniser does not see the read — gfx950 needs one wait state between a
// transcendental op and a VALU read of its result (without it the first element of a block occasionally used a stale
// reciprocal: fq_kv_quant_kernel, 10 of 131072 rows).
__device__ __forceinline__ float fq_fast_inv(float scale) {
    if (FQ_IEEE_INV) return 1.0f / scale;
    float r;
    asm("v_rcp_f32_e32 %0, %1\n\ts_nop 0" : "=v"(r) : "v"(scale));
    return r;
}


// Single-instruction 3-input max/min. fmaxf(fmaxf(a,b),c) compiles to v_max3_f32 only after hipcc has inserted
// a canonicalising v_max_f32 x,x per MFMA-produced operand (one extra VALU per element); NaNs are not part of the
// contract here, so use the instruction directly.
__device__ __forceinline__ float fq_max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float fq_min3(float a, float b, float c) {
    float d;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float fq_max3_abs(float a, float b, float c) {  // max(a, |b|, |c|)
    float d;
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Packed fp16 max / min as the bare instruction. __builtin_elementwise_max on data that came straight from memory makes
// hipcc canonicalise every operand first (a v_pk_max_f16 x, x per loaded dword: +50 % on the extrema of the row
// quantisers); NaNs are not part of the contract here.
#ifdef FQ_PK_BUILTIN  // measurement builds: the builtin (with its canonicalising copies)
__device__ __forceinline__ f16x2 fq_pk_max(f16x2 a, f16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ f16x2 fq_pk_min(f16x2 a, f16x2 b) { return __builtin_elementwise_min(a, b); }
#else
__device__ __forceinline__ f16x2 fq_pk_max(f16x2 a, f16x2 b) {
    f16x2 d;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f16x2 fq_pk_min(f16x2 a, f16x2 b) {
    f16x2 d;
    asm("v_pk_min_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
#endif
// three-operand forms (gfx950: v_pk_maximum3_f16 / v_pk_minimum3_f16 — IEEE maximum / minimum: as the two-operand max / min on data without NaNs)
__device__ __forceinline__ f16x2 fq_pk_max3(f16x2 a, f16x2 b, f16x2 c) {
    f16x2 d;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f16x2 fq_pk_min3(f16x2 a, f16x2 b, f16x2 c) {
    f16x2 d;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// Running extrema of 16-byte chunks of an activation row (row quantisers; the group-128 epilogue of the Kronecker kernels).
template <typename T> struct RowExtrema;
template <> struct RowExtrema<f16> {
    f16x2 pmax = {(f16)-INFINITY, (f16)-INFINITY}, pmin = {(f16)INFINITY, (f16)INFINITY};
    __device__ __forceinline__ void take(const f16x8& v) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f16x2 pr = {v[e], v[e + 1]};
            pmax = fq_pk_max(pmax, pr);
            pmin = fq_pk_min(pmin, pr);
        }
    }
    __device__ __forceinline__ float vmax() const { return fmaxf((float)pmax[0], (float)pmax[1]); }
    __device__ __forceinline__ float vmin() const { return fminf((float)pmin[0], (float)pmin[1]); }
};
template <> struct RowExtrema<bf16> {
    float mx = -INFINITY, mn = INFINITY;
    __device__ __forceinline__ void take(const bf16x8& v) {
        const u32x4 w = __builtin_bit_cast(u32x4, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float lo, hi;
            fq_bf16_pair(w[k], lo, hi);
            mx = fq_max3(mx, lo, hi);
            mn = fq_min3(mn, lo, hi);
        }
    }
    __device__ __forceinline__ float vmax() const { return mx; }
    __device__ __forceinline__ float vmin() const { return mn; }
};


// Two elements at once: v_pk_mul_f32 / v_pk_add_f32 process a register pair per instruction on gfx950.
// Returns the clamped integer-valued pair; dmax accumulates max |t - rint(t)| (see fq_qfast).
__device__ __forceinline__ f32x2 fq_qfast2(f32x2 y, f32x2 inv2, float& dmax) {
    const f32x2 t = y * inv2;
    f32x2 r;
    r.x = __builtin_rintf(t.x);
    r.y = __builtin_rintf(t.y);
    const f32x2 d = t - r;
    dmax = fq_max3_abs(dmax, d.x, d.y);
    r.x = __builtin_amdgcn_fmed3f(r.x, -8.0f, 7.0f);
    r.y = __builtin_amdgcn_fmed3f(r.y, -8.0f, 7.0f);
    return r;
}

// Cheaper form of the same test (2 VALU per element instead of 2.5, 3 with the clamp): round with the
// magic-number add instead of v_rndne. u = fma(y, inv, 1.5*2^23) is the EXACT product y*inv rounded to an integer
// (the ulp of u is 1 for |y*inv| < 2^22), r = u - magic is that integer, e = fma(y, inv, -r) the exact residual
// rounded once. |e| <= 0.5 - FQ_NEAR proves rint(fl(y/s)) == r by the same argument as above (y*inv differs from
// y/s, and fl(y/s) from y/s, by <= |t| 2^-24 each). Callers guarantee |y*inv| < 2^21 (fq_magic_ok) and, for
// CLAMP == false, that every quotient of the token rounds into [-8, 7] (fq_needs_clamp).
constexpr float FQ_MAGIC = 12582912.0f;
template <bool CLAMP>
__device__ __forceinline__ f32x2 fq_qmagic2(f32x2 y, f32x2 inv2, float& dmax) {
#ifndef FQ_SCALAR_QUANT
#define FQ_SCALAR_QUANT 0  // 1: single-lane-width v_fma/v_add instead of v_pk_*_f32 (build with -fno-slp-vectorize)
#endif
#if FQ_SCALAR_QUANT
    f32x2 r;
    float ex, ey;
    {
        const float ux = __builtin_fmaf(y.x, inv2.x, FQ_MAGIC), uy = __builtin_fmaf(y.y, inv2.y, FQ_MAGIC);
        r.x = ux - FQ_MAGIC;
        r.y = uy - FQ_MAGIC;
        ex = __builtin_fmaf(y.x, inv2.x, -r.x);
        ey = __builtin_fmaf(y.y, inv2.y, -r.y);
    }
    dmax = fq_max3_abs(dmax, ex, ey);
#else
    const f32x2 magic = {FQ_MAGIC, FQ_MAGIC};
    const f32x2 u = __builtin_elementwise_fma(y, inv2, magic);
    f32x2 r = u - magic;
    const f32x2 e = __builtin_elementwise_fma(y, inv2, -r);
    dmax = fq_max3_abs(dmax, e.x, e.y);
#endif
    if (CLAMP) {
        r.x = __builtin_amdgcn_fmed3f(r.x, -8.0f, 7.0f);
        r.y = __builtin_amdgcn_fmed3f(r.y, -8.0f, 7.0f);
    }
    return r;
}
// token-uniform guards for fq_qmagic2 (vmax / vmin: the token's raw extremes, inv = 1/scale > 0)
__device__ __forceinline__ bool fq_magic_ok(float vmax, float vmin, float inv) {
    return fmaxf(vmax, -vmin) * inv < 2097152.0f;
}
__device__ __forceinline__ bool fq_needs_clamp(float vmax, float vmin, float inv) {
    return !(vmax * inv < 7.49f && vmin * inv > -8.49f);
}

// Eight integer-valued floats in [-8, 7], given as the pairs p_j = (r_2j, r_2j+1), -> one dword of
// two's-complement nibbles. Two packed fmas build (r0 + 256 r2, r1 + 256 r3) and the same for r4..r7, one fma
// each interleaves them into a signed 16-bit digit string, and adding 1.5*2^23 + 0x8888 leaves the offset-binary
// 16-bit value in the low mantissa bits (everything is an integer < 2^24, so every step is exact).
__device__ __forceinline__ uint32_t fq_pack8p(f32x2 p0, f32x2 p1, f32x2 p2, f32x2 p3) {
#if FQ_SCALAR_QUANT
    const f32x2 lo = {__builtin_fmaf(p1.x, 256.0f, p0.x), __builtin_fmaf(p1.y, 256.0f, p0.y)};
    const f32x2 hi = {__builtin_fmaf(p3.x, 256.0f, p2.x), __builtin_fmaf(p3.y, 256.0f, p2.y)};
#else
    const f32x2 c256 = {256.0f, 256.0f};
    const f32x2 lo = __builtin_elementwise_fma(p1, c256, p0);
    const f32x2 hi = __builtin_elementwise_fma(p3, c256, p2);
#endif
    // (kept as two scalars: with both halves in one f32x2, hipcc 7.2 folds the v_perm_b32 below onto a single source)
    const float wl = __builtin_fmaf(lo.y, 16.0f, lo.x) + (FQ_MAGIC + 34952.0f);
    const float wh = __builtin_fmaf(hi.y, 16.0f, hi.x) + (FQ_MAGIC + 34952.0f);
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, wh), __builtin_bit_cast(uint32_t, wl), 0x05040100u) ^
           0x88888888u;
}

// The quantiser of the packed output: one asm block per dword (8 elements), single-width VALU only.
// v_pk_fma_f32 / v_pk_add_f32 (what hipcc's SLP vectoriser makes of fq_qmagic2 / fq_pack8p in fq_common.hpp) slow a
// SIMD down next to another wave's MFMAs (tools/microbench/phase_overlap.hip: MFMA phase + pk phase take MORE than their
// sum, single-width VALU hides a third of the MFMA time) and each one drags a hazard s_nop along; the C++ form of
// single-width arithmetic (-fno-slp-vectorize) is scheduled into 72 spilled VGPRs at fq_kron64_kernel's 128-register cap.
// Here the six temporaries are all there is.
//
// The exactness proof, two-sided. The pinned result is q = rint(fl(y / s)) with a correctly rounded division.
// With ilo = inv (1 - 2^-21) and ihi = inv (1 + 2^-21)
// (inv = v_rcp_f32(s), 1 ulp), both the true quotient y/s and its fp32 rounding fl(y/s) lie between the exact products
// y*ilo and y*ihi (relative slack 2^-21 against 2^-23 + 2^-24 + 2^-24 of rcp, the rounding of ilo/ihi and of the
// quotient). u = fma(y, ilo, MAGIC) and v = fma(y, ihi, MAGIC) are those products rounded to integers, once, half to
// even, and rint is monotone: u == v  =>  rint(fl(y/s)) == u. The integer r sits in the low mantissa bits of u
// (bits(u) = 0x4B400000 + r, two's complement), so the eight digits are combined by integer Horner steps
// v_lshl_add_u32 on the raw bits of u and of v; the MAGIC exponent bits fall out of the dword except for the
// constant K = 0x3F400000, (acc - K + 0x88888888) ^ 0x88888888 is the two's-complement nibble string (offset
// binary r + 8 per digit, then the XOR), and ONE v_cmp_ne of the two dwords says whether any of the eight digits
// was ambiguous (then the caller redoes that dword with the true division). CLAMP: see below.
template <bool CLAMP>
__device__ __forceinline__ uint32_t fq_quant8_two(float y0, float y1, float y2, float y3, float y4, float y5, float y6,
                                               float y7, float ilo, float ihi, unsigned long long& differ) {
    uint32_t a, b;
    float t0, s0, t1, s1;
    const float magic = FQ_MAGIC;
    // CLAMP: the bounds go onto y, ONE v_med3_f32 per element in front of the two products (on u and v it took two).
    // yhi = 7 / ilo and ylo = -8 / ilo (v_rcp_f32, 1 ulp) lie within 2^-20 (relative) of 7 s and -8 s: both products of
    // a bound round to 7 / -8 with no ambiguity, the quotient is monotone in y, so every y beyond a bound gets the digit
    // of the bound, which is what the clamp of the pinned result gives it; every y inside is untouched.
    float ylo = 0.0f, yhi = 0.0f;
    if (CLAMP) {
        const float s = __builtin_amdgcn_rcpf(ilo);
        ylo = -8.0f * s;
        yhi = 7.0f * s;
    }
#define FQ_UV(t, s, y) "v_fma_f32 %[" #t "], %[" #y "], %[ilo], %[mg]\n\tv_fma_f32 %[" #s "], %[" #y "], %[ihi], %[mg]\n\t"
#define FQ_CV(t, s, y) "v_med3_f32 %[" #t "], %[" #y "], %[ylo], %[yhi]\n\tv_fma_f32 %[" #s "], %[" #t "], %[ihi], %[mg]\n\tv_fma_f32 %[" #t "], %[" #t "], %[ilo], %[mg]\n\t"
#define FQ_HN(t, s) "v_lshl_add_u32 %[a], %[a], 4, %[" #t "]\n\tv_lshl_add_u32 %[b], %[b], 4, %[" #s "]\n\t"
#define FQ_TAIL "v_cmp_ne_u32_e64 %[m], %[a], %[b]\n\tv_add_u32_e32 %[a], 0x49488888, %[a]\n\tv_xor_b32_e32 %[a], 0x88888888, %[a]"
#define FQ_OUTS [a] "=&v"(a), [b] "=&v"(b), [t0] "=&v"(t0), [s0] "=&v"(s0), [t1] "=&v"(t1), [s1] "=&v"(s1), [m] "=&s"(differ)
#define FQ_INS [y0] "v"(y0), [y1] "v"(y1), [y2] "v"(y2), [y3] "v"(y3), [y4] "v"(y4), [y5] "v"(y5), [y6] "v"(y6), \
               [y7] "v"(y7), [ilo] "v"(ilo), [ihi] "v"(ihi), [mg] "s"(magic)
    if (CLAMP)
        asm(FQ_CV(a, b, y7) FQ_CV(t0, s0, y6) FQ_CV(t1, s1, y5) FQ_HN(t0, s0) FQ_CV(t0, s0, y4) FQ_HN(t1, s1)
            FQ_CV(t1, s1, y3) FQ_HN(t0, s0) FQ_CV(t0, s0, y2) FQ_HN(t1, s1) FQ_CV(t1, s1, y1) FQ_HN(t0, s0)
            FQ_CV(t0, s0, y0) FQ_HN(t1, s1) FQ_HN(t0, s0) FQ_TAIL
            : FQ_OUTS
            : FQ_INS, [ylo] "v"(ylo), [yhi] "v"(yhi));
    else
        asm(FQ_UV(a, b, y7) FQ_UV(t0, s0, y6) FQ_UV(t1, s1, y5) FQ_HN(t0, s0) FQ_UV(t0, s0, y4) FQ_HN(t1, s1)
            FQ_UV(t1, s1, y3) FQ_HN(t0, s0) FQ_UV(t0, s0, y2) FQ_HN(t1, s1) FQ_UV(t1, s1, y1) FQ_HN(t0, s0)
            FQ_UV(t0, s0, y0) FQ_HN(t1, s1) FQ_HN(t0, s0) FQ_TAIL
            : FQ_OUTS
            : FQ_INS);
#undef FQ_UV
#undef FQ_CV
#undef FQ_HN
#undef FQ_TAIL
#undef FQ_OUTS
#undef FQ_INS
    return a;
}

// ---------------------------------------------------------------------------------------------------
// Round 4: the packed quantiser with the FRACTION IN THE LOW HALF — 23 VALU per 8 elements (19 with v_pk_fma_f32) against the 33
// of fq_quant8_two; with the clamp 31 (27) against 41. tools/microbench/quant3.hip holds the experiment (exhaustive-style check
// against the true division, issue rate alone and next to MFMA phases).
//
//   u = fma(y, inv, C),  C = 200.5 + 2^-16 (0x43488001: 24 significant bits, exact).  For p = y inv in [-8.5, 7.5) the sum lies
//   in [192, 208) where ulp = 2^-16, so the ONE rounding of the fma leaves
//       bits(u) = 0x43400000 + floor-to-nearest((p + 8.5 + 2^-16) 2^16)
//   high half = 0x4340 + I,  I = floor(F) in [0, 15],  low half = frac(F) in units of 2^-16,  F = p + 8.5 + 2^-16 + e, |e| <= 2^-17.
//   Claim: low half >= 2  =>  rint(fl(y / s)) = I - 8.   With f = low half 2^-16 in [2^-15, 1 - 2^-16]:
//       p = (I - 8) - 0.5 + (f - 2^-16 - e),  f - 2^-16 - e in [2^-17, 1 - 1.5 2^-16]
//   so p is at least 2^-17 = 7.6e-6 away from both half-integers next to I - 8, while fl(y / s) differs from p by at most
//   |y/s| (2^-23 + 2^-24) (1 ulp of v_rcp_f32 in inv, the rounding of the quotient) < 3.1e-6 for |y/s| <= 17: fl(y / s) lies
//   strictly inside (I - 8.5, I - 7.5) and rounds to I - 8 — no tie can occur. A low half of 0 or 1 (probability 3e-5 per
//   element) flags the dword; the caller redoes it with the true division, as with fq_quant8_two.
//   Digits: I is the offset-binary nibble. v_mad_u32_u16 (op_sel: high half of src0) accumulates sum (0x4340 + I_k) 16^k over
//   four elements into 32 bits; the constant part 0x4340 * 0x1111 = 0x047BB740 of both halves leaves with the start value
//   -(0x047BB740 * 65537) = 0x444448C0, (b << 16) + a is the offset-binary string and ^ 0x88888888 the two's complement.
//   Test: v_min3_u16 over the low halves, one v_cmp per dword.
//   CLAMP: v_med3_i32 on the BITS of u (positive floats order like integers, a negative u is a negative integer) between
//   0x43408000 and 0x434F8000: p < -8.5 gives digit -8 and p >= 7.5 digit 7, which is what clamp(rint(.), -8, 7) gives them
//   (at p = -8.5 exactly rint may be -8 or -9: both clamp to -8), and the clamped low half 0x8000 is never flagged.
//   Callers guarantee what fq_quant8_two's callers do: without CLAMP every quotient of the token rounds into [-8, 7]
//   (fq_needs_clamp), and |y inv| < 2^21 (fq_magic_ok) so that u is finite and monotone in y.
// ---------------------------------------------------------------------------------------------------
#define FQ_LO_MAD(acc, u, k, src2) "v_mad_u32_u16 %[" #acc "], %[" #u "], " k ", " src2 " op_sel:[1,0,0,0]\n\t"
#define FQ_LO_CL(u) "v_med3_i32 %[" #u "], %[" #u "], %[blo], %[bhi]\n\t"
#define FQ_LO_NOCL(u)
// elements 0..3 of a dword: u0..u3 hold fma(y, inv, C); -> a (digits 0..3 + start value), tm (running minimum of the low halves)
#define FQ_LO_HALF_A(CL)                                                                                     \
    CL(u0) CL(u1) CL(u2) CL(u3)                                                                              \
    FQ_LO_MAD(a, u0, "1", "%[ini]") FQ_LO_MAD(a, u1, "16", "%[a]") "v_min3_u16 %[tm], %[u0], %[u1], %[u2]\n\t" \
    FQ_LO_MAD(a, u2, "%[k256]", "%[a]") FQ_LO_MAD(a, u3, "%[k4096]", "%[a]") "v_min_u16_e32 %[tm], %[tm], %[u3]\n\t"
// elements 4..7: -> the finished dword in a, the ambiguity mask in m
#define FQ_LO_HALF_B(CL)                                                                                     \
    CL(u0) CL(u1) CL(u2) CL(u3)                                                                              \
    FQ_LO_MAD(b, u0, "1", "0") FQ_LO_MAD(b, u1, "16", "%[b]") "v_min3_u16 %[tm], %[tm], %[u0], %[u1]\n\t"     \
    FQ_LO_MAD(b, u2, "%[k256]", "%[b]") FQ_LO_MAD(b, u3, "%[k4096]", "%[b]") "v_min3_u16 %[tm], %[tm], %[u2], %[u3]\n\t" \
    "v_cmp_gt_u16_e64 %[m], 2, %[tm]\n\t"                                                                    \
    "v_lshl_add_u32 %[a], %[b], 16, %[a]\n\t"                                                                \
    "v_xor_b32_e32 %[a], 0x88888888, %[a]"
#define FQ_LO_FMA4(ya, yb, yc, yd)                                                                           \
    "v_fma_f32 %[u0], %[" #ya "], %[inv], %[cc]\n\tv_fma_f32 %[u1], %[" #yb "], %[inv], %[cc]\n\t"             \
    "v_fma_f32 %[u2], %[" #yc "], %[inv], %[cc]\n\tv_fma_f32 %[u3], %[" #yd "], %[inv], %[cc]\n\t"
template <bool CLAMP, bool PK>
__device__ __forceinline__ uint32_t fq_quant8_lo(f32x2 y01, f32x2 y23, f32x2 y45, f32x2 y67, float inv, unsigned long long& amb) {
    uint32_t a, b, tm;
    const uint32_t ini = 0x444448C0u, k256 = 256u, k4096 = 4096u, blo = 0x43408000u;
    uint32_t bhi = 0x434F8000u;
    if (CLAMP) asm volatile("" : "+v"(bhi));   // (v_med3_i32 is VOP3: one scalar operand per instruction on gfx9, the other bound sits in a VGPR)
    if (PK) {
        // (y0 inv + C, y1 inv + C) in one v_pk_fma_f32: src1 / src2 broadcast their low dword (op_sel_hi 0)
        f32x2 u01, u23;
        const f32x2 inv2 = {inv, inv};
        const unsigned long long cc2 = 0x4348800143488001ull;
        asm("v_pk_fma_f32 %[u01], %[y01], %[inv2], %[cc2]\n\tv_pk_fma_f32 %[u23], %[y23], %[inv2], %[cc2]"
            : [u01] "=&v"(u01), [u23] "=&v"(u23) : [y01] "v"(y01), [y23] "v"(y23), [inv2] "v"(inv2), [cc2] "s"(cc2));
        {
            uint32_t u0 = __builtin_bit_cast(uint32_t, u01.x), u1 = __builtin_bit_cast(uint32_t, u01.y), u2 = __builtin_bit_cast(uint32_t, u23.x),
                     u3 = __builtin_bit_cast(uint32_t, u23.y);
            if (CLAMP)
                asm(FQ_LO_HALF_A(FQ_LO_CL) : [a] "=&v"(a), [tm] "=&v"(tm), [u0] "+v"(u0), [u1] "+v"(u1), [u2] "+v"(u2), [u3] "+v"(u3)
                    : [ini] "s"(ini), [k256] "s"(k256), [k4096] "s"(k4096), [blo] "s"(blo), [bhi] "v"(bhi));
            else
                asm(FQ_LO_HALF_A(FQ_LO_NOCL) : [a] "=&v"(a), [tm] "=&v"(tm) : [u0] "v"(u0), [u1] "v"(u1), [u2] "v"(u2), [u3] "v"(u3),
                    [ini] "s"(ini), [k256] "s"(k256), [k4096] "s"(k4096));
        }
        asm("v_pk_fma_f32 %[u01], %[y01], %[inv2], %[cc2]\n\tv_pk_fma_f32 %[u23], %[y23], %[inv2], %[cc2]"
            : [u01] "=&v"(u01), [u23] "=&v"(u23) : [y01] "v"(y45), [y23] "v"(y67), [inv2] "v"(inv2), [cc2] "s"(cc2));
        {
            uint32_t u0 = __builtin_bit_cast(uint32_t, u01.x), u1 = __builtin_bit_cast(uint32_t, u01.y), u2 = __builtin_bit_cast(uint32_t, u23.x),
                     u3 = __builtin_bit_cast(uint32_t, u23.y);
            if (CLAMP)
                asm(FQ_LO_HALF_B(FQ_LO_CL) : [a] "+v"(a), [b] "=&v"(b), [tm] "+v"(tm), [m] "=&s"(amb), [u0] "+v"(u0), [u1] "+v"(u1), [u2] "+v"(u2),
                    [u3] "+v"(u3) : [k256] "s"(k256), [k4096] "s"(k4096), [blo] "s"(blo), [bhi] "v"(bhi));
            else
                asm(FQ_LO_HALF_B(FQ_LO_NOCL) : [a] "+v"(a), [b] "=&v"(b), [tm] "+v"(tm), [m] "=&s"(amb) : [u0] "v"(u0), [u1] "v"(u1), [u2] "v"(u2),
                    [u3] "v"(u3), [k256] "s"(k256), [k4096] "s"(k4096));
        }
    } else {
        uint32_t u0, u1, u2, u3;
        const float y0 = y01.x, y1 = y01.y, y2 = y23.x, y3 = y23.y, y4 = y45.x, y5 = y45.y, y6 = y67.x, y7 = y67.y;
        const uint32_t cc = 0x43488001u;
#define FQ_LO_OUTS [a] "=&v"(a), [b] "=&v"(b), [tm] "=&v"(tm), [m] "=&s"(amb), [u0] "=&v"(u0), [u1] "=&v"(u1), [u2] "=&v"(u2), [u3] "=&v"(u3)
#define FQ_LO_INS [y0] "v"(y0), [y1] "v"(y1), [y2] "v"(y2), [y3] "v"(y3), [y4] "v"(y4), [y5] "v"(y5), [y6] "v"(y6), [y7] "v"(y7), \
                  [inv] "v"(inv), [cc] "s"(cc), [ini] "s"(ini), [k256] "s"(k256), [k4096] "s"(k4096)
        if (CLAMP)
            asm(FQ_LO_FMA4(y0, y1, y2, y3) FQ_LO_HALF_A(FQ_LO_CL) FQ_LO_FMA4(y4, y5, y6, y7) FQ_LO_HALF_B(FQ_LO_CL)
                : FQ_LO_OUTS : FQ_LO_INS, [blo] "s"(blo), [bhi] "v"(bhi));
        else
            asm(FQ_LO_FMA4(y0, y1, y2, y3) FQ_LO_HALF_A(FQ_LO_NOCL) FQ_LO_FMA4(y4, y5, y6, y7) FQ_LO_HALF_B(FQ_LO_NOCL)
                : FQ_LO_OUTS : FQ_LO_INS);
#undef FQ_LO_OUTS
#undef FQ_LO_INS
    }
    return a;
}
#undef FQ_LO_MAD
#undef FQ_LO_CL
#undef FQ_LO_NOCL
#undef FQ_LO_HALF_A
#undef FQ_LO_HALF_B
#undef FQ_LO_FMA4

// What the packed kernels call. FQ_QUANT_LO selects the formulation at build time: 0 fq_quant8_two (two-sided, round 3),
// 1 fq_quant8_lo with single-width fmas, 2 with v_pk_fma_f32. Bit-identical results by construction (both are exact or flagged).
#ifndef FQ_QUANT_LO
#define FQ_QUANT_LO 1
#endif
template <bool CLAMP>
__device__ __forceinline__ uint32_t fq_quant8(float y0, float y1, float y2, float y3, float y4, float y5, float y6, float y7, float inv,
                                              float ilo, float ihi, unsigned long long& differ) {
    if (FQ_QUANT_LO == 0) return fq_quant8_two<CLAMP>(y0, y1, y2, y3, y4, y5, y6, y7, ilo, ihi, differ);
    return fq_quant8_lo<CLAMP, FQ_QUANT_LO == 2>(f32x2{y0, y1}, f32x2{y2, y3}, f32x2{y4, y5}, f32x2{y6, y7}, inv, differ);
}

// The quantiser of the FAKE-QUANT output (FlatQuantizedLinear._eval_forward, flat_linear.py:75-80 -> quant_utils.py:77-83):
// eight transformed values -> eight fp16 (bf16) values scale * q, one asm block, single-width VALU only (round 3; the C++ form
// fq_qmagic2 is turned into v_pk_fma_f32 / v_pk_add_f32 by the SLP vectoriser: slow next to other waves' MFMAs, see above).
// One-sided exactness test, as fq_qmagic2: u = fma(y, inv, MAGIC) rounds the exact product to an integer, r = u - MAGIC
// (never -0.0: x - x = +0), e = fma(y, inv, -r) is the residual rounded once; |e| <= 0.5 - FQ_NEAR proves
// rint(fl(y / s)) == r (|y inv - y/s| and |fl(y/s) - y/s| <= |t| 2^-23.4 < FQ_NEAR for |t| <= 16; beyond, both sides clamp
// alike). dmax accumulates max |e| over the caller's whole token: ONE wave vote per token instead of one per 16-byte piece.
// p = r * scale is rounded to fp32 by v_mul_f32 and to T by the conversion: the two roundings of (scale * q).to(x_dtype).
// 41 VALU per 8 elements (49 with the clamp).
template <bool CLAMP, typename T>
__device__ __forceinline__ u32x4 fq_fake8(float y0, float y1, float y2, float y3, float y4, float y5, float y6, float y7,
                                          float inv, float scale, float& dmax) {
    uint32_t d0, d1, d2, d3;
    float u0, u1, e0, e1;
    const float magic = FQ_MAGIC;
    float lo = -8.0f, hi = 7.0f;
    if (CLAMP) asm volatile("" : "+v"(lo), "+v"(hi));  // (VGPR operands: neither is an inline constant, one SGPR per VALU op)
#define FQ_FK1(u, e, y)                                                 \
    "v_fma_f32 %[" #u "], %[" #y "], %[inv], %[mg]\n\t"                 \
    "v_subrev_f32_e32 %[" #u "], %[mg], %[" #u "]\n\t"                  \
    "v_fma_f32 %[" #e "], %[" #y "], %[inv], -%[" #u "]\n\t"
#define FQ_FKC(u) "v_med3_f32 %[" #u "], %[" #u "], %[lo], %[hi]\n\t"
#define FQ_FK_MAX "v_max3_f32 %[dm], %[dm], |%[e0]|, |%[e1]|\n\t"
#define FQ_FK_MUL(u) "v_mul_f32_e32 %[" #u "], %[sc], %[" #u "]\n\t"
#define FQ_FK_CVT_F16(d) "v_cvt_pk_f16_f32 %[" #d "], %[u0], %[u1]\n\t"
#define FQ_FK_CVT_BF16(d) "v_cvt_pk_bf16_f32 %[" #d "], %[u0], %[u1]\n\t"
#define FQ_FK_PAIR(d, ya, yb, CL, CVT) FQ_FK1(u0, e0, ya) FQ_FK1(u1, e1, yb) FQ_FK_MAX CL(u0) CL(u1) FQ_FK_MUL(u0) FQ_FK_MUL(u1) CVT(d)
#define FQ_FK_NOCL(u)
#define FQ_FK_OUTS [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3), [u0] "=&v"(u0), [u1] "=&v"(u1), \
                   [e0] "=&v"(e0), [e1] "=&v"(e1), [dm] "+v"(dmax)
#define FQ_FK_INS [y0] "v"(y0), [y1] "v"(y1), [y2] "v"(y2), [y3] "v"(y3), [y4] "v"(y4), [y5] "v"(y5), [y6] "v"(y6), \
                  [y7] "v"(y7), [inv] "v"(inv), [sc] "v"(scale), [mg] "s"(magic)
#define FQ_FK_BODY(CL, CVT) FQ_FK_PAIR(d0, y0, y1, CL, CVT) FQ_FK_PAIR(d1, y2, y3, CL, CVT) FQ_FK_PAIR(d2, y4, y5, CL, CVT) \
                            FQ_FK_PAIR(d3, y6, y7, CL, CVT)
    if (FqVec<T>::is_f16) {
        if (CLAMP) asm(FQ_FK_BODY(FQ_FKC, FQ_FK_CVT_F16) : FQ_FK_OUTS : FQ_FK_INS, [lo] "v"(lo), [hi] "v"(hi));
        else asm(FQ_FK_BODY(FQ_FK_NOCL, FQ_FK_CVT_F16) : FQ_FK_OUTS : FQ_FK_INS);
    } else {
        if (CLAMP) asm(FQ_FK_BODY(FQ_FKC, FQ_FK_CVT_BF16) : FQ_FK_OUTS : FQ_FK_INS, [lo] "v"(lo), [hi] "v"(hi));
        else asm(FQ_FK_BODY(FQ_FK_NOCL, FQ_FK_CVT_BF16) : FQ_FK_OUTS : FQ_FK_INS);
    }
#undef FQ_FK1
#undef FQ_FKC
#undef FQ_FK_MAX
#undef FQ_FK_MUL
#undef FQ_FK_CVT_F16
#undef FQ_FK_CVT_BF16
#undef FQ_FK_PAIR
#undef FQ_FK_NOCL
#undef FQ_FK_OUTS
#undef FQ_FK_INS
#undef FQ_FK_BODY
    return u32x4{d0, d1, d2, d3};
}

// The reciprocals fq_quant8_two takes, from inv = fq_fast_inv(scale): 1 -+ 2^-21.
__device__ __forceinline__ float fq_inv_lo(float inv) { return inv * 0.999999523162841796875f; }
__device__ __forceinline__ float fq_inv_hi(float inv) { return inv * 1.000000476837158203125f; }

// ---------------------------------------------------------------------------------------------------
// The fp16 (deploy Quantizer) contract on PACKED fp16 pairs, single-width VALU, no division: 8 elements -> one dword.
//   q = clamp(rint(RN16(x / s)), -8, 7)                     (quant.cu:40 __hdiv, __half2int_rn; x, s fp16)
// Exact fp16 quotient in TWO fp32 operations per element that read the fp16 halves directly (v_fma_mix_f32) — round 4; round 3
// used three (t = x r, e = fma(-t, s, x), t' = fma(e, r, t) with r = v_rcp_f32(s)):
//   once per row:  rhi = RN32(1 / s) (the IEEE division),  e = fma(-s, rhi, 1) (EXACT: s has 11 significant bits, the residual of
//                  a correctly rounded reciprocal at most 11),  rlo = RN32(e rhi)   =>   rhi + rlo = (1 / s)(1 + d), |d| < 2^-46
//   per element:   p = RN32(x rlo),  t = fma(x, rhi, p) = RN32(Q (1 + d')), Q = x / s, |d'| < 2^-45   (x rhi is exact inside the fma)
//   RN16(t) == RN16(Q): a quotient of two 11-bit significands is never within 2^-23 (relative) of an fp16 rounding boundary m (a
//   12-bit significand) unless it IS m — x - m s is a multiple of one unit in the last place of the 23-bit product m s — and t is
//   within 2^-24 + 2^-45 of Q: the same side of every boundary; and Q == m gives t == m exactly (m is an fp32 value), the same tie.
//   (Checked on the device against the native _Float16 division for every positive fp16 x and 3072 scales: tools/microbench/h16div2.hip.)
// Then v_cvt_pk_f16_f32, an optional packed clamp (rint and the clamp to [-8, 7] commute: the bounds are
// integers), v_pk_add_f16 with 1536.0 = 1.5 * 2^10 (the sum rounds to an integer, half to even, and 1536 is even: the
// low byte of each half is the two's-complement digit), and the eight low nibbles are gathered with v_perm_b32 / v_bfi_b32.
// 31 VALU per 8 elements (39 with the clamp) against 39 (47) of the round-3 form and ~100 of the per-element C++ form.
struct FqH16Recip {
    float hi, lo;
};
// (pure function of a row-uniform scale: call it once per row, outside the element loop)
__device__ __forceinline__ FqH16Recip fq_h16_recip(float s) {
    FqH16Recip r;
    r.hi = 1.0f / s;                                  // correctly rounded (the build keeps IEEE division)
    r.lo = __builtin_fmaf(-s, r.hi, 1.0f) * r.hi;
    return r;
}
template <bool CLAMP>
__device__ __forceinline__ uint32_t fq_quant8_h16(uint32_t xa, uint32_t xb, uint32_t xc, uint32_t xd, FqH16Recip rc) {
    uint32_t ha, hb, hc, hd;
    float t0, t1, l0, l1;
    const float rhi = rc.hi, rlo = rc.lo;
    const uint32_t magic2 = 0x66006600u;   // (1536.0h, 1536.0h)
    const uint32_t sel = 0x06040200u;      // bytes 0 and 2 of the second source, then of the first
#ifndef FQ_H16_MIXLO
#define FQ_H16_MIXLO 0   // measurement: the second fma and the conversion as ONE v_fma_mixlo_f16 / v_fma_mixhi_f16 (4 VALU per pair instead of 5)
#endif
#if FQ_H16_MIXLO
#define FQ_H16_PAIR(h, x)                                                                       \
    "v_fma_mix_f32 %[l0], %[" #x "], %[rlo], 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"            \
    "v_fma_mix_f32 %[l1], %[" #x "], %[rlo], 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"            \
    "v_fma_mixlo_f16 %[" #h "], %[" #x "], %[rhi], %[l0] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"  \
    "v_fma_mixhi_f16 %[" #h "], %[" #x "], %[rhi], %[l1] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
#else
#define FQ_H16_PAIR(h, x)                                                                       \
    "v_fma_mix_f32 %[l0], %[" #x "], %[rlo], 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"            \
    "v_fma_mix_f32 %[l1], %[" #x "], %[rlo], 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"            \
    "v_fma_mix_f32 %[t0], %[" #x "], %[rhi], %[l0] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"        \
    "v_fma_mix_f32 %[t1], %[" #x "], %[rhi], %[l1] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"        \
    "v_cvt_pk_f16_f32 %[" #h "], %[t0], %[t1]\n\t"
#endif
    asm(FQ_H16_PAIR(ha, xa) FQ_H16_PAIR(hb, xb) FQ_H16_PAIR(hc, xc) FQ_H16_PAIR(hd, xd)
        : [ha] "=&v"(ha), [hb] "=&v"(hb), [hc] "=&v"(hc), [hd] "=&v"(hd), [t0] "=&v"(t0), [t1] "=&v"(t1), [l0] "=&v"(l0),
          [l1] "=&v"(l1)
        : [xa] "v"(xa), [xb] "v"(xb), [xc] "v"(xc), [xd] "v"(xd), [rhi] "v"(rhi), [rlo] "v"(rlo));
#undef FQ_H16_PAIR
    if (CLAMP) {
        const uint32_t lo = 0xC800C800u;   // (-8.0h, -8.0h)
        uint32_t hi = 0x47004700u;         // ( 7.0h,  7.0h)
        asm volatile("" : "+v"(hi));
        asm("v_pk_max_f16 %[ha], %[ha], %[lo]\n\tv_pk_max_f16 %[hb], %[hb], %[lo]\n\t"
            "v_pk_max_f16 %[hc], %[hc], %[lo]\n\tv_pk_max_f16 %[hd], %[hd], %[lo]\n\t"
            "v_pk_min_f16 %[ha], %[ha], %[hi]\n\tv_pk_min_f16 %[hb], %[hb], %[hi]\n\t"
            "v_pk_min_f16 %[hc], %[hc], %[hi]\n\tv_pk_min_f16 %[hd], %[hd], %[hi]"
            : [ha] "+v"(ha), [hb] "+v"(hb), [hc] "+v"(hc), [hd] "+v"(hd)
            : [lo] "s"(lo), [hi] "v"(hi));
    }
    uint32_t d, p1, p2, u1, u2;
    asm("v_pk_add_f16 %[ha], %[ha], %[mg]\n\t"
        "v_pk_add_f16 %[hb], %[hb], %[mg]\n\t"
        "v_pk_add_f16 %[hc], %[hc], %[mg]\n\t"
        "v_pk_add_f16 %[hd], %[hd], %[mg]\n\t"
        "v_perm_b32 %[p1], %[hb], %[ha], %[sel]\n\t"        // low bytes of e0, e1, e2, e3
        "v_perm_b32 %[p2], %[hd], %[hc], %[sel]\n\t"        // low bytes of e4 .. e7
        "v_lshrrev_b32_e32 %[u1], 4, %[p1]\n\t"
        "v_lshrrev_b32_e32 %[u2], 4, %[p2]\n\t"
        "v_bfi_b32 %[p1], %[m4], %[u1], %[p1]\n\t"          // byte 0 = n0 | n1 << 4, byte 2 = n2 | n3 << 4
        "v_bfi_b32 %[p2], %[m4], %[u2], %[p2]\n\t"
        "v_perm_b32 %[d], %[p2], %[p1], %[sel]"
        : [d] "=v"(d), [p1] "=&v"(p1), [p2] "=&v"(p2), [u1] "=&v"(u1), [u2] "=&v"(u2), [ha] "+v"(ha), [hb] "+v"(hb),
          [hc] "+v"(hc), [hd] "+v"(hd)
        : [mg] "s"(magic2), [sel] "s"(sel), [m4] "v"(0x00F000F0u));
    return d;
}
// wave-uniform: must the packed clamp run? (all quotients of the row / token round into [-8, 7] otherwise)
__device__ __forceinline__ bool fq_h16_needs_clamp(float vmax, float vmin, float r) {
    return !(vmax * r < 7.49f && vmin * r > -8.49f);
}

__device__ __forceinline__ float fq_qfast(float y, float inv, float& dmax) {
    const float t = y * inv;
    const float r = __builtin_rintf(t);
    dmax = fmaxf(dmax, fabsf(t - r));
    return __builtin_amdgcn_fmed3f(r, -8.0f, 7.0f);
}
__device__ __forceinline__ float fq_qexact(float y, float scale) {
    return __builtin_amdgcn_fmed3f(__builtin_rintf(y / scale), -8.0f, 7.0f);
}
__device__ __forceinline__ bool fq_wave_needs_exact(float dmax) {
    return __any(dmax > 0.5f - FQ_NEAR) != 0;
}

// Eight integer-valued floats in [-8, 7] -> one dword of two's-complement nibbles, element 0 in bits 3:0.
// Horner in fp32 on offset-binary digits (all intermediates are integers < 2^24, so every fma is exact);
// XOR 0x8 per nibble turns offset-binary (r + 8) into two's complement.
__device__ __forceinline__ uint32_t fq_pack8(float r0, float r1, float r2, float r3, float r4, float r5,
                                             float r6, float r7) {
    const float lo = __builtin_fmaf(r3, 4096.0f, __builtin_fmaf(r2, 256.0f, __builtin_fmaf(r1, 16.0f, r0 + 34952.0f)));
    const float hi = __builtin_fmaf(r7, 4096.0f, __builtin_fmaf(r6, 256.0f, __builtin_fmaf(r5, 16.0f, r4 + 34952.0f)));
    return (((uint32_t)lo) | (((uint32_t)hi) << 16)) ^ 0x88888888u;
}
