// fq_kron_fast.hpp — the workgroup-per-token Kronecker kernel (fq_kron_fast_kernel) and its launcher, shared by the two
// translation units that instantiate it: fq_kron_generic.hip (fp16: every output set, compile-time output sets, SiLU.mul input)
// and fq_kron_generic2.hip (bf16, per-group matrices, the group-128 epilogue). One file held ~80 instantiations of a 600-line
// kernel and compiled for 3.3 minutes; split, the two halves build in parallel.
#pragma once
#include "fq_common.hpp"
#include <stdlib.h>

// Environment switches exist in measurement builds only (-DFQ_MEASURE, tools/microbench/*.sh): the product library reads
// no environment variables and keeps no mutable global state.
#ifdef FQ_MEASURE
static inline bool fq_measure_env(const char* name) { return getenv(name) != nullptr; }
#else
static constexpr bool fq_measure_env(const char*) { return false; }
#endif


namespace {


struct KronGeom {
    int M, N;        // factor sizes
    int KS1;         // K-steps of GEMM 1 = ceil(N / 16)
    int pitch;       // LDS row pitch of the staged token, in 16-byte chunks (odd)
};

// n' (physical output column) of GEMM-1 tile column c of tile nt; see fq_kron64.hip for the derivation.
__device__ __forceinline__ int ncol(int NT, int nt, int c) {
    return ((c >> 2) & 1) * (NT * 16) + nt * 16 + (c & 3) + 4 * (c >> 3);
}

// ---------------------------------------------------------------------------------------------------------------
// Compile-time K-steps (the factor pairs real models use). Per-token traffic kept off L2:
//   * a wave's R fragments never change -> loaded once into registers (TPW x KS1 x 4 VGPRs);
//   * the L fragments every wave needs in full live in LDS (2 MT^2 KB, copied once per workgroup);
//   * the next token is fetched into registers (coalesced 16-byte loads) while the current one is being
//     multiplied, and written to LDS after the barrier that frees the stage: HBM latency is off the critical path;
//   * the quantiser is the magic-number one of fq_common.hpp.
// Barriers per token: stage written | statistics exchanged (= stage free) | output stage complete.
// ---------------------------------------------------------------------------------------------------------------
// OCC = workgroups per CU the register allocation must leave room for (waves per SIMD = OCC * WAVES / 4).
// CTF >= 0: the OUTPUT-SET bits of `flags` are this compile-time constant (packed-only builds of the deploy shapes: the
// transform / fake-quant / fp16-quantiser branches, their register copies and exec-mask juggling drop out); the
// run-time bits (FQ_ROUND_Y_F16, FQ_NO_CLAMP0, measurement bits) still come from the argument.
// NV != 0: the TRUE row length N = NV with N % 16 != 0 (N % 4 == 0; 148 = Qwen2.5-7B's ffn pair 128 x 148), packed-only
// instantiations: KS1 / NT describe N padded to whole K-steps, the token is staged in 8-byte units (a 16-byte chunk would
// straddle two rows), the last 16-column run of a row is cut by N (extrema and stores take its valid part) and rows of the
// packed stage (N / 2 = 74 bytes) are written in 2-byte pieces.
// GM: the grouped launch with one factor pair per group (fq_kron_quant_grouped_mats_*): its own instantiations, so that the
// cursor and the image reload do not cost the ordinary launches registers (128 x 224 would spill).
template <int MT, int NT, int KS1, int WAVES, int OCC, bool SILU = false, int CTF = -1, int NV = 0, typename T = f16, bool GM = false,
          bool G128 = false>
__global__ __launch_bounds__(WAVES * 64, (OCC * WAVES + 3) / 4) void fq_kron_fast_kernel(const T* __restrict__ x, const uint4* __restrict__ ws,
                                                           const T* __restrict__ diag, int64_t rows, int M, int /*N*/,
                                                           FqQuantOut out, int flags_rt) {
    typedef typename FqVec<T>::x8 X8;
    static_assert(FqVec<T>::is_f16 || (!SILU && CTF < 0 && NV == 0), "bf16: the all-output-sets instantiation only");
    static_assert(!G128 || (!SILU && CTF < 0 && NV == 0), "FQ_GROUP128: the all-output-sets instantiation only");
    const int flags = CTF >= 0 ? (CTF | (flags_rt & (FQ_ROUND_Y_F16 | FQ_NO_CLAMP0 | FQ_SIG_F16 | 0xF000))) : flags_rt;
    constexpr int N = NV ? NV : KS1 * 16;          // N % 16 == 0 unless NV says otherwise, so KS1 fixes N
    constexpr bool ODD = NV != 0;
    static_assert(!ODD || (CTF == FQ_OUT_PACKED && !SILU && NV % 4 == 0 && NV > KS1 * 16 - 16 && NV < KS1 * 16), "NV: packed-only");
    constexpr int THREADS = WAVES * 64;
    constexpr int TPW = (NT + WAVES - 1) / WAVES;  // n'-tiles per wave
    constexpr int PITCH = (KS1 * 2) | 1;           // LDS row pitch of the staged token, in 16-byte chunks (odd)
    constexpr int XS_CHUNKS = MT * 32 * PITCH;
    constexpr int NPF = ODD ? (MT * 32 * (N / 4) + 2 * THREADS - 1) / (2 * THREADS)    // (ODD: two 8-byte units per register)
                            : (MT * 32 * KS1 * 2 + THREADS - 1) / THREADS;  // prefetch registers (uint4) per thread, upper bound
    constexpr int LFR = 2 * MT * MT * 64;          // L fragments, uint4 each
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* lfr = reinterpret_cast<uint4*>(smem);
    uint4* xs = lfr + LFR;                                             // [MT*32][PITCH]
    unsigned char* obuf = reinterpret_cast<unsigned char*>(xs + XS_CHUNKS);  // packed output stage: M*N/2 bytes
    float* red = reinterpret_cast<float*>(obuf + ((M * N / 2 + 15) & ~15));  // [2][WAVES]

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int cpr = N >> 3;      // 16-byte chunks per token row (ODD: unused, rows are N / 4 8-byte units)
    const int n_chunks = M * cpr;    // chunks per token
    constexpr int upr = N >> 2;      // ODD: 8-byte units per token row
    const int n_units = M * upr;
    const int64_t d = (int64_t)M * N;

    // ---- once per workgroup ----
    const uint4* lsrc = ws + (size_t)NT * KS1 * 64;
    for (int i = tid; i < LFR; i += THREADS) lfr[i] = lsrc[i];
    for (int i = tid; i < XS_CHUNKS; i += THREADS) xs[i] = make_uint4(0, 0, 0, 0);  // padding stays zero
    X8 RF[TPW][KS1];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int nt = wave + WAVES * t;
#pragma unroll
        for (int s = 0; s < KS1; ++s)
            RF[t][s] = nt < NT ? __builtin_bit_cast(X8, ws[((size_t)nt * KS1 + s) * 64 + lane]) : __builtin_bit_cast(X8, u32x4{0, 0, 0, 0});
    }
    // Make the R fragments "arrived" in the compiler's bookkeeping HERE: otherwise it covers their first use inside
    // the token loop with an s_waitcnt vmcnt(n), and that counter also sees the hand-issued prefetch loads in flight
    // there (the wait for loads of the prologue then drains the prefetch in front of GEMM 1).
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int s = 0; s < KS1; ++s) asm volatile("" : "+v"(RF[t][s]));
    // LDS slot of prefetch register k of this thread (chunk q = tid + 256 k of the token)
    u32x4 PF[NPF];
    u32x2 PF8[ODD ? 2 * NPF : 1];  // ODD: the prefetch registers are 8-byte units (the asm load writes them directly)
    u32x4 PF2[SILU ? NPF : 1];  // FQ_IN_SILU_MUL: x is `gate`, out.in2 is `up`; x_up * silu(x_gate) is formed while staging
    int64_t tok = blockIdx.x;
    // Prefetch loads are inline asm with a hand-placed wait: left to hipcc, an s_waitcnt vmcnt(1) appeared in front of
    // GEMM 1's first MFMA, i.e. the loads that were meant to land during the multiplication were waited for before it.
    // Out-of-range chunks of a ragged last register re-load the token's last chunk (no divergent branch around the asm).
#define FQ_PF_LOAD(tokidx)                                                                               \
    if (ODD) {                                                                                           \
        const uint2* xp_ = reinterpret_cast<const uint2*>(x + (tokidx) * d);                             \
        _Pragma("unroll") for (int k = 0; k < 2 * NPF; ++k) {                                            \
            int q_ = pf_q0 + THREADS * k;                                                                \
            q_ = q_ < n_units ? q_ : n_units - 1;                                                        \
            asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(PF8[k]) : "v"(xp_ + q_) : "memory"); \
        }                                                                                                \
    } else {                                                                                             \
        const u32x4* xp_ = reinterpret_cast<const u32x4*>(x + (tokidx) * d);                             \
        _Pragma("unroll") for (int k = 0; k < NPF; ++k) {                                                \
            int q_ = pf_q0 + THREADS * k;                                                                \
            q_ = q_ < n_chunks ? q_ : n_chunks - 1;                                                      \
            asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(PF[k]) : "v"(xp_ + q_) : "memory"); \
            if (SILU) {                                                                                  \
                const u32x4* up_ = reinterpret_cast<const u32x4*>(out.in2 + (tokidx) * d);               \
                asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(PF2[k]) : "v"(up_ + q_) : "memory"); \
            }                                                                                            \
        }                                                                                                \
    }
    int pf_q0 = tid;
    if (tok < rows) FQ_PF_LOAD(tok)
    FqGroupCursor gcur;  // grouped launches: the clip pair follows the token's group
    int mats_g = 0;      // ws_group_stride != 0: the group whose fragment image is loaded (group 0's after the prologue)
    // The zero fill of xs above and the first token's staging below touch the same LDS words from DIFFERENT threads
    // (fill: chunk tid + k THREADS; staging: row * PITCH + chunk): without this barrier a wave that is late in the
    // prologue (cold instruction cache on a kernel's first launches) zeroes rows another wave has already staged —
    // seen as one wrong output token per affected workgroup, about once in a few dozen fresh processes (round 2).
    __syncthreads();

    for (; tok < rows; tok += gridDim.x) {
        // ---- stage the prefetched token (the previous token's readers passed the statistics barrier) ----
        // The thread index is laundered once per token: otherwise every per-chunk address (global pointer, LDS slot,
        // diag pointer) is loop-invariant, gets hoisted out of the token loop and SPILLED (37 VGPRs in the 112x128
        // build), and each reload sat in front of its load behind an s_waitcnt vmcnt(0) that serialised the seven
        // prefetch loads into seven HBM round trips per token (the kernel ran 2x slower than it should).
        int q0 = tid;
        asm volatile("" : "+v"(q0));
        pf_q0 = q0;
        {   // the prefetch has had a whole token's time to land
            if (ODD) {
#pragma unroll
                for (int k = 0; k < 2 * NPF; ++k) asm volatile("s_waitcnt vmcnt(0)" : "+v"(PF8[k]));
            } else if (NPF == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(PF[0]));
            else if (NPF <= 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(PF[0]), "+v"(PF[1 % NPF]), "+v"(PF[2 % NPF]), "+v"(PF[3 % NPF]));
            else {
#pragma unroll
                for (int k = 0; k < NPF; k += 4)
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(PF[k]), "+v"(PF[(k + 1) % NPF]), "+v"(PF[(k + 2) % NPF]), "+v"(PF[(k + 3) % NPF]));
            }
            if (SILU) {
#pragma unroll
                for (int k = 0; k < NPF; ++k) asm volatile("" : "+v"(PF2[k]));  // arrived with the vmcnt(0) above
            }
            const uint4* dp = reinterpret_cast<const uint4*>(diag);
            if (ODD) {
                unsigned char* xsb = reinterpret_cast<unsigned char*>(xs);
#pragma unroll
                for (int k = 0; k < 2 * NPF; ++k) {
                    const int q = q0 + THREADS * k;
                    if (q < n_units) {
                        const int row = q / upr, u = q - row * upr;
                        *reinterpret_cast<uint2*>(xsb + (row * PITCH) * 16 + u * 8) = __builtin_bit_cast(uint2, PF8[k]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < (ODD ? 0 : NPF); ++k) {
                const int q = q0 + THREADS * k;
                if (q < n_chunks) {
                    uint4 v = __builtin_bit_cast(uint4, PF[k]);
                    if constexpr (SILU)
                        v = __builtin_bit_cast(uint4, fq_silu_mul8(__builtin_bit_cast(f16x8, PF[k]),
                                                                   __builtin_bit_cast(f16x8, PF2[k])));
                    if (diag != nullptr)
                        v = __builtin_bit_cast(uint4, __builtin_bit_cast(X8, v) * __builtin_bit_cast(X8, dp[q]));
                    const int row = q / cpr, ch = q - row * cpr;
                    xs[row * PITCH + ch] = v;
                }
            }
        }
        if (GM) {
            // every group has its own factor pair (routed_w2_trans[i], deepseekv3_utils.py:446): when the token's group
            // changes — rows are sorted by group, so once per group and workgroup — the R fragments (registers) and the L
            // image (LDS) are re-read from that group's image in the workspace (L2-resident, a few KB)
            fq_group_locate(out, tok, gcur);
            if (gcur.g != mats_g) {
                mats_g = gcur.g;
                const uint4* wg = ws + (size_t)mats_g * out.ws_group_stride;
                const uint4* lg = wg + (size_t)NT * KS1 * 64;
                for (int i = tid; i < LFR; i += THREADS) lfr[i] = lg[i];   // (every wave passed the previous token's barriers)
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const int nt = wave + WAVES * t;
#pragma unroll
                    for (int s = 0; s < KS1; ++s)
                        if (nt < NT) RF[t][s] = __builtin_bit_cast(X8, wg[((size_t)nt * KS1 + s) * 64 + lane]);
                }
            }
        }
        __syncthreads();
        if (tok + gridDim.x < rows && !(flags & 0x4000)) FQ_PF_LOAD(tok + gridDim.x)  // lands during this token's work

        int loff = lane;
        asm volatile("" : "+v"(loff));  // keep the L-fragment reads inside the token loop (see fq_kron64.hip)
        const uint4* mylfr = lfr + loff;
        f32x16 Y[TPW][MT];  // Y^T of tile (nt = wave + WAVES t, mo): rows n' = h*NT*16 + nt*16 + r, col m' = 32mo + c
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int nt = wave + WAVES * t;
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) Y[t][mo] = f32x16{0};
            if (nt < NT && !(flags & 0x1000)) {
                // Both GEMMs are software-pipelined by hand, one K-step of LDS fragment reads ahead of the MFMAs,
                // with a scheduling barrier per step: left alone, hipcc hoists ALL fragment reads of a GEMM (128
                // VGPRs each) in front of its first MFMA and spills.
                f32x16 U[MT];
                X8 A[2][MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    U[mt] = f32x16{0};
                    A[0][mt] = __builtin_bit_cast(X8, xs[(mt * 32 + c) * PITCH + h]);
                }
#pragma unroll
                for (int s = 0; s < KS1; ++s) {
                    if (s + 1 < KS1) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            A[(s + 1) & 1][mt] = __builtin_bit_cast(X8, xs[(mt * 32 + c) * PITCH + (s + 1) * 2 + h]);
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) U[mt] = fq_mfma32<T>(A[s & 1][mt], RF[t][s], U[mt]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                X8 Uh[MT][2];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int j = 0; j < 8; ++j) Uh[mt][p][j] = (T)U[mt][p * 8 + j];
                X8 B[2][MT];
#pragma unroll
                for (int mo = 0; mo < MT; ++mo) B[0][mo] = __builtin_bit_cast(X8, mylfr[mo * 64]);
#pragma unroll
                for (int ks = 0; ks < 2 * MT; ++ks) {
                    if (ks + 1 < 2 * MT) {
#pragma unroll
                        for (int mo = 0; mo < MT; ++mo)
                            B[(ks + 1) & 1][mo] = __builtin_bit_cast(X8, mylfr[((ks + 1) * MT + mo) * 64]);
                    }
#pragma unroll
                    for (int mo = 0; mo < MT; ++mo) Y[t][mo] = fq_mfma32<T>(Uh[ks >> 1][ks & 1], B[ks & 1][mo], Y[t][mo]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // Packed-only instantiations with the fp16 (deploy Quantizer) arithmetic: the whole epilogue runs on PACKED fp16
        // pairs H — post-scale + rounding to fp16 (the product is rounded to fp32 first, as in the generic path below),
        // extrema with v_pk_max/min_f16 — and the quantiser takes the pairs (fq_quant8_h16). Same bits as the generic path.
        constexpr bool H16 = CTF == (FQ_OUT_PACKED | FQ_QUANT_F16) && FqVec<T>::is_f16;
        uint32_t H[H16 ? TPW : 1][H16 ? MT : 1][8];
        float vmax = -INFINITY, vmin = INFINITY;
        if (H16) {
            const float ps = out.post_scale != 0.0f ? out.post_scale : 1.0f;
            f16x2 pmax = {(f16)-INFINITY, (f16)-INFINITY}, pmin = {(f16)INFINITY, (f16)INFINITY};
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int nt = wave + WAVES * t;
                const bool col_ok = nt < NT && (h * NT * 16 + nt * 16) < N;
#pragma unroll
                for (int mo = 0; mo < MT; ++mo) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const f16x2 pr = fq_mul_to_f16x2(Y[t][mo][2 * j], Y[t][mo][2 * j + 1], f32x2{ps, ps});
                        H[H16 ? t : 0][H16 ? mo : 0][j] = __builtin_bit_cast(uint32_t, pr);
                        if (col_ok && (mo * 32 + c) < M) {
                            pmax = fq_pk_max(pmax, pr);
                            pmin = fq_pk_min(pmin, pr);
                        }
                    }
                }
            }
            vmax = fmaxf((float)pmax[0], (float)pmax[1]);
            vmin = fminf((float)pmin[0], (float)pmin[1]);
        }
        if (!H16 && out.post_scale != 0.0f) {  // (fq_kron_quant_ex_f16: e.g. the 1/sqrt(n) of a Hadamard rotation run as a Kronecker product)
            const float ps = out.post_scale;
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float p = Y[t][mo][r] * ps;
                        asm volatile("" : "+v"(p));  // the product is an fp32 VALUE (no fusion with a later rounding to fp16)
                        Y[t][mo][r] = p;
                    }
        }
        if (!H16 && (flags & FQ_ROUND_Y_F16)) {
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Y[t][mo][r] = (float)(T)Y[t][mo][r];
        }

        // ---- FQ_GROUP128 (ActivationQuantizer(groupsize=128), vllm_custom/.../fake_quant_utils.py:72-78; round 3): one scale per
        // 128 CONSECUTIVE elements of the transformed token. The reference hands the quantiser the transformed tensor, i.e. Y
        // rounded to the activation dtype (path A, FQ_ROUND_Y_F16 required): the token is staged in xs in its own layout — as
        // for the transform output — and a second pass walks it in linear order: a thread owns a 16-byte chunk, a group is 16
        // consecutive chunks = 16 consecutive lanes, extrema by xor butterflies inside them, every lane its group's scale,
        // outputs written straight to HBM, fully coalesced. (Before: transform launch + row-quantiser launch, 8 d bytes of HBM
        // traffic per token instead of 4 d.)
        if (G128) {   // (its own instantiations: in the common ones the epilogue cost 15-30 VGPRs and a wave per SIMD)
            __syncthreads();   // every wave has finished reading xs (GEMM 1)
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int nt = wave + WAVES * t, n0 = h * NT * 16 + nt * 16;
#pragma unroll
                for (int mo = 0; mo < MT; ++mo)
                    if (nt < NT && n0 < N && (mo * 32 + c) < M) {
                        X8 v0, v1;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            v0[e] = (T)Y[t][mo][e];
                            v1[e] = (T)Y[t][mo][8 + e];
                        }
                        uint4* sp = xs + (mo * 32 + c) * PITCH + (n0 >> 3);
                        sp[0] = __builtin_bit_cast(uint4, v0);
                        sp[1] = __builtin_bit_cast(uint4, v1);
                    }
            }
            __syncthreads();
            float sig_max = out.sig_max[0], sig_min = out.sig_min[0];
            fq_token_sigs(out, 0, tok, gcur, sig_max, sig_min);
            for (int q = tid; q < n_chunks; q += THREADS) {   // (n_chunks % 16 == 0: a group never straddles the loop stride)
                const int row = q / cpr, ch = q - row * cpr;
                const X8 v = __builtin_bit_cast(X8, xs[row * PITCH + ch]);
                RowExtrema<T> ext;
                ext.take(v);
                float gmax = ext.vmax(), gmin = ext.vmin();
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    gmax = fmaxf(gmax, __shfl_xor(gmax, m));
                    gmin = fminf(gmin, __shfl_xor(gmin, m));
                }
                float scale;
                if (flags & FQ_QUANT_F16) scale = fq_token_scale<FQ_QUANT_F16, T>(gmax, gmin, sig_max, sig_min, flags);
                else scale = fq_token_scale<0, T>(gmax, gmin, sig_max, sig_min, flags);
                const float inv = fq_fast_inv(scale);
                // the fake-quant contract alone, fp32 arithmetic (FlatQuantizedLinear / the vLLM quantiser): the single-width asm
                // block fq_fake8 — wave-uniform route (the clamped form is right for every lane; one exactness vote per chunk row)
                bool done = false;
                if ((flags & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT | FQ_QUANT_F16)) == FQ_OUT_FAKEQUANT && !__any(!fq_magic_ok(gmax, gmin, inv))) {
                    float dmax = 0.0f;
                    u32x4 o;
                    if (__any(fq_needs_clamp(gmax, gmin, inv)))
                        o = fq_fake8<true, T>((float)v[0], (float)v[1], (float)v[2], (float)v[3], (float)v[4], (float)v[5], (float)v[6], (float)v[7], inv, scale, dmax);
                    else
                        o = fq_fake8<false, T>((float)v[0], (float)v[1], (float)v[2], (float)v[3], (float)v[4], (float)v[5], (float)v[6], (float)v[7], inv, scale, dmax);
                    if (!fq_wave_needs_exact(dmax)) {
                        __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(reinterpret_cast<T*>(out.fq[0]) + tok * d) + q);
                        done = true;
                    }
                }
                int dq[8];   // the digits
                if (done) {
                } else if (flags & FQ_QUANT_F16) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) dq[e] = fq_quant1_h(v[e], (T)scale);
                } else {
                    float dmax = 0.0f, r[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = fq_qfast((float)v[e], inv, dmax);
                    if (fq_wave_needs_exact(dmax)) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) r[e] = fq_qexact((float)v[e], scale);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) dq[e] = (int)r[e];
                }
                if (flags & FQ_OUT_TRANSFORM)
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(reinterpret_cast<T*>(out.y) + tok * d) + q);
                if ((flags & FQ_OUT_FAKEQUANT) && !done) {
                    X8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (flags & FQ_QUANT_F16) o[e] = fq_dequant1<FQ_QUANT_F16, T>(dq[e], scale);
                        else o[e] = fq_fake<T>(scale, (float)dq[e]);
                    }
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), reinterpret_cast<u32x4*>(reinterpret_cast<T*>(out.fq[0]) + tok * d) + q);
                }
                if (flags & FQ_OUT_PACKED) {
                    uint32_t pk = 0;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pk |= (uint32_t)(dq[e] & 15) << (4 * e);
                    reinterpret_cast<uint32_t*>(out.q[0] + tok * (d >> 1))[q] = pk;
                }
                if ((flags & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT)) && !(q & 15) && out.scale[0] != nullptr)
                    reinterpret_cast<T*>(out.scale[0])[tok * (d >> 7) + (q >> 4)] = (T)scale;
            }
            __syncthreads();   // the stage is read: the next token may overwrite it
            continue;
        }

        // ---- per-token extrema over the VALID entries (padding rows/columns are excluded) ----
#pragma unroll
        for (int t = 0; t < (H16 ? 0 : TPW); ++t) {
            const int nt = wave + WAVES * t;
            const int nval = N - (h * NT * 16 + nt * 16);                // valid columns of the lane's run of this tile
            const bool col_ok = nt < NT && nval > 0;                     // the lane's 16 columns of this tile
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) {
                if (col_ok && (mo * 32 + c) < M) {
                    if (!ODD || nval >= 16) {
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            vmax = fq_max3(vmax, Y[t][mo][r], Y[t][mo][r + 1]);
                            vmin = fq_min3(vmin, Y[t][mo][r], Y[t][mo][r + 1]);
                        }
                    } else {  // the run N cuts (N % 4 == 0: whole pairs)
#pragma unroll
                        for (int r = 0; r < 16; r += 2)
                            if (r < nval) {
                                vmax = fq_max3(vmax, Y[t][mo][r], Y[t][mo][r + 1]);
                                vmin = fq_min3(vmin, Y[t][mo][r], Y[t][mo][r + 1]);
                            }
                    }
                }
            }
        }
        vmax = fq_wave_max(vmax);
        vmin = fq_wave_min(vmin);
        if (lane == 0) {
            red[wave] = vmax;
            red[WAVES + wave] = vmin;
        }
        __syncthreads();  // also: every wave has finished reading xs -> it may be reused as an output stage
        vmax = red[0];
        vmin = red[WAVES];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            vmax = fmaxf(vmax, red[w]);
            vmin = fminf(vmin, red[WAVES + w]);
        }

        // ---- 16-bit outputs (transform / fake-quant) are staged in xs IN THE TOKEN'S OWN LAYOUT (row pitch PITCH chunks: the
        // padding chunks and rows are never written and stay zero for the next token — round 3; the dense stage of round 2
        // had to be re-zeroed, 35 KB of LDS writes and a barrier per token), then streamed out in whole 16-byte chunks ----
        if (flags & FQ_OUT_TRANSFORM) {
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int nt = wave + WAVES * t, n0 = h * NT * 16 + nt * 16;
#pragma unroll
                for (int mo = 0; mo < MT; ++mo)
                    if (nt < NT && n0 < N && (mo * 32 + c) < M) {
                        X8 v0, v1;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            v0[e] = (T)Y[t][mo][e];
                            v1[e] = (T)Y[t][mo][8 + e];
                        }
                        uint4* sp = xs + (mo * 32 + c) * PITCH + (n0 >> 3);
                        sp[0] = __builtin_bit_cast(uint4, v0);
                        sp[1] = __builtin_bit_cast(uint4, v1);
                    }
            }
            __syncthreads();
            u32x4* yp = reinterpret_cast<u32x4*>(reinterpret_cast<T*>(out.y) + tok * d);
            for (int q = tid; q < n_chunks; q += THREADS) {
                const int row = q / cpr, ch = q - row * cpr;
                __builtin_nontemporal_store(__builtin_bit_cast(u32x4, xs[row * PITCH + ch]), yp + q);
            }
            __syncthreads();
        }

        for (int ci = 0; ci < out.n_clips; ++ci) {
            if (!(flags & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT))) break;
            float scale, sig_max = out.sig_max[ci], sig_min = out.sig_min[ci];
            if (!SILU) fq_token_sigs(out, ci, tok, gcur, sig_max, sig_min);  // (the SiLU.mul launches are never grouped)
            if (flags & FQ_QUANT_F16) scale = fq_token_scale<FQ_QUANT_F16, T>(vmax, vmin, sig_max, sig_min, flags);
            else scale = fq_token_scale<0, T>(vmax, vmin, sig_max, sig_min, flags);
            const float inv = fq_fast_inv(scale);
            const FqH16Recip rc = H16 ? fq_h16_recip(scale) : FqH16Recip{0.0f, 0.0f};
            const f32x2 inv2 = {inv, inv};
            const bool magic = !(flags & FQ_QUANT_F16) && fq_magic_ok(vmax, vmin, inv);
            const bool clampq = fq_needs_clamp(vmax, vmin, inv);

            // The fake-quant contract alone (FlatQuantizedLinear._eval_forward; fp32 quantiser arithmetic): the single-width asm
            // block fq_fake8 (fq_common.hpp) per half tile, ONE exactness vote per wave and token; an ambiguous digit anywhere in
            // the wave (~3 % of tokens) sends the wave through the generic code below, which then rewrites the same stage slots.
            bool fake_done = false;
            if (CTF < 0 && (flags & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT | FQ_QUANT_F16 | 0x2000)) == FQ_OUT_FAKEQUANT && magic) {
                float dmax = 0.0f;
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const int nt = wave + WAVES * t, n0 = h * NT * 16 + nt * 16;
#pragma unroll
                    for (int mo = 0; mo < MT; ++mo) {
                        const f32x16& yv = Y[t][mo];
                        u32x4 o0, o1;
                        if (clampq) {
                            o0 = fq_fake8<true, T>(yv[0], yv[1], yv[2], yv[3], yv[4], yv[5], yv[6], yv[7], inv, scale, dmax);
                            o1 = fq_fake8<true, T>(yv[8], yv[9], yv[10], yv[11], yv[12], yv[13], yv[14], yv[15], inv, scale, dmax);
                        } else {
                            o0 = fq_fake8<false, T>(yv[0], yv[1], yv[2], yv[3], yv[4], yv[5], yv[6], yv[7], inv, scale, dmax);
                            o1 = fq_fake8<false, T>(yv[8], yv[9], yv[10], yv[11], yv[12], yv[13], yv[14], yv[15], inv, scale, dmax);
                        }
                        if (nt < NT && n0 < N && (mo * 32 + c) < M) {
                            uint4* sp = xs + (mo * 32 + c) * PITCH + (n0 >> 3);
                            sp[0] = __builtin_bit_cast(uint4, o0);
                            sp[1] = __builtin_bit_cast(uint4, o1);
                        }
                    }
                }
                fake_done = !fq_wave_needs_exact(dmax);
            }

#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                if (fake_done) break;   // (wave-uniform; a run-time trip count would push Y into scratch)
                const int nt = wave + WAVES * t, n0 = h * NT * 16 + nt * 16;
#pragma unroll
                for (int mo = 0; mo < MT; ++mo) {
                    const bool ok = nt < NT && n0 < N && (mo * 32 + c) < M;
                    const f32x16& yv = Y[t][mo];
                    if (CTF == FQ_OUT_PACKED && !(flags & 0x2000)) {
                        // packed-only instantiations: the single-width asm quantiser (fq_quant8_two, fq_common.hpp) — no
                        // v_pk_*_f32 next to the other wave's MFMAs, no per-element residual bookkeeping
                        uint2 pk = {0u, 0u};
                        unsigned long long d0 = ~0ull, d1 = ~0ull;
                        if (magic) {
                            const float ilo = fq_inv_lo(inv), ihi = fq_inv_hi(inv);
                            if (clampq) {
                                pk.x = fq_quant8<true>(yv[0], yv[1], yv[2], yv[3], yv[4], yv[5], yv[6], yv[7], inv, ilo, ihi, d0);
                                pk.y = fq_quant8<true>(yv[8], yv[9], yv[10], yv[11], yv[12], yv[13], yv[14], yv[15], inv, ilo, ihi, d1);
                            } else {
                                pk.x = fq_quant8<false>(yv[0], yv[1], yv[2], yv[3], yv[4], yv[5], yv[6], yv[7], inv, ilo, ihi, d0);
                                pk.y = fq_quant8<false>(yv[8], yv[9], yv[10], yv[11], yv[12], yv[13], yv[14], yv[15], inv, ilo, ihi, d1);
                            }
                        }
                        if (d0)  // rare: an ambiguous digit somewhere in the wave -> the true division for this dword
                            pk.x = fq_pack8(fq_qexact(yv[0], scale), fq_qexact(yv[1], scale), fq_qexact(yv[2], scale), fq_qexact(yv[3], scale),
                                            fq_qexact(yv[4], scale), fq_qexact(yv[5], scale), fq_qexact(yv[6], scale), fq_qexact(yv[7], scale));
                        if (d1)
                            pk.y = fq_pack8(fq_qexact(yv[8], scale), fq_qexact(yv[9], scale), fq_qexact(yv[10], scale), fq_qexact(yv[11], scale),
                                            fq_qexact(yv[12], scale), fq_qexact(yv[13], scale), fq_qexact(yv[14], scale), fq_qexact(yv[15], scale));
                        if (ODD) {  // rows of N / 2 bytes are only 2-byte aligned: four 2-byte pieces (4 digits each), cut by N
                            if (ok) {
                                unsigned short* op = reinterpret_cast<unsigned short*>(obuf + (mo * 32 + c) * (N >> 1) + (n0 >> 1));
                                const int nval = N - n0;
                                op[0] = (unsigned short)pk.x;
                                if (nval > 4) op[1] = (unsigned short)(pk.x >> 16);
                                if (nval > 8) op[2] = (unsigned short)pk.y;
                                if (nval > 12) op[3] = (unsigned short)(pk.y >> 16);
                            }
                            continue;
                        }
                        if (ok) *reinterpret_cast<uint2*>(obuf + (mo * 32 + c) * (N >> 1) + (n0 >> 1)) = pk;
                        continue;
                    }
                    if (H16 && !(flags & 0x2000)) {
                        // the fp16 pairs of this tile: exact fp16 quotient without a division, packed rounding and pack
                        const uint32_t(&hv)[8] = H[H16 ? t : 0][H16 ? mo : 0];
                        uint2 pk;
                        if (clampq) {
                            pk.x = fq_quant8_h16<true>(hv[0], hv[1], hv[2], hv[3], rc);
                            pk.y = fq_quant8_h16<true>(hv[4], hv[5], hv[6], hv[7], rc);
                        } else {
                            pk.x = fq_quant8_h16<false>(hv[0], hv[1], hv[2], hv[3], rc);
                            pk.y = fq_quant8_h16<false>(hv[4], hv[5], hv[6], hv[7], rc);
                        }
                        if (ok) *reinterpret_cast<uint2*>(obuf + (mo * 32 + c) * (N >> 1) + (n0 >> 1)) = pk;
                        continue;
                    }
                    f32x2 qp[8];  // integer-valued pairs (r_2j, r_2j+1)
                    bool exact = !magic;
                    if (flags & 0x2000) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) qp[j] = f32x2{yv[2 * j], yv[2 * j + 1]};
                        exact = false;
                    } else if (magic) {
                        float dmax = 0.0f;
                        if (clampq) {
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                qp[j] = fq_qmagic2<true>(f32x2{yv[2 * j], yv[2 * j + 1]}, inv2, dmax);
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                qp[j] = fq_qmagic2<false>(f32x2{yv[2 * j], yv[2 * j + 1]}, inv2, dmax);
                        }
                        exact = fq_wave_needs_exact(dmax);
                    }
                    if (exact) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            if (flags & FQ_QUANT_F16)
                                qp[j] = f32x2{(float)fq_quant1<FQ_QUANT_F16, T>(yv[2 * j], scale),
                                              (float)fq_quant1<FQ_QUANT_F16, T>(yv[2 * j + 1], scale)};
                            else
                                qp[j] = f32x2{fq_qexact(yv[2 * j], scale), fq_qexact(yv[2 * j + 1], scale)};
                        }
                    }
                    if (ok && (flags & FQ_OUT_PACKED)) {
                        uint2 pk;
                        pk.x = fq_pack8p(qp[0], qp[1], qp[2], qp[3]);
                        pk.y = fq_pack8p(qp[4], qp[5], qp[6], qp[7]);
                        *reinterpret_cast<uint2*>(obuf + (mo * 32 + c) * (N >> 1) + (n0 >> 1)) = pk;
                    }
                    if (ok && (flags & FQ_OUT_FAKEQUANT)) {
                        X8 v0, v1;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float q0 = (e & 1) ? qp[e >> 1].y : qp[e >> 1].x;
                            const float q1 = (e & 1) ? qp[4 + (e >> 1)].y : qp[4 + (e >> 1)].x;
                            if (flags & FQ_QUANT_F16) {
                                v0[e] = fq_dequant1<FQ_QUANT_F16, T>((int)q0, scale);
                                v1[e] = fq_dequant1<FQ_QUANT_F16, T>((int)q1, scale);
                            } else {
                                v0[e] = fq_fake<T>(scale, q0);
                                v1[e] = fq_fake<T>(scale, q1);
                            }
                        }
                        uint4* sp = xs + (mo * 32 + c) * PITCH + (n0 >> 3);
                        sp[0] = __builtin_bit_cast(uint4, v0);
                        sp[1] = __builtin_bit_cast(uint4, v1);
                    }
                }
            }
            __syncthreads();
            if (flags & FQ_OUT_PACKED) {
                if (tid == 0) reinterpret_cast<T*>(out.scale[ci])[tok] = (T)scale;
                uint4* qp4 = reinterpret_cast<uint4*>(out.q[ci] + tok * (d >> 1));
                for (int q = tid; q < (M * N) / 32; q += THREADS) qp4[q] = reinterpret_cast<const uint4*>(obuf)[q];
            }
            if (flags & FQ_OUT_FAKEQUANT) {
                u32x4* fp = reinterpret_cast<u32x4*>(reinterpret_cast<T*>(out.fq[ci]) + tok * d);
                for (int q = tid; q < n_chunks; q += THREADS) {
                    const int row = q / cpr, ch = q - row * cpr;
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, xs[row * PITCH + ch]), fp + q);
                }
            }
            if ((flags & FQ_OUT_FAKEQUANT) || ci + 1 < out.n_clips) __syncthreads();  // stage / obuf are rewritten next
        }

    }
}

template <int MT, int NT, int KS1, int WAVES, int OCC, bool SILU = false, int CTF = -1, int NV = 0, typename T = f16, bool GM = false,
          bool G128 = false>
int launch_fast(int flags, const T* x, const uint4* ws, const T* diag, int64_t rows, int M, int N,
                const FqQuantOut& out, int n_cu, hipStream_t stream) {
    constexpr int PITCH = (KS1 * 2) | 1;
    const size_t lds = (size_t)2 * MT * MT * 1024 + (size_t)MT * 32 * PITCH * 16 + (((size_t)M * N / 2 + 15) & ~(size_t)15) + 128;
    if (lds > 160 * 1024) return -1000;
    if (G128 != ((out.rt_flags & FQ_GROUP128) != 0)) return -1000;   // (the group epilogue is its own instantiation: registers)
    auto kern = fq_kron_fast_kernel<MT, NT, KS1, WAVES, OCC, SILU, CTF, NV, T, GM, G128>;
    FQ_RAISE_LDS_CAP(kern, 160 * 1024);
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > OCC) per_cu = OCC;
    if (per_cu < 1) per_cu = 1;
    int64_t blocks = (int64_t)n_cu * per_cu;
    if (blocks > rows) blocks = rows;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WAVES * 64), lds, stream, x, ws, diag, rows, M, N, out, flags);
    return (int)hipGetLastError();
}

}  // namespace

static inline int tiles32(int n) { return (n + 31) / 32; }
int64_t fq_kron_generic_workspace_bytes(int M, int N);
int fq_launch_kron_prepare(const f16* left, const f16* right, int M, int N, void* workspace, hipStream_t stream, int groups = 1);
int fq_launch_kron_general(int flags, const f16* x, const void* ws, const f16* diag, int64_t rows, int M, int N,
                           const FqQuantOut& out, int n_cu, hipStream_t stream);
