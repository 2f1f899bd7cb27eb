// fq_kron64p.hip — software-pipelined variant of the d = 4096 fused kernel for the headline contract
// (single clip set, packed INT4 + fp16 scale out, fp32 statistics; FQ_NO_CLAMP0 honoured at run time).
//
// Same mathematics, data path and fragment chaining as fq_kron64.hip. The difference is the token loop: the
// quantise/pack work of token t (VALU) and the two MFMA GEMMs of token t+1 live in ONE basic block of ONE wave, so
// the scheduler can put the quantiser's instructions into the shadow of the 32-cycle MFMAs. In fq_kron64.hip the two
// only overlap across waves, and the PMC shows they mostly do not: SQ_VALU_MFMA_COEXEC_CYCLES = 21 % of the MFMA
// busy cycles, matrix pipe and VALU each idle more than half of the kernel.
//
// Cost: Y of two tokens is live (2 x 64 VGPRs) -> 512-thread workgroups, 2 waves per SIMD.
#include "fq_common.hpp"

namespace {

constexpr int KM = 64, KN = 64, KD = KM * KN;
constexpr int FRAG_BYTES = 16 * 64 * 16;
constexpr int TOK_BYTES = KD * 2;
constexpr int THREADS = 512, WAVES = THREADS / 64;

__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int nperm(int nt, int pos) {
    return ((pos >> 2) & 1) * 32 + nt * 16 + (pos & 3) + 4 * (pos >> 3);
}
typedef __attribute__((address_space(3))) void lds_void;

// identical to fq_kron64.hip::dma_token (LDS-DMA of one token, source-side swizzle, counted by the caller)
// `valid` (wave-uniform) is applied through EXEC inside the asm, NOT with a branch: the steady-state loop body must stay
// one basic block so that the scheduler can interleave the quantiser with the MFMAs.
__device__ __forceinline__ void dma_token(const f16* __restrict__ x, int64_t tok, unsigned lds_base, int lane,
                                          bool valid = true) {
    const int emask = __builtin_amdgcn_readfirstlane(valid ? -1 : 0);  // provably wave-uniform for the "s" operand
    const unsigned ce = ((lane & 7) ^ (lane >> 4)) << 4;
    const unsigned voff_e = (lane >> 3) * 128 + ce;
    const unsigned voff_o = (lane >> 3) * 128 + (ce ^ 64);
    const unsigned char* base = reinterpret_cast<const unsigned char*>(x) + tok * TOK_BYTES;
    const unsigned lo32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)base);
    const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((size_t)base >> 32));
    const unsigned long long sb0 = (unsigned long long)lo32 | ((unsigned long long)hi32 << 32);
    const unsigned long long sb1 = sb0 + 4096;
    unsigned keep;
    unsigned long long keep_exec;
    asm volatile(
        "s_nop 4\n\t"  // SGPR operands may come straight from v_readfirstlane: 5 wait states before VMEM reads them
        "s_mov_b64 %1, exec\n\t"
        "s_and_b32 exec_lo, exec_lo, %8\n\t"
        "s_and_b32 exec_hi, exec_hi, %8\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %4 nt\n\t"
        "global_load_lds_dwordx4 %3, %4 offset:1024 nt\n\t"
        "global_load_lds_dwordx4 %2, %4 offset:2048 nt\n\t"
        "global_load_lds_dwordx4 %3, %4 offset:3072 nt\n\t"
        "s_mov_b32 m0, %7\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %5 nt\n\t"
        "global_load_lds_dwordx4 %3, %5 offset:1024 nt\n\t"
        "global_load_lds_dwordx4 %2, %5 offset:2048 nt\n\t"
        "global_load_lds_dwordx4 %3, %5 offset:3072 nt\n\t"
        "s_mov_b32 m0, %0\n\t"
        "s_mov_b64 exec, %1"
        : "=&s"(keep), "=&s"(keep_exec)
        : "v"(voff_e), "v"(voff_o), "s"(sb0), "s"(sb1), "s"(lds_base), "s"(lds_base + 4096), "s"(emask)
        : "memory");
}

// per-token statistics of a finished Y (wave all-reduce) -> scale
__device__ __forceinline__ float token_scale(const f32x16 (&Y)[2][2], const FqQuantOut& out) {
    float pmax[4], pmin[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x16& t = Y[k >> 1][k & 1];
        float a = FqMaxOp()(t[0], t[1]), b = FqMinOp()(t[0], t[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) {
            a = fq_max3(a, t[r], t[r + 1]);
            b = fq_min3(b, t[r], t[r + 1]);
        }
        pmax[k] = a;
        pmin[k] = b;
    }
    float vmax = fq_max3(pmax[0], pmax[1], FqMaxOp()(pmax[2], pmax[3]));
    float vmin = fq_min3(pmin[0], pmin[1], FqMinOp()(pmin[2], pmin[3]));
    vmax = fq_wave_max(vmax);
    vmin = fq_wave_min(vmin);
    return fq_token_scale<0>(vmax, vmin, out.sig_max[0], out.sig_min[0], out.rt_flags);
}

#define FQ_YV(Y, mo, w, e) Y[(w) >> 1][mo][((w) & 1) * 8 + (e)]

// fast exact quantiser of one token into 8 dwords; returns the mask of dwords that need the exact redo
__device__ __forceinline__ unsigned quant_fast(const f32x16 (&Y)[2][2], float inv, uint32_t (&pw)[2][4]) {
    unsigned near = 0;
    const f32x2 inv2 = {inv, inv};
#pragma unroll
    for (int mo = 0; mo < 2; ++mo)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            float dmax = 0.0f;
            const f32x2 q01 = fq_qfast2(f32x2{FQ_YV(Y, mo, w, 0), FQ_YV(Y, mo, w, 1)}, inv2, dmax);
            const f32x2 q23 = fq_qfast2(f32x2{FQ_YV(Y, mo, w, 2), FQ_YV(Y, mo, w, 3)}, inv2, dmax);
            const f32x2 q45 = fq_qfast2(f32x2{FQ_YV(Y, mo, w, 4), FQ_YV(Y, mo, w, 5)}, inv2, dmax);
            const f32x2 q67 = fq_qfast2(f32x2{FQ_YV(Y, mo, w, 6), FQ_YV(Y, mo, w, 7)}, inv2, dmax);
            pw[mo][w] = fq_pack8(q01.x, q01.y, q23.x, q23.y, q45.x, q45.y, q67.x, q67.y);
            near |= fq_wave_needs_exact(dmax) ? (1u << (4 * mo + w)) : 0u;
        }
    return near;
}

__device__ __forceinline__ void quant_fix_and_store(const f32x16 (&Y)[2][2], float scale, unsigned near,
                                                    uint32_t (&pw)[2][4], uint8_t* __restrict__ q, f16* __restrict__ sc,
                                                    int64_t tok, int lane) {
    const int h = lane >> 5, c = lane & 31;
    if (near) {  // rare: a quotient within FQ_NEAR of a tie somewhere in the wave -> true division for those dwords
#pragma unroll
        for (int mo = 0; mo < 2; ++mo)
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if (near & (1u << (4 * mo + w)))
                    pw[mo][w] = fq_pack8(fq_qexact(FQ_YV(Y, mo, w, 0), scale), fq_qexact(FQ_YV(Y, mo, w, 1), scale),
                                         fq_qexact(FQ_YV(Y, mo, w, 2), scale), fq_qexact(FQ_YV(Y, mo, w, 3), scale),
                                         fq_qexact(FQ_YV(Y, mo, w, 4), scale), fq_qexact(FQ_YV(Y, mo, w, 5), scale),
                                         fq_qexact(FQ_YV(Y, mo, w, 6), scale), fq_qexact(FQ_YV(Y, mo, w, 7), scale));
    }
    if (lane == 0) sc[tok] = (f16)scale;
#pragma unroll
    for (int mo = 0; mo < 2; ++mo)
        *reinterpret_cast<uint4*>(q + tok * (KD / 2) + (mo * 32 + c) * (KN / 2) + h * 16) =
            make_uint4(pw[mo][0], pw[mo][1], pw[mo][2], pw[mo][3]);
}

__global__ __launch_bounds__(THREADS) void fq_kron64p_kernel(const f16* __restrict__ x, const f16* __restrict__ left,
                                                             const f16* __restrict__ right, int64_t rows,
                                                             FqQuantOut out) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[FRAG_BYTES + WAVES * TOK_BYTES + 16];
    unsigned* next_slot = reinterpret_cast<unsigned*>(smem + FRAG_BYTES + WAVES * TOK_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* tokbuf = smem + FRAG_BYTES + wave * TOK_BYTES;
    const unsigned tok_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)tokbuf);

    const int64_t tpb = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t blk_base = (int64_t)blockIdx.x * tpb;
    const int blk_cnt = (int)(rows - blk_base < tpb ? (rows - blk_base < 0 ? 0 : rows - blk_base) : tpb);
    if (tid == 0) *next_slot = WAVES;
    int slot = wave;
    if (slot < blk_cnt) dma_token(x, blk_base + slot, tok_lds, lane);

    uint4* frag = reinterpret_cast<uint4*>(smem);
    for (int item = tid; item < 16 * 64; item += THREADS) {
        const int f = item >> 6, ln = item & 63, fh = ln >> 5, fc = ln & 31;
        f16x8 v;
        if (f < 8) {
            const int nt = f >> 2, sk = f & 3, np = nperm(nt, fc);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = right[(fh * 32 + sk * 8 + j) * KN + np];
        } else {
            const int ks = (f - 8) >> 1, mo = (f - 8) & 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = (ks >> 1) * 32 + 16 * (ks & 1) + 8 * (j >> 2) + 4 * fh + (j & 3);
                v[j] = left[m * KM + mo * 32 + fc];
            }
        }
        frag[item] = __builtin_bit_cast(uint4, v);
    }
    __syncthreads();
    const int sw = (c >> 1) & 7;
    uint8_t* const qout = out.q[0];
    f16* const sout = out.scale[0];

    if (slot >= blk_cnt) return;

    // LOAD: token buffer -> A fragments; then the buffer is free -> refill it (DMA of slot `nx`, masked by `nx_ok`).
#define FQ_LOAD_X_AND_PREFETCH(nx, nx_ok)                                                        \
    u32x4 X[2][4];                                                                               \
    {                                                                                            \
        const u32x4* tb = reinterpret_cast<const u32x4*>(tokbuf);                                \
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                         \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                            \
            X[mt][s] = tb[(mt * 32 + c) * 8 + ((h * 4 + s) ^ sw)];                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
        dma_token(x, blk_base + ((nx_ok) ? (nx) : 0), tok_lds, lane, (nx_ok));                   \
    }
    // GEMMS: Y = L^T . fp16(X . R) on 32 MFMAs (see fq_kron64.hip for the fragment chaining)
#define FQ_GEMMS()                                                                               \
    f32x16 Y[2][2];                                                                              \
    {                                                                                            \
        int foff = lane;                                                                         \
        asm volatile("" : "+v"(foff));                                                           \
        const uint4* myfrag = frag + foff;                                                       \
        f32x16 U[2][2];                                                                          \
        U[0][0] = f32x16{0}; U[0][1] = f32x16{0}; U[1][0] = f32x16{0}; U[1][1] = f32x16{0};      \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                          \
            const f16x8 b0 = __builtin_bit_cast(f16x8, myfrag[(0 * 4 + s) * 64]);                \
            const f16x8 b1 = __builtin_bit_cast(f16x8, myfrag[(1 * 4 + s) * 64]);                \
            U[0][0] = mfma32(__builtin_bit_cast(f16x8, X[0][s]), b0, U[0][0]);                   \
            U[1][0] = mfma32(__builtin_bit_cast(f16x8, X[1][s]), b0, U[1][0]);                   \
            U[0][1] = mfma32(__builtin_bit_cast(f16x8, X[0][s]), b1, U[0][1]);                   \
            U[1][1] = mfma32(__builtin_bit_cast(f16x8, X[1][s]), b1, U[1][1]);                   \
        }                                                                                        \
        f16x8 Uh[2][4];                                                                          \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                         \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                         \
        _Pragma("unroll") for (int j = 0; j < 8; ++j)                                            \
            Uh[nt][ks][j] = (f16)U[ks >> 1][nt][(ks & 1) * 8 + j];                               \
        Y[0][0] = f32x16{0}; Y[0][1] = f32x16{0}; Y[1][0] = f32x16{0}; Y[1][1] = f32x16{0};      \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                       \
            const f16x8 b0 = __builtin_bit_cast(f16x8, myfrag[(8 + ks * 2 + 0) * 64]);           \
            const f16x8 b1 = __builtin_bit_cast(f16x8, myfrag[(8 + ks * 2 + 1) * 64]);           \
            Y[0][0] = mfma32(Uh[0][ks], b0, Y[0][0]);                                            \
            Y[1][0] = mfma32(Uh[1][ks], b0, Y[1][0]);                                            \
            Y[0][1] = mfma32(Uh[0][ks], b1, Y[0][1]);                                            \
            Y[1][1] = mfma32(Uh[1][ks], b1, Y[1][1]);                                            \
        }                                                                                        \
    }
    // FINISH: optional fp16 rounding, statistics -> scale, Y becomes the "previous" token
#define FQ_FINISH_Y(tok_)                                                                        \
    if (out.rt_flags & FQ_ROUND_Y_F16) {                                                         \
        _Pragma("unroll") for (int a = 0; a < 2; ++a)                                            \
        _Pragma("unroll") for (int b = 0; b < 2; ++b)                                            \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) Y[a][b][r] = (float)(f16)Y[a][b][r];      \
    }                                                                                            \
    scale_p = token_scale(Y, out);                                                               \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                \
    _Pragma("unroll") for (int b = 0; b < 2; ++b) Yp[a][b] = Y[a][b];                            \
    tok_p = (tok_);

    auto pull = [&]() {
        int v = 0;
        if (lane == 0) v = (int)atomicAdd(next_slot, 1u);
        return __builtin_amdgcn_readfirstlane(v);
    };

    f32x16 Yp[2][2];
    float scale_p;
    int64_t tok_p;
    int nxt = pull();  // the slot whose DMA the next LOAD issues

    {   // ---- prologue: first token, nothing to quantise yet ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FQ_LOAD_X_AND_PREFETCH(nxt, nxt < blk_cnt)
        FQ_GEMMS()
        FQ_FINISH_Y(blk_base + slot)
        slot = nxt;
        nxt = pull();
    }
    bool counted = false;  // true when exactly 3 stores (scale + 2 x 16 B) were issued after the pending DMA
    while (slot < blk_cnt) {
        if (counted) {
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // all but the 3 younger stores: the DMA has landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const int64_t tok = blk_base + slot;
        uint32_t pw[2][4];
        // ---- ONE basic block: quantiser of the previous token (VALU) + transform of this token (LDS, DMA, MFMA) ----
        FQ_LOAD_X_AND_PREFETCH(nxt, nxt < blk_cnt)
        const unsigned near = quant_fast(Yp, 1.0f / scale_p, pw);
        FQ_GEMMS()
        // -------------------------------------------------------------------------------------------------------------
        quant_fix_and_store(Yp, scale_p, near, pw, qout, sout, tok_p, lane);
        counted = true;
        FQ_FINISH_Y(tok)
        slot = nxt;
        nxt = pull();
    }
    {   // ---- drain: the last token's quantiser ----
        uint32_t pw[2][4];
        const unsigned near = quant_fast(Yp, 1.0f / scale_p, pw);
        quant_fix_and_store(Yp, scale_p, near, pw, qout, sout, tok_p, lane);
    }
#undef FQ_LOAD_X_AND_PREFETCH
#undef FQ_GEMMS
#undef FQ_FINISH_Y
}

}  // namespace

int fq_launch_kron64p(const f16* x, const f16* left, const f16* right, int64_t rows, const FqQuantOut& out, int n_cu,
                      hipStream_t stream) {
    int64_t blocks = (rows + WAVES - 1) / WAVES;
    if (blocks > n_cu) blocks = n_cu;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fq_kron64p_kernel, dim3((unsigned)blocks), dim3(THREADS), 0, stream, x, left, right, rows, out);
    return (int)hipGetLastError();
}
