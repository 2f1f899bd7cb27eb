// fq_kron_trio.hip — fused Kronecker transform + per-token INT4 quantisation for 64 < M <= 128, N = 128 (packed output):
// d = 14336 = 112 x 128 (Llama-3-8B ffn: the down_proj transform, deploy/kernels/kron_matmul.py:213-266 +
// functional/online_trans.py:113-122) and the Hadamard rotation of 14336 run as a Kronecker pair in front of the deploy
// Quantizer (deploy/nn/quantization.py:11-33).
//
// A token of this size needs four waves (one 32-column n'-tile each, 64 accumulator registers), and four waves that share
// a token have to meet three times per token: token staged -> token consumed -> statistics exchanged. In the
// workgroup-per-token kernel (fq_kron_generic.hip) those meetings are workgroup barriers, LDS and registers allow two
// workgroups per CU (two waves per SIMD), and the token is staged through prefetch registers and VALU address arithmetic.
// Here ONE persistent workgroup per CU holds THREE token groups of four waves (12 waves, three per SIMD), and:
//   * the four waves of a group meet on a counter in LDS (ds_add + a ds_read spin), not on s_barrier: the groups are
//     independent of each other and drift apart, so that at any time the SIMD has a wave in GEMM 1 (phase A), one in GEMM 2 +
//     extrema (phase B) and one in the quantiser (phase C) to choose from;
//   * a group claims its next token from a workgroup counter in LDS (one claim ahead), so a group the SIMD arbiter
//     favours (the oldest waves) simply processes more tokens;
//   * staging is LDS-DMA (global_load_lds_dwordx4, source-side XOR swizzle as in fq_kron_wave.hip): no prefetch registers, no
//     staging VALU; the group's next token is requested at the start of phase B, when the group has consumed its buffer,
//     and waited for in front of the stores of phase C (the counter then only holds that request; stores are never waited for);
//   * the L fragment image (32 KB) is shared by the three groups; 3 x 32 KB of token buffers: 128 KB of LDS, <= 168 VGPRs,
//     no spills (a spill's reload waits on vmcnt(0), i.e. on the stores or the DMA in flight);
//   * no output stage in LDS: a lane stores its 8 contiguous bytes (16 n') of each of its rows; the four waves of a group
//     complete a row's 64 bytes within the same phase (L2 merges them). (Measured: two contiguous 16-byte stores per wave,
//     or two row-wise ones, time the same within 2 %.)
// Same mathematics, rounding points and fragment chaining as the other Kronecker kernels; same workspace image
// (fq_kron_prepare_kernel). Everything this kernel does not take (fp16 outputs, SiLU.mul input, diag, per-128 scales)
// stays with fq_kron_generic.hip. What limits it is in DESIGN.md 4.2 (the shader clock under this load).
#include "fq_common.hpp"
#ifndef FQ_PRIO_MFMA
#define FQ_PRIO_MFMA 2   // s_setprio level of a wave inside its GEMM phases (0: off)
#endif
#include "fq_dma.hpp"

namespace {

__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

#ifndef TRIO_DA
#define TRIO_DA 8   // GEMM 1: A fragments in flight (x 4 VGPRs)
#endif
#ifndef TRIO_DB
#define TRIO_DB 2   // GEMM 2: L fragments in flight
#endif
#ifndef TRIO_ABL
#define TRIO_ABL 0  // measurement builds: 1 no quantiser, 2 no GEMM 1, 4 no GEMM 2, 8 no stores, 16 no DMA
#endif
constexpr int TRIO_KS1 = 8, TRIO_NT = 4, TRIO_CPR = 16;  // N = 128
constexpr int TRIO_GROUPS = 3, TRIO_WPG = 4, TRIO_THREADS = TRIO_GROUPS * TRIO_WPG * 64;

template <int MT>
struct TrioGeom {
    static constexpr int LFR = 2 * MT * MT * 64;                // uint4
    static constexpr int TOKBUF = MT * 32 * TRIO_CPR * 16;      // bytes, rows padded to the tile
    static constexpr int LDS = LFR * 16 + TRIO_GROUPS * TOKBUF + TRIO_GROUPS * 8 * 4 + 32;  // + [max x4, min x4] per group, 8 control words
};

#ifdef TRIO_TRACE  // measurement builds: s_memtime stamps of workgroup 0 (fq_trio_trace_read copies them out)
__device__ unsigned long long trio_trace[12 * 32 * 12];
#define TRIO_STAMP(slot)                                                                                     \
    if (blockIdx.x == TRIO_TRACE && it < 32 && lane == 0) trio_trace[(wave * 32 + it) * 12 + (slot)] = __builtin_amdgcn_s_memtime();
#else
#define TRIO_STAMP(slot)
#endif

// LDS control words are touched through explicit ds_* instructions: a generic-pointer access would be a FLAT operation,
// whose s_waitcnt vmcnt(0) drains the DMA in flight.
__device__ __forceinline__ unsigned trio_lds_read(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ unsigned trio_lds_add_rtn(unsigned addr, unsigned val) {
    unsigned v;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "v"(val) : "memory");
    return v;
}
__device__ __forceinline__ void trio_lds_write(unsigned addr, unsigned val) {
    asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(val) : "memory");
}

// Group meeting on a counter in LDS: every wave adds one when it arrives (its own LDS traffic done) and spins until the
// four arrivals of this meeting are in. LDS operations of one wave complete in order, so a wave that sees the count
// also sees what the others wrote before arriving.
__device__ __forceinline__ void trio_meet(unsigned cnt_lds, unsigned target, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(cnt_lds), "v"(1u) : "memory");
    for (;;) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(cnt_lds) : "memory");
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)v) >= target) break;
        __builtin_amdgcn_s_sleep(1);
    }
}
#define TRIO_MEET() { meet_n += 4; trio_meet(meet, meet_n, lane); }

template <int MT, bool H16, bool TAIL>
__global__ __launch_bounds__(TRIO_THREADS) void fq_kron_trio_kernel(const f16* __restrict__ x, const uint4* __restrict__ ws,
                                                                  int64_t rows, int64_t tpb, int M, FqQuantOut out) {
    typedef TrioGeom<MT> G;
    constexpr int KS1 = TRIO_KS1, NT = TRIO_NT, CPR = TRIO_CPR, N = 128;
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;  // token group, n'-tile of this wave
    uint4* lfr = reinterpret_cast<uint4*>(smem);
    unsigned char* tokbuf = smem + G::LFR * 16 + grp * G::TOKBUF;
    float* red = reinterpret_cast<float*>(smem + G::LFR * 16 + TRIO_GROUPS * G::TOKBUF) + grp * 8;  // [max x4][min x4]
    unsigned* ctl = reinterpret_cast<unsigned*>(smem + G::LFR * 16 + TRIO_GROUPS * G::TOKBUF + TRIO_GROUPS * 32);  // [meet x3][next][claim x3]
    const unsigned ctl_lds = (unsigned)(size_t)(lds_void*)ctl, meet = ctl_lds + grp * 4;
    const unsigned tok_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)tokbuf);
    const int64_t tok_bytes = (int64_t)M * (N * 2);
    const int n_dma = (M + 3) >> 2;        // 1 KB instructions per token; the last one covers M % 4 rows only when M % 4 != 0
    const int per = (n_dma + 3) >> 2;      // this wave stages instructions [d0, d0 + dn)
    const int d0 = wq * per;
    const int dn = n_dma - d0 < per ? (n_dma - d0 < 0 ? 0 : n_dma - d0) : per;
    const int tail_lanes = (M & 3) * 16;   // lanes of that last instruction that carry data (M = 86: 32)
    const bool own_tail = TAIL && dn > 0 && d0 + dn == n_dma;   // TAIL: M % 4 != 0 (its own instantiation: registers)
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(x);

    const int64_t blk_base = (int64_t)blockIdx.x * tpb;
    const int blk_cnt = (int)(rows - blk_base < tpb ? (rows - blk_base < 0 ? 0 : rows - blk_base) : tpb);

    // ---- once per workgroup: L image, zero rows below the token, this wave's R fragments, first DMA ----
    {
        const uint4* lsrc = ws + NT * KS1 * 64;
        for (int i = tid; i < G::LFR; i += TRIO_THREADS) lfr[i] = lsrc[i];
        if (tid < 8) ctl[tid] = tid == 3 ? TRIO_GROUPS : 0;  // meeting counters, the next unclaimed token, (published claims)
        for (int i = M * CPR + (tid & 255); i < MT * 32 * CPR; i += 256)
            reinterpret_cast<uint4*>(tokbuf)[i] = make_uint4(0, 0, 0, 0);
    }
    f16x8 RF[KS1];
#pragma unroll
    for (int s = 0; s < KS1; ++s) RF[s] = __builtin_bit_cast(f16x8, ws[(wq * KS1 + s) * 64 + lane]);
    unsigned rv[4];
    dma_span_offsets<CPR>(lane, d0, rv);
    __syncthreads();  // (the R fragments have arrived: vmcnt(0); the zero fill is visible before any DMA lands next to it)
#pragma unroll
    for (int s = 0; s < KS1; ++s) asm volatile("" : "+v"(RF[s]));
    // this wave's share of token k's DMA; a partial last instruction runs with the lanes beyond the token masked off
    // (they would fetch the next token's bytes into the zero rows below this one)
    auto stage_token = [&](int k, const unsigned (&rvv)[4]) {
        const unsigned char* src = xb + (blk_base + k) * tok_bytes + (int64_t)d0 * 1024;
        const int nfull = own_tail ? dn - 1 : dn;
        if (nfull > 0) dma_span(src, nfull, tok_lds + (unsigned)d0 * 1024, rvv);
        if (own_tail && lane < tail_lanes) {
            const unsigned r1[4] = {rvv[nfull & 3], 0u, 0u, 0u};
            dma_span(src + (int64_t)nfull * 1024, 1, tok_lds + (unsigned)(d0 + nfull) * 1024, r1);
        }
    };
    if (grp < blk_cnt && dn > 0) stage_token(grp, rv);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    FqGroupCursor gcur;
    const float ps = out.post_scale != 0.0f ? out.post_scale : 1.0f;

    unsigned meet_n = 0;

    int it = 0;
    for (int k = grp; k < blk_cnt; ++it) {  // k: the group's current token (of this workgroup's range), claimed one token ahead
        const int64_t tok = blk_base + k;

        // ================= phase A: GEMM 1 (U = X . R for this wave's n'-tile), fp16 rounding =================
        TRIO_STAMP(0)
#ifdef TRIO_TRACE
        if (blockIdx.x == TRIO_TRACE && it < 32 && lane == 0) trio_trace[(wave * 32 + it) * 12 + 11] = __builtin_amdgcn_s_memrealtime();
#endif
        TRIO_MEET()  // C|A: every wave of the group waited for its share of the DMA before its stores of phase C
        TRIO_STAMP(1)
        // (round 4) a wave inside its GEMM phases goes ahead of the quantising ones at the SIMD's arbiter: the matrix pipe is the scarcer
        // resource and an MFMA that waits behind another wave's VALU burst idles it (measured: 128 x 224 194.5 -> 190.8 us, 112 x 128 159.3 -> 158.0)
        if (FQ_PRIO_MFMA) __builtin_amdgcn_s_setprio(FQ_PRIO_MFMA);
        f16x8 Uh[MT][2];
        {
            int cl = c;
            asm volatile("" : "+v"(cl));  // keep the address arithmetic inside the loop (hoisted, it is spilled)
            const int foff = cl * CPR, swl = swz<CPR>(cl);
            const uint4* tb = reinterpret_cast<const uint4*>(tokbuf) + foff;
            f32x16 U[MT];
            // fragment reads run TRIO_DA - 1 MFMAs (32 cycles each) ahead of their use, flattened over (s, mt)
            constexpr int DA = TRIO_DA, NA = KS1 * MT;
            f16x8 A[DA];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) U[mt] = f32x16{0};
#pragma unroll
            for (int i = 0; i < DA - 1; ++i)
                A[i] = __builtin_bit_cast(f16x8, tb[(i % MT) * 32 * CPR + (((i / MT) * 2 + h) ^ swl)]);
#pragma unroll
            for (int i = 0; i < NA; ++i) {  // i = s * MT + mt
                if (i + DA - 1 < NA) {
                    const int i1 = i + DA - 1;
                    A[i1 % DA] = __builtin_bit_cast(f16x8, tb[(i1 % MT) * 32 * CPR + (((i1 / MT) * 2 + h) ^ swl)]);
                }
                if (!(TRIO_ABL & 2)) U[i % MT] = mfma32(A[i % DA], RF[i / MT], U[i % MT]);
                else if (TRIO_ABL & 32) asm volatile("" : : "v"(A[i % DA]), "v"(RF[i / MT]));   // (round 6: the operand reads stay)
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int j = 0; j < 8; ++j) Uh[mt][p][j] = (f16)U[mt][p * 8 + j];
        }

        // ================= phase B: next token's DMA, GEMM 2 (Y^T = U^T . L), extrema =================
        if (wq == 0 && lane == 0) trio_lds_write(ctl_lds + 16 + grp * 4, trio_lds_add_rtn(ctl_lds + 12, 1u));  // claim the group's next token; published by the meeting
        TRIO_STAMP(2)
        TRIO_MEET()  // A|B: the group has read its token buffer
        TRIO_STAMP(3)
        const int knext = __builtin_amdgcn_readfirstlane((int)trio_lds_read(ctl_lds + 16 + grp * 4));
        const bool more = !(TRIO_ABL & 16) && knext < blk_cnt && dn > 0;
        {
            // per-lane DMA offsets (dma_span_offsets for CPR = 16, in closed form): slot q = 64 i + lane -> row 4 i + lane / 16,
            // chunk lane % 16 ^ (row & 15); 4 i and lane / 16 do not overlap, so the XOR splits into a lane part (pe) and a
            // wave-uniform part. Recomputed per token: four VGPRs less across the GEMMs.
            int pe = (lane & 15) ^ (lane >> 4), lb = lane & 48;
            asm volatile("" : "+v"(pe), "+v"(lb));
#pragma unroll
            for (int j = 0; j < 4; ++j) rv[j] = (unsigned)((lb + (pe ^ ((4 * (d0 + j)) & 15))) << 4);
        }
        if (more) stage_token(knext, rv);
        TRIO_STAMP(4)
        f32x16 Y[MT];  // Y^T of tile (wq, mo): rows n' = h*64 + wq*16 + r, col m' = 32 mo + c
        {
            int loff = lane;
            asm volatile("" : "+v"(loff));
            const uint4* mylfr = lfr + loff;
            constexpr int DB = TRIO_DB, NB = 2 * MT * MT;
            f16x8 B[DB];
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) Y[mo] = f32x16{0};
#pragma unroll
            for (int i = 0; i < DB - 1; ++i) B[i] = __builtin_bit_cast(f16x8, mylfr[i * 64]);
            const int ks_n = (M + 15) >> 4;  // rows of L beyond M are zero: 112 = 7 K-steps of 16 (only the last two can be empty)
#pragma unroll
            for (int i = 0; i < NB; ++i) {  // i = ks * MT + mo
                const int ks = i / MT, mo = i % MT;
                if (i + DB - 1 < NB) B[(i + DB - 1) % DB] = __builtin_bit_cast(f16x8, mylfr[(i + DB - 1) * 64]);
                if (!(TRIO_ABL & 4) && (ks < 2 * MT - 2 || ks < ks_n)) Y[mo] = mfma32(Uh[ks >> 1][ks & 1], B[i % DB], Y[mo]);
                else if (TRIO_ABL & 32) asm volatile("" : : "v"(B[i % DB]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        TRIO_STAMP(5)
        if (FQ_PRIO_MFMA) __builtin_amdgcn_s_setprio(0);
        uint32_t H[H16 ? MT : 1][8];  // H16: the fp16 pairs the deploy Quantizer sees
        float vmax = -INFINITY, vmin = INFINITY;
        if (H16) {
            f16x2 pmax = {(f16)-INFINITY, (f16)-INFINITY}, pmin = {(f16)INFINITY, (f16)INFINITY};
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) {
                // rows of every tile but the last are all valid (M > 32 (MT - 1)): only the last tile's extrema need the mask
                f16x2 tmax = {(f16)-INFINITY, (f16)-INFINITY}, tmin = {(f16)INFINITY, (f16)INFINITY};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f16x2 pr = fq_mul_to_f16x2(Y[mo][2 * j], Y[mo][2 * j + 1], f32x2{ps, ps});
                    H[H16 ? mo : 0][j] = __builtin_bit_cast(uint32_t, pr);
                    if (mo == MT - 1) {
                        tmax = fq_pk_max(tmax, pr);
                        tmin = fq_pk_min(tmin, pr);
                    } else {
                        pmax = fq_pk_max(pmax, pr);
                        pmin = fq_pk_min(pmin, pr);
                    }
                }
                if (mo == MT - 1 && (mo * 32 + c) < M) {
                    pmax = fq_pk_max(pmax, tmax);
                    pmin = fq_pk_min(pmin, tmin);
                }
            }
            vmax = fmaxf((float)pmax[0], (float)pmax[1]);
            vmin = fminf((float)pmin[0], (float)pmin[1]);
        } else {
            if (out.post_scale != 0.0f) {
#pragma unroll
                for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float p = Y[mo][r] * ps;
                        asm volatile("" : "+v"(p));  // an fp32 VALUE (no fusion with a later rounding)
                        Y[mo][r] = p;
                    }
            }
            if (out.rt_flags & FQ_ROUND_Y_F16) {
#pragma unroll
                for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Y[mo][r] = (float)(f16)Y[mo][r];
            }
            float pmx[MT], pmn[MT];  // one independent max3 / min3 chain per tile
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) {
                const f32x16& t = Y[mo];
                float a = FqMaxOp()(t[0], t[1]), b = FqMinOp()(t[0], t[1]);
#pragma unroll
                for (int r = 2; r < 16; r += 2) {
                    a = fq_max3(a, t[r], t[r + 1]);
                    b = fq_min3(b, t[r], t[r + 1]);
                }
                const bool ok = mo < MT - 1 || (mo * 32 + c) < M;   // (only the last row tile can hold padding rows)
                pmx[mo] = ok ? a : -INFINITY;
                pmn[mo] = ok ? b : INFINITY;
            }
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) {
                vmax = fmaxf(vmax, pmx[mo]);
                vmin = fminf(vmin, pmn[mo]);
            }
        }
        vmax = fq_wave_max(vmax);
        vmin = fq_wave_min(vmin);
        if (lane == 0) {
            red[wq] = vmax;
            red[4 + wq] = vmin;
        }

        // ================= phase C: the token's extrema, scale, quantiser, pack, stores =================
        TRIO_STAMP(6)
        TRIO_MEET()  // B|C: the four partial extrema are in LDS
        TRIO_STAMP(7)
        {
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(red), r1 = *reinterpret_cast<const f32x4*>(red + 4);
            // (wave-uniform by construction; told to the compiler, or every branch on the scale is an exec-mask region: fq_kron_duo.hip)
            vmax = fq_uniform_f32(fmaxf(fmaxf(r0[0], r0[1]), fmaxf(r0[2], r0[3])));
            vmin = fq_uniform_f32(fminf(fminf(r1[0], r1[1]), fminf(r1[2], r1[3])));
        }
        bool waited = false;
        for (int ci = 0; ci < out.n_clips; ++ci) {
            float sig_max, sig_min;
            fq_token_sigs(out, ci, tok, gcur, sig_max, sig_min);
            float scale;
            if (H16) scale = fq_token_scale<FQ_QUANT_F16>(vmax, vmin, sig_max, sig_min, out.rt_flags);
            else scale = fq_token_scale<0>(vmax, vmin, sig_max, sig_min, out.rt_flags);
            const float inv = fq_uniform_f32(fq_fast_inv(scale));
            const FqH16Recip rc = H16 ? fq_h16_recip(scale) : FqH16Recip{0.0f, 0.0f};
            const bool magic = H16 || fq_magic_ok(vmax, vmin, inv);
            const bool clampq = fq_needs_clamp(vmax, vmin, inv);
            uint2 pk[MT];
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) {
                if (TRIO_ABL & 1) {
                    pk[mo] = uint2{__builtin_bit_cast(uint32_t, Y[mo][0]), __builtin_bit_cast(uint32_t, Y[mo][8])};
                } else if (H16) {
                    const uint32_t(&hv)[8] = H[H16 ? mo : 0];
                    if (clampq) {
                        pk[mo].x = fq_quant8_h16<true>(hv[0], hv[1], hv[2], hv[3], rc);
                        pk[mo].y = fq_quant8_h16<true>(hv[4], hv[5], hv[6], hv[7], rc);
                    } else {
                        pk[mo].x = fq_quant8_h16<false>(hv[0], hv[1], hv[2], hv[3], rc);
                        pk[mo].y = fq_quant8_h16<false>(hv[4], hv[5], hv[6], hv[7], rc);
                    }
                } else {
                    const f32x16& yv = Y[mo];
                    unsigned long long d0m = ~0ull, d1m = ~0ull;
                    pk[mo] = uint2{0u, 0u};
                    if (magic) {
                        const float ilo = fq_inv_lo(inv), ihi = fq_inv_hi(inv);
                        if (clampq) {
                            pk[mo].x = fq_quant8<true>(yv[0], yv[1], yv[2], yv[3], yv[4], yv[5], yv[6], yv[7], inv, ilo, ihi, d0m);
                            pk[mo].y = fq_quant8<true>(yv[8], yv[9], yv[10], yv[11], yv[12], yv[13], yv[14], yv[15], inv, ilo, ihi, d1m);
                        } else {
                            pk[mo].x = fq_quant8<false>(yv[0], yv[1], yv[2], yv[3], yv[4], yv[5], yv[6], yv[7], inv, ilo, ihi, d0m);
                            pk[mo].y = fq_quant8<false>(yv[8], yv[9], yv[10], yv[11], yv[12], yv[13], yv[14], yv[15], inv, ilo, ihi, d1m);
                        }
                    }
                    if (d0m)  // rare: an ambiguous digit somewhere in the wave -> the true division for this dword
                        pk[mo].x = fq_pack8(fq_qexact(yv[0], scale), fq_qexact(yv[1], scale), fq_qexact(yv[2], scale), fq_qexact(yv[3], scale),
                                            fq_qexact(yv[4], scale), fq_qexact(yv[5], scale), fq_qexact(yv[6], scale), fq_qexact(yv[7], scale));
                    if (d1m)
                        pk[mo].y = fq_pack8(fq_qexact(yv[8], scale), fq_qexact(yv[9], scale), fq_qexact(yv[10], scale), fq_qexact(yv[11], scale),
                                            fq_qexact(yv[12], scale), fq_qexact(yv[13], scale), fq_qexact(yv[14], scale), fq_qexact(yv[15], scale));
                }
            }
            // The DMA of the group's next token (requested at the start of phase B) is waited for HERE, in front of the
            // stores: the counter then only holds that request, and the stores are never waited for.
            TRIO_STAMP(8)
            if (!waited) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            waited = true;
            TRIO_STAMP(9)
            if (!(TRIO_ABL & 8)) {
                uint8_t* qtok = out.q[ci] + tok * ((int64_t)M * (N / 2)) + h * 32 + wq * 8;
#pragma unroll
                for (int mo = 0; mo < MT; ++mo)
                    if ((mo * 32 + c) < M) *reinterpret_cast<uint2*>(qtok + (mo * 32 + c) * (N / 2)) = pk[mo];
                if (wq == 0 && lane == 0) out.scale[ci][tok] = (f16)scale;
            }
        }
        TRIO_STAMP(10)
        k = knext;
    }
}

template <int MT, bool H16, bool TAIL>
int launch_trio_t(const f16* x, const uint4* ws, int64_t rows, int M, const FqQuantOut& out, int n_cu, hipStream_t stream) {
    typedef TrioGeom<MT> G;
    static_assert(G::LDS <= 160 * 1024, "LDS budget");
    int64_t blocks = (rows + TRIO_GROUPS - 1) / TRIO_GROUPS;
    if (blocks > n_cu) blocks = n_cu;  // one persistent workgroup per CU
    if (blocks < 1) blocks = 1;
    const int64_t tpb = (rows + blocks - 1) / blocks;
    hipLaunchKernelGGL((fq_kron_trio_kernel<MT, H16, TAIL>), dim3((unsigned)blocks), dim3(TRIO_THREADS), 0, stream, x, ws, rows, tpb,
                       M, out);
    return (int)hipGetLastError();
}

template <int MT, bool H16>
int launch_trio(const f16* x, const uint4* ws, int64_t rows, int M, const FqQuantOut& out, int n_cu, hipStream_t stream) {
    return (M & 3) ? launch_trio_t<MT, H16, true>(x, ws, rows, M, out, n_cu, stream)
                   : launch_trio_t<MT, H16, false>(x, ws, rows, M, out, n_cu, stream);
}

}  // namespace

#ifdef TRIO_TRACE
extern "C" int fq_trio_trace_read(void* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(trio_trace), sizeof(unsigned long long) * 12 * 32 * 12);
}
#endif

// Returns -1000 when the shape / output set is not one this kernel covers (the caller goes on to the generic kernels).
// ws: fragment workspace already filled by fq_kron_prepare_kernel (rfrag [4][8][64], lfrag [2MT][MT][64]).
int fq_launch_kron_trio(int flags, const f16* x, const void* ws, const f16* diag, int64_t rows, int M, int N,
                        const FqQuantOut& out, int n_cu, hipStream_t stream) {
    if (N != 128 || M <= 64 || M > 128 || diag != nullptr) return -1000;
    if (out.rt_flags & FQ_GROUP128) return -1000;
    const int ct = flags & FQ_CT_MASK;
    const bool h16 = ct == (FQ_OUT_PACKED | FQ_QUANT_F16) && (flags & FQ_ROUND_Y_F16);
    if (ct != FQ_OUT_PACKED && !h16) return -1000;
    const uint4* w = reinterpret_cast<const uint4*>(ws);
    if (M > 96) return h16 ? launch_trio<4, true>(x, w, rows, M, out, n_cu, stream) : launch_trio<4, false>(x, w, rows, M, out, n_cu, stream);
    return h16 ? launch_trio<3, true>(x, w, rows, M, out, n_cu, stream) : launch_trio<3, false>(x, w, rows, M, out, n_cu, stream);
}
