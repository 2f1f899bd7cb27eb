// fq_kron_duo.hip — fused Kronecker transform + per-token INT4 quantisation for 96 < M <= 128, N = 224 (packed output):
// d = 28672 = 128 x 224, Llama-2-70B's ffn width — the down_proj transform of BASELINE config 4
// (deploy/kernels/kron_matmul.py:213-266, the reference's split path; functional/online_trans.py:113-122).
//
// Why another kernel (round 3; profiles/r03_kron_128x224_pmc.txt). In the workgroup-per-token kernel (fq_kron_generic.hip) this
// pair is ONE 8-wave workgroup per CU (LDS: 32 KB of L fragments + 59 KB of token + stages; 213-239 VGPRs), every wave does the
// same thing at the same time — 88 MFMAs, then the extrema, then 262 VALU of quantiser — between three workgroup barriers:
// matrix pipe 31 % busy, VALU 21 %, waves parked 42 % of their cycles. Nothing is saturated; nothing overlaps.
// Here ONE persistent 8-wave workgroup per CU holds TWO independent token groups of four waves (the structure of
// fq_kron_trio.hip, two groups instead of three because a 128 x 224 token needs 56 KB of LDS):
//   * a wave owns TWO NEIGHBOURING n'-tiles (2 w and 2 w + 1; 7 tiles: the fourth wave owns one) and runs them TOGETHER: every
//     A fragment of the token and every L fragment is read from LDS once for both tiles (the workgroup-per-token kernel reads
//     each seven times per token: 616 KB of LDS traffic per token against 128 B/clk), and both Y slices (128 accumulators) stay
//     in registers: <= 256 VGPRs, two waves per SIMD;
//   * the R fragments (14 KB per tile, 98 KB for the token) cannot stay in registers next to that and do not fit the LDS next to
//     two tokens: they STREAM from the fragment image in L2 through a ring four K-steps deep, primed before the previous token's
//     stores (vmcnt counts loads and stores in issue order); the L image (32 KB) is shared by both groups in LDS;
//   * the four waves of a group meet three times per token on a counter in LDS (token staged | GEMM 1 done = buffer free |
//     extrema posted), never on s_barrier, so the groups drift apart: one quantises and stores while the other multiplies;
//   * token claims run one token ahead; behind the second meeting the group starts the LDS-DMA of its next token into the
//     buffer GEMM 1 has just finished with — rows unpadded, chunks rotated per row for conflict-free A-fragment reads, seven
//     per-lane source offsets, four instructions per M0 — one block of four instructions at a time between the phases that
//     follow (an LDS-DMA instruction costs its wave ~200 cycles of issue: tools/microbench/duo_trace.py);
//   * no output stage: a lane's two 8-byte runs (16 n' of each tile) of a row are neighbours: one 16-byte store.
// Measured (round 3, 8192 tokens of 128 x 224): 256-259 us -> 208 us (0.285 -> 0.353 of 8 TB/s). Tried and dropped on the way:
// one tile after the other (LDS-bound: 221 us), token staged through registers (spills next to 128 accumulators), the whole
// token's DMA from the fourth wave (27000 cycles per token: 354 us), DMA issue inside GEMM 2's K-loop (spills: 268 us).
// Same mathematics, rounding points, fragment chaining and workspace image (fq_kron_prepare_kernel) as the other Kronecker
// kernels; bit-identical results to the workgroup-per-token kernel (tests/test_gpu_kron_duo.py). Everything this kernel does
// not take (fp16 / bf16 outputs, SiLU.mul input, diag, fp16 quantiser, post-scale) stays with fq_kron_generic.hip.
#include <type_traits>

#include "fq_common.hpp"
#ifndef FQ_PRIO_MFMA
#define FQ_PRIO_MFMA 2   // s_setprio level of a wave inside its GEMM phases (0: off)
#endif

namespace {

#ifndef DUO_MEET_SLEEP
#define DUO_MEET_SLEEP 1   // s_sleep argument (x 64 cycles) between two polls of a meeting counter (measured: 4 -> +2 us, 12 -> +4 us)
#endif
#ifndef DUO_ABL
#define DUO_ABL 0   // measurement builds (tools/variants.sh): 1 no GEMM 1 MFMAs, 2 no GEMM 2 MFMAs, 4 no quantiser, 8 no DMA after the first; 256 (round 6): with 1 | 2, the MFMAs' operand reads (LDS, L2) STAY
#endif              // token, 16 no extrema, 32 no A-fragment reads, 64 no L-fragment reads, 128 no R stream
constexpr int DUO_KS1 = 14, DUO_NT = 7, DUO_CPR = 28, DUO_PITCH = 28, DUO_N = 224;
constexpr int DUO_GROUPS = 2, DUO_WPG = 4, DUO_GT = DUO_WPG * 64, DUO_THREADS = DUO_GROUPS * DUO_GT;

template <int MT>
struct DuoGeom {
    static constexpr int LFR = 2 * MT * MT * 64;              // uint4: the L fragment image
    static constexpr int XS = MT * 32 * DUO_PITCH;            // uint4: one group's token buffer
    static constexpr int CTL = LFR * 16 + DUO_GROUPS * XS * 16;          // byte offset of [max x4, min x4] per group + control words
    static constexpr int DMT = CTL + DUO_GROUPS * 32 + 32;                // byte offset of the DMA offset table: [7 starts][64 lanes] uint4 (7 KB)
    static constexpr int LDS = DMT + 7 * 64 * 16;
};

typedef __attribute__((address_space(3))) void duo_lds_void;
typedef u32x4 u32x4_a8 __attribute__((aligned(8)));

#ifdef DUO_TRACE  // measurement builds: s_memtime stamps of one workgroup (fq_duo_trace_read copies them out)
__device__ unsigned long long duo_trace[8 * 32 * 12];
#define DUO_STAMP(slot) \
    if (blockIdx.x == DUO_TRACE && it < 32 && lane == 0) duo_trace[(wave * 32 + it) * 12 + (slot)] = __builtin_amdgcn_s_memtime();
#else
#define DUO_STAMP(slot)
#endif

// LDS control words through explicit ds_* instructions (a generic-pointer access would be a FLAT operation whose
// s_waitcnt vmcnt(0) drains the loads in flight) — as in fq_kron_trio.hip.
__device__ __forceinline__ unsigned duo_lds_read(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ unsigned duo_lds_add_rtn(unsigned addr, unsigned val) {
    unsigned v;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "v"(val) : "memory");
    return v;
}
__device__ __forceinline__ void duo_lds_write(unsigned addr, unsigned val) {
    asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(val) : "memory");
}
// Group meeting: every wave adds one when it arrives (its own LDS traffic done) and spins until all four arrivals of this
// meeting are in. LDS operations of one wave complete in order: a wave that sees the count sees what the others wrote before.
__device__ __forceinline__ void duo_meet(unsigned cnt_lds, unsigned target, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(cnt_lds), "v"(1u) : "memory");
    for (;;) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(cnt_lds) : "memory");
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)v) >= target) break;
        __builtin_amdgcn_s_sleep(DUO_MEET_SLEEP);
    }
}
#define DUO_MEET() { meet_n += DUO_WPG; duo_meet(meet, meet_n, lane); }

// FULL: M == 32 MT (128 x 224, the shape of BASELINE config 4): every DMA block is four whole instructions and no row is padding —
// the lane-masked tail code, its scalar state (the launch spilled ~100 SGPRs to VGPR lanes) and the row tests are compiled out.
template <int MT, bool FULL, typename T = f16>
__global__ __launch_bounds__(DUO_THREADS) void fq_kron_duo_kernel(const T* __restrict__ x, const uint4* __restrict__ ws,
                                                                int64_t rows, int64_t tpb, int M_rt, FqQuantOut out) {
    const int M = FULL ? MT * 32 : M_rt;
    typedef DuoGeom<MT> G;
    typedef typename FqVec<T>::x8 X8;   // (round 4: bf16 activations too — bf16 MFMA, bf16 rounding points)
    constexpr int KS1 = DUO_KS1, NT = DUO_NT, CPR = DUO_CPR, PITCH = DUO_PITCH, N = DUO_N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;      // token group; this wave's n'-tiles are 2 wq and 2 wq + 1
    uint4* lfr = reinterpret_cast<uint4*>(smem);
    uint4* xs = lfr + G::LFR + grp * G::XS;        // [MT*32][PITCH]
    float* red = reinterpret_cast<float*>(smem + (G::LFR + DUO_GROUPS * G::XS) * 16) + grp * 8;   // [max x4][min x4]
    unsigned* ctl = reinterpret_cast<unsigned*>(smem + (G::LFR + DUO_GROUPS * G::XS) * 16 + DUO_GROUPS * 32);  // [meet x2][next][claim x2]
    unsigned* dmt = reinterpret_cast<unsigned*>(smem + G::DMT);
    const unsigned ctl_lds = (unsigned)(size_t)(duo_lds_void*)ctl, meet = ctl_lds + grp * 4;
    const int64_t d = (int64_t)M * N;

    const int64_t blk_base = (int64_t)blockIdx.x * tpb;
    const int blk_cnt = (int)(rows - blk_base < tpb ? (rows - blk_base < 0 ? 0 : rows - blk_base) : tpb);

    // ---- once per workgroup: L image, both token buffers zeroed (padding rows and the pad chunk stay zero), control words ----
    {
        const uint4* lsrc = ws + NT * KS1 * 64;
        for (int i = tid; i < G::LFR; i += DUO_THREADS) lfr[i] = lsrc[i];
        uint4* xall = lfr + G::LFR;
        for (int i = tid; i < DUO_GROUPS * G::XS; i += DUO_THREADS) xall[i] = make_uint4(0, 0, 0, 0);
        if (tid < 8) ctl[tid] = tid == 2 ? DUO_GROUPS : 0;   // meeting counters, the next unclaimed token, (published claims)
        // the per-lane source offsets of a DMA block (see below), for each of the seven residues of its first instruction mod 7
        for (int e = tid; e < 7 * 64 * 4; e += DUO_THREADS) {
            const int kk = e & 3, ln = (e >> 2) & 63, st = e >> 8;
            const int m = (st + kk) % 7;
            const int q = 64 * m + ln, r = (q * 2341) >> 16;   // q / 28 for q < 512
            int pos = q - 28 * r - ((r >> 2) & 3);
            pos += pos < 0 ? CPR : 0;
            dmt[e] = (unsigned)((r * CPR + pos - 64 * m) * 16 + 64);   // relative to the instruction's KB; + 64: never negative
        }
    }
    __syncthreads();

    // The group's next token goes HBM -> LDS by LDS-DMA (no registers: it is in flight while 128 accumulators are quantised).
    // Buffer layout: row r at 28 r chunks (no padding), its chunks ROTATED by g(r) = (r >> 2) & 3 — chunk ch sits at position
    // (ch + g) mod 28 — so that the 16 lanes of a ds_read_b128 group (rows c, same chunk) hit 16 different 16-byte bank groups
    // (28 r mod 16 alone repeats every four rows). DMA instruction i fills the 64 slots [64 i, 64 i + 64) linearly; its lane
    // fetches the chunk that the rotation maps to its slot. That per-lane source offset has period 7 in i (448 slots = 16 rows),
    // and relative to the instruction's own KB it does not depend on i / 7 at all: seven per-lane values (recomputed for the four
    // instructions of a block: seven more permanent registers do not fit), four instructions per M0 / base pair through the
    // instruction offset field (which advances the global AND the LDS address).
    // An LDS-DMA instruction costs its wave ~200 cycles of issue inside a busy phase (MI355X_MICROARCH.md: 100-185; measured here,
    // tools/microbench/duo_trace.py: a wave that issues its quarter of a token — 14 instructions — in one go stands there for ~3000
    // cycles, one wave issuing the whole token 27000). A wave's share goes out one block of four instructions at a time, at four
    // points of the token's schedule (dma_block): the cost is the same in total, but no phase waits for all of it.
    const unsigned xs_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(duo_lds_void*)xs);
    const int n_slots = M * CPR, n_full = n_slots >> 6, tail_lanes = n_slots & 63;   // (M = 128: 56 full instructions)
    auto dma_base = [&](int k) -> unsigned long long {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(x + (blk_base + k) * d) - 64;   // wave-uniform
        const unsigned lo32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)src);
        const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((size_t)src >> 32));
        return (unsigned long long)lo32 | ((unsigned long long)hi32 << 32);
    };
    // block b = instructions 4 b .. 4 b + 3 of the token at sb (dma_base)
    auto dma_block = [&](unsigned long long sb, int b) {
        const int i0 = 4 * b;
        if (i0 > n_full || (i0 == n_full && tail_lanes == 0)) return;
        // (round 4: the four offsets come from the table in LDS — one ds_read_b128 instead of ~50 VALU per block, 200 per token and wave)
        int ln = lane;
        asm volatile("" : "+v"(ln));   // (the address arithmetic stays here: hoisted, it is spilled, and a reload waits for the DMA in flight)
        const u32x4 rvv = reinterpret_cast<const u32x4*>(dmt)[(i0 % 7) * 64 + ln];
        const unsigned rv[4] = {rvv[0], rvv[1], rvv[2], rvv[3]};
        const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(xs_lds + (unsigned)i0 * 1024));
        // (the compiler's uniformity analysis loses sb and i0 through the lambdas: an "s" operand it believes divergent is
        //  handed over in VGPRs, so both halves go through v_readfirstlane explicitly)
        const unsigned long long sbi = sb + (unsigned long long)i0 * 1024;
        const unsigned long long sbu = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sbi) |
                                       ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sbi >> 32)) << 32);
        if (FULL || i0 + 4 <= n_full) {
            unsigned keep;
            asm volatile(
                "s_nop 4\n\t"
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %6\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, %5 nt\n\t"
                "global_load_lds_dwordx4 %2, %5 offset:1024 nt\n\t"
                "global_load_lds_dwordx4 %3, %5 offset:2048 nt\n\t"
                "global_load_lds_dwordx4 %4, %5 offset:3072 nt\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(rv[0]), "v"(rv[1]), "v"(rv[2]), "v"(rv[3]), "s"(sbu), "s"(m0v)
                : "memory");
        } else {   // M < 128: the last full instructions one by one, then the one that ends inside the token (prefix mask)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int i = i0 + kk;
                if (i < n_full || (i == n_full && lane < tail_lanes)) {
                    unsigned keep;
                    asm volatile(
                        "s_nop 4\n\t"
                        "s_mov_b32 %0, m0\n\t"
                        "s_mov_b32 m0, %3\n\t"
                        "s_nop 0\n\t"
                        "global_load_lds_dwordx4 %1, %2 nt\n\t"
                        "s_mov_b32 m0, %0"
                        : "=&s"(keep)
                        : "v"(rv[kk]), "s"(sbu + (unsigned long long)kk * 1024),
                          "s"((unsigned)__builtin_amdgcn_readfirstlane((int)(xs_lds + (unsigned)i * 1024)))
                        : "memory");
                }
            }
        }
    };
    auto stage_token = [&](int k) {   // the whole share at once: the first token only
        const unsigned long long sb = dma_base(k);
        for (int n = 0; n < 4; ++n) dma_block(sb, wq + DUO_WPG * n);
    };
    if (grp < blk_cnt) stage_token(grp);

    // Both tiles' R fragments (2 x 14 KB, 1 KB per tile and K-step) stream from the image in L2 through a ring of DR K-steps;
    // the ring is primed while the wave waits for its token. (The fourth wave of a group owns one tile: its second stream
    // repeats the first and is not used.)
    constexpr int DR = 4;
    const bool two = 2 * wq + 1 < NT;
    const uint4* rf0 = ws + (size_t)(2 * wq) * KS1 * 64;                      // wave-uniform bases, lane offset
    const uint4* rf1 = ws + (size_t)(two ? 2 * wq + 1 : 2 * wq) * KS1 * 64;
    X8 RB[2][DR];
#define DUO_PRIME_R()                                                         \
    {                                                                         \
        int ln_ = lane;                                                       \
        asm volatile("" : "+v"(ln_));                                         \
        _Pragma("unroll") for (int i = 0; i < DR - 1; ++i) {                  \
            RB[0][i] = __builtin_bit_cast(X8, rf0[i * 64 + ln_]);          \
            RB[1][i] = __builtin_bit_cast(X8, rf1[i * 64 + ln_]);          \
        }                                                                     \
    }
    DUO_PRIME_R()
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // Token claims run one token ahead: wave 0 of a group publishes the claim for token i + 1 before the group's first meeting
    // of token i; the group reads it behind the second meeting (GEMM 1 done, the buffer free) and starts the DMA at once.
    if (wq == 0 && lane == 0) duo_lds_write(ctl_lds + 12 + grp * 4, duo_lds_add_rtn(ctl_lds + 8, 1u));

    FqGroupCursor gcur;
    unsigned meet_n = 0;
    const int ks_n = (M + 15) >> 4;   // rows of L beyond M are zero

    int it = 0;
#ifdef DUO_TRACE
    unsigned long long store_ticks = 0;
#endif
    for (int k = grp; k < blk_cnt; ++it) {   // k: the group's current token (of this workgroup's range)
        const int64_t tok = blk_base + k;
        DUO_STAMP(0)
#ifdef DUO_TRACE
        if (blockIdx.x == DUO_TRACE && it < 32 && lane == 0) duo_trace[(wave * 32 + it) * 12 + 11] = __builtin_amdgcn_s_memrealtime();
#endif
        DUO_MEET()   // the group's token is in its buffer
        // (round 4) a wave inside its GEMM phases goes ahead of the quantising ones at the SIMD's arbiter: the matrix pipe is the scarcer
        // resource and an MFMA that waits behind another wave's VALU burst idles it (measured: 128 x 224 194.5 -> 190.8 us, 112 x 128 159.3 -> 158.0)
        if (FQ_PRIO_MFMA) __builtin_amdgcn_s_setprio(FQ_PRIO_MFMA);
        DUO_STAMP(1)

        // ===== the wave's two n'-tiles TOGETHER: every A fragment (token, LDS) and every L fragment (LDS) read once for both =====
        f32x16 Y[2][MT];   // Y^T of tile (nt = 2 wq + t, mo): rows n' = h*NT*16 + nt*16 + r, col m' = 32 mo + c
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) Y[t][mo] = f32x16{0};
        int knext = 0;
        bool more = false;
        unsigned long long sbn = 0;
        auto gemms = [&](auto two_c) {
            constexpr int TN = decltype(two_c)::value ? 2 : 1;
            X8 Uh[TN][MT][2];
            {
                int cl = c, ln = lane;
                asm volatile("" : "+v"(cl), "+v"(ln));   // keep the address arithmetic inside the loop
                const int p0 = h + ((cl >> 2) & 3);   // this lane's rotation (rows 32 mt + c: the same for every mt) + its K-half
                const uint4* tb = xs + cl * PITCH + p0;
                const int w12 = p0 >= 4 ? -CPR : 0, w13 = p0 >= 2 ? -CPR : 0;   // K-steps 12 and 13 wrap round the row for some lanes
                f32x16 U[TN][MT];
                X8 A[2][MT];
#pragma unroll
                for (int t = 0; t < TN; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) U[t][mt] = f32x16{0};
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) A[0][mt] = __builtin_bit_cast(X8, tb[mt * 32 * PITCH]);
#pragma unroll
                for (int s = 0; s < KS1; ++s) {
                    if (s + DR - 1 < KS1 && !(DUO_ABL & 128)) {
                        RB[0][(s + DR - 1) % DR] = __builtin_bit_cast(X8, rf0[(s + DR - 1) * 64 + ln]);
                        if (TN == 2) RB[1][(s + DR - 1) % DR] = __builtin_bit_cast(X8, rf1[(s + DR - 1) * 64 + ln]);
                    }
                    if (s + 1 < KS1 && !(DUO_ABL & 32)) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) A[(s + 1) & 1][mt] = __builtin_bit_cast(X8, tb[mt * 32 * PITCH + (s + 1) * 2 + (s + 1 == 12 ? w12 : s + 1 == 13 ? w13 : 0)]);
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int t = 0; t < TN; ++t)
                            if (!(DUO_ABL & 1)) U[t][mt] = fq_mfma32<T>(A[s & 1][mt], RB[t][s % DR], U[t][mt]);
                            else if (DUO_ABL & 256) asm volatile("" : : "v"(A[s & 1][mt]), "v"(RB[t][s % DR]));   // (the operand reads stay)
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int t = 0; t < TN; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
#pragma unroll
                            for (int j = 0; j < 8; ++j) Uh[t][mt][p][j] = (T)U[t][mt][p * 8 + j];
            }
            DUO_STAMP(2)
            DUO_MEET()   // every wave of the group has read the token: its buffer is free, the published claim is visible
            DUO_STAMP(3)
            knext = __builtin_amdgcn_readfirstlane((int)duo_lds_read(ctl_lds + 12 + grp * 4));
            more = knext < blk_cnt;
            // the next token's DMA: this wave's instructions (wq, wq + 4, ... : at most 15) go out two per K-step of GEMM 2 — a
            // VMEM issue that waits for room in the memory pipeline then waits under MFMAs already issued, not in front of them
            more = more && !(DUO_ABL & 8);
            sbn = more ? dma_base(knext) : 0ull;
            if (more) dma_block(sbn, wq);
            DUO_STAMP(4)
            {
                int loff = lane;
                asm volatile("" : "+v"(loff));
                const uint4* mylfr = lfr + loff;
                constexpr int NBUF = 2;
                X8 B[NBUF][MT];
                if (NBUF == 2) {
#pragma unroll
                    for (int mo = 0; mo < MT; ++mo) B[0][mo] = __builtin_bit_cast(X8, mylfr[mo * 64]);
                }
#pragma unroll
                for (int ks = 0; ks < 2 * MT; ++ks) {
                    if (!(DUO_ABL & 64) && (NBUF == 1 || ks + 1 < 2 * MT)) {
                        const int kl = NBUF == 1 ? ks : ks + 1;
#pragma unroll
                        for (int mo = 0; mo < MT; ++mo) B[kl % NBUF][mo] = __builtin_bit_cast(X8, mylfr[(kl * MT + mo) * 64]);
                    }
                    if (ks < 2 * MT - 2 || ks < ks_n) {
#pragma unroll
                        for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                            for (int t = 0; t < TN; ++t)
                                if (!(DUO_ABL & 2)) Y[t][mo] = fq_mfma32<T>(Uh[t][ks >> 1][ks & 1], B[ks % NBUF][mo], Y[t][mo]);
                                else if (DUO_ABL & 256) asm volatile("" : : "v"(B[ks % NBUF][mo]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if (two) gemms(std::true_type{});
        else gemms(std::false_type{});
        DUO_STAMP(5)
        if (FQ_PRIO_MFMA) __builtin_amdgcn_s_setprio(0);
        if (more) dma_block(sbn, wq + DUO_WPG);

        if (out.rt_flags & FQ_ROUND_Y_F16) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Y[t][mo][r] = (float)(T)Y[t][mo][r];
        }
        // ---- extrema over the valid entries (one independent max3 / min3 chain per tile) ----
        float vmax = -INFINITY, vmin = INFINITY;
        {
            float pmx[2][MT], pmn[2][MT];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int mo = 0; mo < MT; ++mo) {
                    const f32x16& tl = Y[t][mo];
                    float a = FqMaxOp()(tl[0], tl[1]), b = FqMinOp()(tl[0], tl[1]);
#pragma unroll
                    for (int r = 2; r < 16; r += 2) {
                        a = fq_max3(a, tl[r], tl[r + 1]);
                        b = fq_min3(b, tl[r], tl[r + 1]);
                    }
                    const bool ok = !(DUO_ABL & 16) && (2 * wq + t) < NT && (mo < MT - 1 || (mo * 32 + c) < M);
                    pmx[t][mo] = ok ? a : -INFINITY;
                    pmn[t][mo] = ok ? b : INFINITY;
                }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int mo = 0; mo < MT; ++mo) {
                    vmax = fmaxf(vmax, pmx[t][mo]);
                    vmin = fminf(vmin, pmn[t][mo]);
                }
        }
        vmax = fq_wave_max(vmax);
        vmin = fq_wave_min(vmin);
        if (lane == 0) {
            red[wq] = vmax;
            red[4 + wq] = vmin;
        }
        if (more) dma_block(sbn, wq + 2 * DUO_WPG);
        DUO_STAMP(6)
        DUO_MEET()   // the four partial extrema are posted
        DUO_STAMP(7)
        if (more) dma_block(sbn, wq + 3 * DUO_WPG);
        {
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(red), r1 = *reinterpret_cast<const f32x4*>(red + 4);
            // (wave-uniform BY CONSTRUCTION, but values out of LDS are divergent to the compiler: every branch on the scale below became
            //  an exec-mask region, the quantiser's SGPR masks were copied into VGPRs to be tested, and ~100 SGPRs were spilled to lanes)
            vmax = fq_uniform_f32(fmaxf(fmaxf(r0[0], r0[1]), fmaxf(r0[2], r0[3])));
            vmin = fq_uniform_f32(fminf(fminf(r1[0], r1[1]), fminf(r1[2], r1[3])));
        }
        if (wq == 0 && lane == 0) duo_lds_write(ctl_lds + 12 + grp * 4, duo_lds_add_rtn(ctl_lds + 8, 1u));  // the claim after next (everyone has read this one)
        // The ring is primed BEFORE this token's stores: vmcnt counts loads and stores in issue order, so a load behind the
        // stores would make the next token's first MFMA wait for their acknowledgements.
        DUO_PRIME_R()
        for (int ci = 0; ci < (DUO_ABL & 4 ? 0 : out.n_clips); ++ci) {
            float sig_max, sig_min;
            fq_token_sigs(out, ci, tok, gcur, sig_max, sig_min);
            const float scale = fq_token_scale<0, T>(vmax, vmin, sig_max, sig_min, out.rt_flags);
            const float inv = fq_uniform_f32(fq_fast_inv(scale));   // (an inline-asm VGPR result is divergent to the compiler: see above)
            const bool magic = fq_magic_ok(vmax, vmin, inv), clampq = fq_needs_clamp(vmax, vmin, inv);
            const float ilo = fq_inv_lo(inv), ihi = fq_inv_hi(inv);
            // the wave's tiles are neighbours: a lane's two 8-byte runs (16 n' each) of a row are 16 contiguous bytes
            uint8_t* qtok = out.q[ci] + tok * (d >> 1) + wq * 16;   // wave-uniform; the lane's part is a 32-bit offset recomputed here
            int lq = lane;                                            // (kept live across the GEMMs it is spilled: a reload waits for the DMA)
            asm volatile("" : "+v"(lq));
            const unsigned lane_off = (unsigned)((lq >> 5) * (NT * 8) + (lq & 31) * (N / 2));
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) {
                uint2 pk[2] = {{0u, 0u}, {0u, 0u}};
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (t == 1 && !two) continue;
                    const f32x16& yv = Y[t][mo];
                    unsigned long long d0m = ~0ull, d1m = ~0ull;
                    if (magic) {
                        if (clampq) {
                            pk[t].x = fq_quant8<true>(yv[0], yv[1], yv[2], yv[3], yv[4], yv[5], yv[6], yv[7], inv, ilo, ihi, d0m);
                            pk[t].y = fq_quant8<true>(yv[8], yv[9], yv[10], yv[11], yv[12], yv[13], yv[14], yv[15], inv, ilo, ihi, d1m);
                        } else {
                            pk[t].x = fq_quant8<false>(yv[0], yv[1], yv[2], yv[3], yv[4], yv[5], yv[6], yv[7], inv, ilo, ihi, d0m);
                            pk[t].y = fq_quant8<false>(yv[8], yv[9], yv[10], yv[11], yv[12], yv[13], yv[14], yv[15], inv, ilo, ihi, d1m);
                        }
                    }
                    if (d0m)   // rare: an ambiguous digit somewhere in the wave -> the true division for this dword
                        pk[t].x = fq_pack8(fq_qexact(yv[0], scale), fq_qexact(yv[1], scale), fq_qexact(yv[2], scale), fq_qexact(yv[3], scale),
                                           fq_qexact(yv[4], scale), fq_qexact(yv[5], scale), fq_qexact(yv[6], scale), fq_qexact(yv[7], scale));
                    if (d1m)
                        pk[t].y = fq_pack8(fq_qexact(yv[8], scale), fq_qexact(yv[9], scale), fq_qexact(yv[10], scale), fq_qexact(yv[11], scale),
                                           fq_qexact(yv[12], scale), fq_qexact(yv[13], scale), fq_qexact(yv[14], scale), fq_qexact(yv[15], scale));
                }
#ifdef DUO_TRACE   // (round 4) ticks spent ISSUING the four 16-byte stores of a token (slot 10): part of the quantiser phase's time
                unsigned long long ts0 = 0;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts0) : : "memory");
#endif
                if ((mo * 32 + (lq & 31)) < M) {
                    uint8_t* dst = qtok + (mo * 32 * (N / 2) + lane_off);
                    if (two) *reinterpret_cast<u32x4_a8*>(dst) = u32x4{pk[0].x, pk[0].y, pk[1].x, pk[1].y};   // (8-byte aligned: h * 56)
                    else *reinterpret_cast<uint2*>(dst) = pk[0];
                }
#ifdef DUO_TRACE
                unsigned long long ts1 = 0;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts1) : : "memory");
                store_ticks += ts1 - ts0;
#endif
            }
            if (wq == 0 && lane == 0) reinterpret_cast<T*>(out.scale[ci])[tok] = (T)scale;
        }
        DUO_STAMP(8)
#ifdef DUO_TRACE
        if (blockIdx.x == DUO_TRACE && it < 32 && lane == 0) duo_trace[(wave * 32 + it) * 12 + 10] = store_ticks;
        store_ticks = 0;
#endif
        // this wave's share of the next token (and the ring) has landed: everything but the (at least) 4 stores behind them
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#pragma unroll
        for (int i = 0; i < DR - 1; ++i) asm volatile("" : "+v"(RB[0][i]), "+v"(RB[1][i]));   // (no compiler wait on the ring after this)
        DUO_STAMP(9)
        k = knext;
    }
}

template <int MT, bool FULL, typename T = f16>
int launch_duo(const T* x, const uint4* ws, int64_t rows, int M, const FqQuantOut& out, int n_cu, hipStream_t stream) {
    typedef DuoGeom<MT> G;
    static_assert(G::LDS <= 160 * 1024, "LDS budget");
    auto kern = fq_kron_duo_kernel<MT, FULL, T>;
    FQ_RAISE_LDS_CAP(kern, 160 * 1024);
    int64_t blocks = (rows + DUO_GROUPS - 1) / DUO_GROUPS;
    if (blocks > n_cu) blocks = n_cu;   // one persistent workgroup per CU
    if (blocks < 1) blocks = 1;
    const int64_t tpb = (rows + blocks - 1) / blocks;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(DUO_THREADS), G::LDS, stream, x, ws, rows, tpb, M, out);
    return (int)hipGetLastError();
}

}  // namespace

#ifdef DUO_TRACE
extern "C" int fq_duo_trace_read(void* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(duo_trace), sizeof(unsigned long long) * 8 * 32 * 12);
}
#endif

// Returns -1000 when the shape / output set is not one this kernel covers (the caller goes on to the generic kernels).
// ws: fragment workspace already filled by fq_kron_prepare_kernel (rfrag [7][14][64], lfrag [2MT][MT][64]).
int fq_launch_kron_duo(int flags, const f16* x, const void* ws, const f16* diag, int64_t rows, int M, int N,
                       const FqQuantOut& out, int n_cu, hipStream_t stream) {
    if (N != DUO_N || M <= 96 || M > 128 || diag != nullptr) return -1000;
    if ((out.rt_flags & FQ_GROUP128) || out.post_scale != 0.0f) return -1000;
    const bool b = (flags & FQ_DT_BF16) != 0;   // (round 4) bf16 activations and factors: the same kernel on bf16 MFMA
    flags &= ~FQ_DT_BF16;
    if ((flags & FQ_CT_MASK) != FQ_OUT_PACKED) return -1000;
    const uint4* w = reinterpret_cast<const uint4*>(ws);
    if (b) return M == 128 ? launch_duo<4, true, bf16>((const bf16*)x, w, rows, M, out, n_cu, stream)
                           : launch_duo<4, false, bf16>((const bf16*)x, w, rows, M, out, n_cu, stream);
    if (M == 128) return launch_duo<4, true>(x, w, rows, M, out, n_cu, stream);
    return launch_duo<4, false>(x, w, rows, M, out, n_cu, stream);
}
