// fq_had_mfma.hip — the online Hadamard rotation of n = K * 512 / K * 1024 (K <= 32, K % 4 == 0: 14336 = 28 * 512, Llama-3-8B's ffn width)
// fused with the deploy Quantizer, with the STRUCTURE of the rotation used instead of two dense Kronecker factors (round 4).
//   y = hadK [K,K] @ FWHT_512( x.view(rows, K, 512) ) / sqrt(n)            hadamard_utils.py:132-141, deploy/functional/online_trans.py:144-151
//   -> deploy.nn.Quantizer (deploy/nn/quantization.py:13-36), packed INT4 + fp16 scale per token
//
// Why. As ONE Kronecker launch (fq_kron_trio.hip: 112 x 128 = (hadK (x) H_4) (x) H_128) a token costs 240 dense 32x32x16 MFMAs, and on
// this part the matrix pipe's ENERGY is what the launch pays for: every fused kernel of this library runs AT the 1400 W package cap
// (profiles/r03_power_clock.txt), 0.8 PFLOP/s of fp16 MFMA next to 3.6 TB/s of HBM traffic, and a 13 % cut of the VALU stream moved the
// three-group kernel by 1 % (profiles/r04_quant_lo_ab.txt). H_512 = H_4 (x) H_4 (x) H_32 needs no dense 128-wide contraction:
//   token x[k, a, b, c]   (k < K rows of hadK; a, b in 0..3; c in 0..31;   memory row 4 k + a of 128 columns 32 b + c)
//   1. b-butterfly (H_4 over the four 32-column blocks) on the A fragments, packed fp16 adds, scaled by nothing (the sums stay in range:
//      the activation is the fp16 input, a sum of four doubles its RMS);
//   2. GEMM 1: contraction over c with H_32 / 16 — K = 32: TWO MFMAs per row tile instead of eight;
//   3. a-butterfly (H_4 over the four ROW TILES: rows are read in the order (a, k), so a is the tile index and the butterfly is
//      element-wise across the wave's four accumulator tiles, fp32);
//   4. rounding to fp16, chained into GEMM 2's A operand in registers as in every kernel here;
//   5. GEMM 2: contraction over k with hadK (zero-padded to 32 x 32) — TWO MFMAs per output tile instead of seven;
//   6. x 16 / sqrt(n) in fp32, rounding to fp16, the Quantizer's fp16 arithmetic (fq_quant8_h16), 8-byte stores.
// 16 MFMAs per wave and token instead of 60; both B operands are 2 x 4 VGPRs generated once per wave (H_32 from the parity of c & c',
// hadK from the caller's [K, K] table): no fragment image, no L image in LDS, no workspace.
// Structure otherwise as fq_kron_trio.hip: one persistent 12-wave workgroup per CU, three token groups of four waves (wave w owns the
// output columns of block b' = w), LDS-DMA staging, meetings on LDS counters, token claims one ahead.
// n = K * 1024 (28672 = 28 * 1024, Llama-2-70B's ffn width; late round 4): the same kernel with EIGHT row tiles per token (template NA = 8):
// token x[k, a, b, c] with a in 0..7 (memory row 8 k + a), H_1024 = H_8 (x) H_4 (x) H_32 — the a-butterfly is three stages over eight
// accumulator tiles, GEMM 1's factor is H_32 / 32, 32 MFMAs per wave and token where the dense 112 x 256 pair needs 128; a token is
// 56 KB, so a CU holds TWO token groups (eight waves, 192 VGPRs).
// Rounding points differ from the FWHT route (fq_hadamard_reg.hip) and from the dense Kronecker route: parity is the reference's own
// tolerance class for this op (tests/test_gpu_hadamard.py: 2e-3 of the row maximum against matmul_hadU's fixtures), not bit identity;
// ops.hadamard_quant(fwht_route=True) keeps the bit-identical route.
#include "fq_common.hpp"
#ifndef HM_PRIO_MFMA
#define HM_PRIO_MFMA 2   // s_setprio level of a wave inside phases A / B (see fq_kron_duo.hip). 0 until the K * 1024 build: the rotation-only launches
                         // gain (28672: 220 -> 207 us, 14336: 212 -> 209), the fused Quantizer launches are even (profiles/r04_hadamard_1024.txt)
#endif

namespace {

typedef __attribute__((address_space(3))) void hm_lds_void;

#ifndef HM_ABL
#define HM_ABL 0   // measurement builds: 1 no quantiser, 2 no GEMM 1, 4 no GEMM 2, 8 no stores, 16 no DMA after the first, 32 no b-butterfly, 64 no a-butterfly
#endif
#ifndef HM_MAX3
#define HM_MAX3 1   // extrema of the rotated token with v_pk_maximum3_f16 / v_pk_minimum3_f16 (two pairs per instruction)
#endif
#ifndef HM_QSTAGE
#define HM_QSTAGE 1   // packed output staged through LDS and written as 1 KB-contiguous 16-byte stores (0: 8-byte stores straight from the registers)
#endif
// (round 5, profiles/r05_had512_variants.txt) Two variants of the copy-out of the staged packed token prepared at the end of round 4 —
// non-temporal stores, the copy-out behind phase A — were timed (130.6 / 131.9 / 132.6 us at 14336: no gain) and removed.
#ifndef HM_YSTAGE
#define HM_YSTAGE 1      // rotation-only launches: the rotated token staged through the group's own token buffer and written as 1 KB-contiguous
                         //    stores instead of 32-byte pieces straight from the registers (0). Measured round 5: 14336 205.5 -> 195.3 us,
                         //    28672 203.2 -> 197.4 us per 8192 tokens; bit-identical (tests/test_gpu_had_mfma.py, test_gpu_hadamard.py on both builds).
                         //    The next token's DMA is deferred: every wave copies out exactly the 1 KB slots its own DMA instructions refill, requests them, then stores.
#endif
#ifndef HM_NGROUPS
#define HM_NGROUPS 3   // token groups per CU (4: sixteen waves, needs <= 128 VGPRs)
#endif
// Geometry. NA = row tiles of a token = size of the a-butterfly: 4 for n = K * 512 (H_512 = H_4 (x) H_4 (x) H_32), 8 for n = K * 1024
// (H_1024 = H_8 (x) H_4 (x) H_32: 28672 = 28 * 1024, the ffn width of Llama-2-70B — a 56 KB token, TWO token groups per CU).
template <int NA, int GROUPS>
struct HmGeo {
    static constexpr int WPG = 4, THREADS = GROUPS * WPG * 64;
    static constexpr int ROWS = NA * 32;                          // LDS rows of a token buffer (rows >= NA * K stay zero)
    static constexpr int TOKBUF = ROWS * 256;                     // bytes: rows of 128 fp16
    static constexpr int LDS = GROUPS * TOKBUF + GROUPS * 32 + 64;   // + [max x4][min x4] per group + control words
    static constexpr int QROWS = NA == 8 ? 224 : 128;             // rows of the packed-output staging buffer (NA = 8: K <= 28, or two token groups
                                                                  // + staging would not fit 160 KB)
    static constexpr int QST = QROWS * 64;                        // bytes: 64 packed bytes per row
    static constexpr int KEY_SHIFT = NA == 8 ? 1 : 0;             // DMA instruction i fills rows 4 i .. 4 i + 3: swizzle key row / NA = i >> KEY_SHIFT
    static_assert(NA == 4 || NA == 8, "row tiles");
    static_assert(LDS + GROUPS * QST <= 160 * 1024, "LDS");
};

__device__ __forceinline__ unsigned hm_lds_read(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ unsigned hm_lds_add_rtn(unsigned addr, unsigned val) {
    unsigned v;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "v"(val) : "memory");
    return v;
}
__device__ __forceinline__ void hm_lds_write(unsigned addr, unsigned val) {
    asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(val) : "memory");
}
// group meeting on a counter in LDS (see fq_kron_trio.hip)
__device__ __forceinline__ void hm_meet(unsigned cnt_lds, unsigned target, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(cnt_lds), "v"(1u) : "memory");
    for (;;) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(cnt_lds) : "memory");
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)v) >= target) break;
        __builtin_amdgcn_s_sleep(1);
    }
}
#define HM_MEET() { meet_n += 4; hm_meet(meet, meet_n, lane); }

// a * sgn + c on packed fp16 pairs with sgn = (+-1, +-1): the product is exact, ONE rounding (of the sum) — a packed add or subtract
// whose sign is a wave-uniform operand instead of a branch
__device__ __forceinline__ uint32_t hm_pk_addsub(uint32_t c, uint32_t a, uint32_t sgn) {
    uint32_t d;
    asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(sgn), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 hm_pk_add32(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 hm_pk_sub32(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// QUANT: packed INT4 + scale (the Quantizer); YOUT: the rotated activation itself, fp16 (matmul_hadU_cuda's result)
// SILU: x is x_gate and the rotation's input is fp16(up * fp16(silu(x_gate))) (deploy/transformers/modeling_llama.py:277-278; fq_silu_mul8):
// a wave loads the chunks of `up` that match its own LDS-DMA instructions into registers next to them and, once both have landed,
// rewrites ITS slots of the token buffer in place — no other wave touches them before the group's next meeting.
template <int NA, int GROUPS, bool QUANT, bool YOUT, bool SILU = false>
__global__ __launch_bounds__(GROUPS * 256) void fq_had512_kernel(const f16* __restrict__ x, const f16* __restrict__ hadK, int K, int64_t rows,
                                                             int64_t tpb, float post_scale, float sig_max, float sig_min,
                                                             uint8_t* __restrict__ q_out, f16* __restrict__ scale_out,
                                                             f16* __restrict__ y_out, const f16* __restrict__ up, int rt_flags) {
    typedef HmGeo<NA, GROUPS> G;
    constexpr int HM_GROUPS = GROUPS, HM_TOKBUF = G::TOKBUF;
    constexpr bool QSTAGE = QUANT && HM_QSTAGE;
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS + (QSTAGE ? GROUPS * G::QST : 0)];
    unsigned char* qst = smem + G::LDS + (threadIdx.x >> 8) * G::QST;   // the group's staging buffer (QSTAGE only)
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;   // token group; this wave's column block b' (output columns 32 wq .. 32 wq + 31)
    unsigned char* tokbuf = smem + grp * HM_TOKBUF;
    float* red = reinterpret_cast<float*>(smem + HM_GROUPS * HM_TOKBUF) + grp * 8;   // [max x4][min x4]
    unsigned* ctl = reinterpret_cast<unsigned*>(smem + HM_GROUPS * HM_TOKBUF + HM_GROUPS * 32);   // [meet x G][next][claim x G]
    const unsigned ctl_lds = (unsigned)(size_t)(hm_lds_void*)ctl, meet = ctl_lds + grp * 4;
    const unsigned tok_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(hm_lds_void*)tokbuf);
    const int M = NA * K;                      // token rows of 256 bytes
    const int64_t tok_bytes = (int64_t)M * 256;
    const int n_dma = M >> 2;                  // 1 KB instructions per token: instruction i = rows 4 i .. 4 i + 3 (NA = 4: k = i, a = 0..3)
    const int per = (n_dma + 3) >> 2, d0 = wq * per;
    const int dn = n_dma - d0 < per ? (n_dma - d0 < 0 ? 0 : n_dma - d0) : per;
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(x);

    const int64_t blk_base = (int64_t)blockIdx.x * tpb;
    const int blk_cnt = (int)(rows - blk_base < tpb ? (rows - blk_base < 0 ? 0 : rows - blk_base) : tpb);

    // ---- once per workgroup: control words, the zero rows below the token ----
    if (tid < 16) ctl[tid] = tid == HM_GROUPS ? HM_GROUPS : 0;   // meeting counters, the next unclaimed token, (published claims)
    constexpr unsigned NEXT = HM_GROUPS * 4, CLAIM = HM_GROUPS * 4 + 4;   // byte offsets inside ctl
    for (int i = M * 16 + (tid & 255); i < G::ROWS * 16; i += 256) reinterpret_cast<uint4*>(tokbuf)[i] = make_uint4(0, 0, 0, 0);

    // ---- once per wave: the two B operands, in registers for the whole launch ----
    // GEMM 1: B1[s] lane (h, c) element j = H_32[cc = 16 s + 8 h + j][pi(c)] / 16. pi puts the 16 registers of an output lane on 16
    // consecutive output columns: accumulator row i = 8 q + 4 h' + t (q = r >> 2, t = r & 3) of GEMM 2's output is the column that
    // GEMM 1's lane c = i produced, and lane (h', .) register r shall hold column 16 h' + r: pi(8 q + 4 h' + t) = 16 h' + 4 q + t.
    // GEMM 2: B2[s] lane (h, c) element j = hadK[k' = c][kk] for the K-order of the chained A operand, kk = 16 s + 8 (j >> 2) + 4 h + (j & 3)
    // (register r = 8 s + j of the rounded accumulator holds row 8 (r >> 2) + 4 h + (r & 3)); zero beyond K.
    f16x8 B1[2], B2[2];
    {
        const int pic = 16 * ((c >> 2) & 1) + 4 * (c >> 3) + (c & 3);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int cc = 16 * s + 8 * h + j;
                // (+-1 / 16 for H_512, +-1 / 32 for H_1024: the fp16 intermediate keeps the same headroom, 32 |x|max at most)
                const f16 b1 = NA == 8 ? (f16)0.03125f : (f16)0.0625f;
                B1[s][j] = (__builtin_popcount(cc & pic) & 1) ? -b1 : b1;
                const int kk = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
                B2[s][j] = (kk < K && c < K) ? hadK[c * K + kk] : (f16)0.0f;
            }
    }
    __syncthreads();   // (the zero fill is visible before any DMA lands next to it; hadK has arrived: vmcnt(0))
#pragma unroll
    for (int s = 0; s < 2; ++s) asm volatile("" : "+v"(B1[s]), "+v"(B2[s]));

    // This wave's share of token k's DMA: instructions [d0, d0 + dn). Instruction i fills the LDS rows 4 i .. 4 i + 3 linearly; lane l
    // (row 4 i + l / 16, position l % 16) fetches the chunk that the swizzle maps there: position ^ (row / NA & 15) = l % 16 ^ (i >> KEY_SHIFT & 15)
    // — keyed on row / NA because an A fragment reads the rows NA c + a of 32 lanes c: their positions must differ with c.
    constexpr bool YSTAGE = HM_YSTAGE && YOUT && !QUANT && !SILU;
    constexpr int MAXDN = (SILU || YSTAGE) ? (NA == 8 ? 14 : 8) : 1;   // DMA instructions of a wave per token (K <= 32 | K <= 28)
    u32x4 UP[MAXDN];                                       // SILU: the matching 16-byte chunks of `up`, one per DMA instruction
    auto stage_token = [&](int k) {
        const unsigned char* src = xb + (blk_base + k) * tok_bytes;
        const unsigned lo32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)src);
        const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((size_t)src >> 32));
        const unsigned long long sb = (unsigned long long)lo32 | ((unsigned long long)hi32 << 32);
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int lb = ln & 48, lp = ln & 15;
        for (int j = 0; j < dn; ++j) {
            const int i = d0 + j;   // wave-uniform
            const unsigned rv = (unsigned)((lb + (lp ^ ((i >> G::KEY_SHIFT) & 15))) << 4);
            unsigned keep;
            asm volatile(
                "s_nop 4\n\t"
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %3\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, %2 nt\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(rv), "s"(sb + (unsigned long long)i * 1024),
                  "s"((unsigned)__builtin_amdgcn_readfirstlane((int)(tok_lds + (unsigned)i * 1024)))
                : "memory");
        }
        if (SILU) {
            const unsigned char* usrc = reinterpret_cast<const unsigned char*>(up) + (blk_base + k) * tok_bytes;
#pragma unroll
            for (int j = 0; j < MAXDN; ++j)
                if (j < dn) {
                    const int i = d0 + j;
                    const unsigned rv = (unsigned)((lb + (lp ^ ((i >> G::KEY_SHIFT) & 15))) << 4);
                    UP[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(usrc + (size_t)i * 1024 + rv));
                }
        }
    };
    // SILU: this wave's slots of the token buffer (slot 64 i + lane of DMA instruction i), gate -> fp16(up * fp16(silu(gate)))
    auto silu_prepass = [&]() {
        if (!SILU) return;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        u32x4* slots = reinterpret_cast<u32x4*>(tokbuf) + ln;
#pragma unroll
        for (int j = 0; j < MAXDN; ++j)
            if (j < dn) {
                const f16x8 g = __builtin_bit_cast(f16x8, slots[(d0 + j) * 64]);
                slots[(d0 + j) * 64] = __builtin_bit_cast(u32x4, fq_silu_mul8(g, __builtin_bit_cast(f16x8, UP[j])));
            }
    };
    if (grp < blk_cnt && dn > 0) stage_token(grp);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (grp < blk_cnt && dn > 0) silu_prepass();

    // QSTAGE: the packed token leaves through LDS. A lane holds 8 bytes of 16-byte pieces that lie 64 bytes apart in memory (a row's other
    // pieces belong to the other three waves): straight from the registers a token is 4 x NA store instructions per wave of 32 separate
    // 16-byte segments each — measured (HM_ABL builds, profiles/r04_hadamard_1024.txt) the stores cost 41 of the 130 us of a 14336 launch
    // for 20 % of its bytes. Staged: ds_write_b64 into the group's [row][4 x 16 B] image (chunk index XOR (c & 15) inside a lane row's
    // NA x 64-byte block: conflict-free), and AFTER the group's next meeting (the C|A meeting of the following iteration — no extra
    // meeting) the 256 lanes of the group copy it out linearly: every store instruction writes 1 KB of contiguous bytes.
    int64_t q_pending = -1;   // token whose packed image is waiting in the staging buffer
    auto q_copy_out = [&](int64_t ptok) {
        int tg = (int)(threadIdx.x & 255);
        asm volatile("" : "+v"(tg));
        uint8_t* qtok = q_out + ptok * ((int64_t)M * 64);
        const int n_chunks = M * 4;
        for (int g = tg; g < n_chunks; g += 256) {
            const int row = g >> 2, cc = NA == 8 ? row >> 3 : row >> 2, aa = row & (NA - 1);
            const u32x4 v = reinterpret_cast<const u32x4*>(qst)[cc * (NA * 4) + ((aa * 4 + (g & 3)) ^ (cc & 15))];
            *reinterpret_cast<u32x4*>(qtok + (int64_t)g * 16) = v;
        }
    };
    unsigned meet_n = 0;
    for (int k = grp; k < blk_cnt;) {   // k: the group's current token (of this workgroup's range), claimed one token ahead
        const int64_t tok = blk_base + k;
        HM_MEET()   // C|A: every wave of the group waited for its share of the DMA before its stores
        if (QSTAGE && q_pending >= 0 && !(HM_ABL & 8)) q_copy_out(q_pending);   // (and has written its share of the previous token's packed image)
        if (HM_PRIO_MFMA) __builtin_amdgcn_s_setprio(HM_PRIO_MFMA);

        // ===== phase A: b-butterfly on the A fragments, GEMM 1 (contraction over c, K = 32), a-butterfly, fp16 rounding =====
        f16x8 Uh[NA][2];
        {
            int cl = c, hl = h;
            asm volatile("" : "+v"(cl), "+v"(hl));   // (address arithmetic stays inside the loop)
            const uint4* tb = reinterpret_cast<const uint4*>(tokbuf) + cl * (NA * 16);   // row NA c (+ a): 16 chunks per row
            const int sw = cl & 15;
            const uint32_t sg1 = (wq & 1) ? 0xBC00BC00u : 0x3C003C00u, sg2 = (wq & 2) ? 0xBC00BC00u : 0x3C003C00u;   // packed (+-1.0h, +-1.0h)
            f32x16 U[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a) U[a] = f32x16{0};
#pragma unroll
            for (int a = 0; a < NA; ++a) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    // the four column blocks of (row tile a, K-step s): chunk 4 b + 2 s + h of row 4 c + a
                    u32x4 X[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) X[b] = __builtin_bit_cast(u32x4, tb[a * 16 + ((4 * b + 2 * s + hl) ^ sw)]);
                    u32x4 A;
                    if (HM_ABL & 32) A = X[wq];
                    else {
                        // H_4 column wq: (+ + + +), (+ - + -), (+ + - -), (+ - - +): x0 + s1 x1 + s2 (x2 + s1 x3)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t p = hm_pk_addsub(X[0][e], X[1][e], sg1), q = hm_pk_addsub(X[2][e], X[3][e], sg1);
                            A[e] = hm_pk_addsub(p, q, sg2);
                        }
                    }
                    if (!(HM_ABL & 2)) U[a] = fq_mfma32<f16>(__builtin_bit_cast(f16x8, A), B1[s], U[a]);
                }
            }
            if (!(HM_ABL & 64)) {
                // a-butterfly: U''[a'] = sum_a H_NA[a][a'] U[a], element-wise across the accumulator tiles, in place: log2(NA) stages of
                // NA / 2 packed add / subtract pairs per register pair (NA = 4: 64 v_pk_add_f32, NA = 8: 192)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    f32x2 u[NA];
#pragma unroll
                    for (int a = 0; a < NA; ++a) u[a] = f32x2{U[a][2 * j], U[a][2 * j + 1]};
#pragma unroll
                    for (int len = 1; len < NA; len <<= 1)
#pragma unroll
                        for (int a = 0; a < NA; ++a)
                            if (!(a & len)) {
                                const f32x2 p = hm_pk_add32(u[a], u[a + len]), q = hm_pk_sub32(u[a], u[a + len]);
                                u[a] = p, u[a + len] = q;
                            }
#pragma unroll
                    for (int a = 0; a < NA; ++a) U[a][2 * j] = u[a].x, U[a][2 * j + 1] = u[a].y;
                }
            }
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int j = 0; j < 8; ++j) Uh[a][p][j] = (f16)U[a][p * 8 + j];
        }

        // ===== phase B: next token's DMA, GEMM 2 (contraction over k with hadK, K = 32 per output tile), extrema =====
        if (wq == 0 && lane == 0) hm_lds_write(ctl_lds + CLAIM + grp * 4, hm_lds_add_rtn(ctl_lds + NEXT, 1u));   // claim the next token
        HM_MEET()   // A|B: the group has read its token buffer
        const int knext = __builtin_amdgcn_readfirstlane((int)hm_lds_read(ctl_lds + CLAIM + grp * 4));
        const bool more = !(HM_ABL & 16) && knext < blk_cnt && dn > 0;
        if (more && !YSTAGE) stage_token(knext);
        f32x16 Y[NA];   // Y^T of (column block wq, row tile a'): register r of lane (h, c) = column 32 wq + 16 h + r of row (a', k' = c)
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            Y[a] = f32x16{0};
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (!(HM_ABL & 4)) Y[a] = fq_mfma32<f16>(Uh[a][s], B2[s], Y[a]);
        }
        if (HM_PRIO_MFMA) __builtin_amdgcn_s_setprio(0);
        uint32_t H[NA][8];   // the fp16 pairs the deploy Quantizer sees (and the transform output)
        float vmax = 0.0f, vmin = 0.0f;
        {
            f16x2 pmax = {(f16)-INFINITY, (f16)-INFINITY}, pmin = {(f16)INFINITY, (f16)INFINITY};
            const f32x2 ps2 = {post_scale, post_scale};
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const f16x2 pr = fq_mul_to_f16x2(Y[a][2 * j], Y[a][2 * j + 1], ps2);   // fp32 product, then fp16: two roundings, as the other routes
                    const f16x2 pr1 = fq_mul_to_f16x2(Y[a][2 * j + 2], Y[a][2 * j + 3], ps2);
                    H[a][j] = __builtin_bit_cast(uint32_t, pr);
                    H[a][j + 1] = __builtin_bit_cast(uint32_t, pr1);
                    if (QUANT) {
                        if (HM_MAX3) {   // one three-operand instruction per two pairs
                            pmax = fq_pk_max3(pmax, pr, pr1);
                            pmin = fq_pk_min3(pmin, pr, pr1);
                        } else {
                            pmax = fq_pk_max(fq_pk_max(pmax, pr), pr1);
                            pmin = fq_pk_min(fq_pk_min(pmin, pr), pr1);
                        }
                    }
                }
            if (QUANT) {
                // lanes c >= K hold the zero columns of the padded hadK: 0 never moves extrema that are clamped through 0 (Quantizer)
                vmax = fq_wave_max(fmaxf((float)pmax[0], (float)pmax[1]));
                vmin = fq_wave_min(fminf((float)pmin[0], (float)pmin[1]));
                if (lane == 0) {
                    red[wq] = vmax;
                    red[4 + wq] = vmin;
                }
            }
        }

        // ===== phase C: the token's extrema, scale, quantiser, pack, stores =====
        uint2 pk[NA];
        float scale = 1.0f;
        if (QUANT) {
            HM_MEET()   // B|C: the four partial extrema are in LDS
            {
                const f32x4 r0 = *reinterpret_cast<const f32x4*>(red), r1 = *reinterpret_cast<const f32x4*>(red + 4);
                vmax = fq_uniform_f32(fmaxf(fmaxf(r0[0], r0[1]), fmaxf(r0[2], r0[3])));   // (uniform by construction; told to the compiler)
                vmin = fq_uniform_f32(fminf(fminf(r1[0], r1[1]), fminf(r1[2], r1[3])));
            }
            // rt_flags: FQ_SIG_F16 — deploy.nn.Quantizer(lac=True) with the fp16-rounded sigmoids; FQ_RATIO_POST — Quantizer(lac=False):
            // scale = fp16(max|x| / 7) * input_clip_ratio (sig_max), no zero guard (an all-zero token stores scale 0: quantization.py:30)
            scale = fq_token_scale<FQ_QUANT_F16>(vmax, vmin, sig_max, sig_min, rt_flags);
            const float inv = fq_uniform_f32(fq_fast_inv(scale));
            const bool clampq = fq_h16_needs_clamp(vmax, vmin, inv);
            const FqH16Recip rc = fq_h16_recip(scale);
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                if (HM_ABL & 1) pk[a] = uint2{H[a][0], H[a][4]};
                else if (clampq) {
                    pk[a].x = fq_quant8_h16<true>(H[a][0], H[a][1], H[a][2], H[a][3], rc);
                    pk[a].y = fq_quant8_h16<true>(H[a][4], H[a][5], H[a][6], H[a][7], rc);
                } else {
                    pk[a].x = fq_quant8_h16<false>(H[a][0], H[a][1], H[a][2], H[a][3], rc);
                    pk[a].y = fq_quant8_h16<false>(H[a][4], H[a][5], H[a][6], H[a][7], rc);
                }
            }
        }
        if (YSTAGE) {
            // the token buffer has been free since the A|B meeting (the DMA was not requested): the rotation goes into it in the token's own
            // memory layout, chunk index XOR (c & 15) as the input's (a lane's two chunks of row NA c + a': positions 4 wq + 2 h, + 1)
            int lq = lane;
            asm volatile("" : "+v"(lq));
            const int cq = lq & 31, hq = lq >> 5;
            u32x4* tb = reinterpret_cast<u32x4*>(tokbuf) + cq * (NA * 16);
            if (cq < K) {
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    tb[a * 16 + ((4 * wq + 2 * hq) ^ (cq & 15))] = u32x4{H[a][0], H[a][1], H[a][2], H[a][3]};
                    tb[a * 16 + ((4 * wq + 2 * hq + 1) ^ (cq & 15))] = u32x4{H[a][4], H[a][5], H[a][6], H[a][7]};
                }
            }
            HM_MEET()   // S: the rotated token is complete in LDS
            // slot 64 i + lane of this wave's DMA instruction i holds chunk (row 4 i + lane / 16, position lane % 16 ^ key) of the token — the
            // source offset `rv` of that instruction: read the slots, request the next token INTO them, then store (1 KB of contiguous bytes per
            // instruction); the counted wait leaves exactly those stores in flight
            const int lb = lq & 48, lp = lq & 15;
            const u32x4* slots = reinterpret_cast<const u32x4*>(tokbuf) + lq;
#pragma unroll
            for (int j = 0; j < MAXDN; ++j)
                if (j < dn) UP[j] = slots[(d0 + j) * 64];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (more) stage_token(knext);
            unsigned char* ytok = reinterpret_cast<unsigned char*>(y_out) + tok * tok_bytes;
#pragma unroll
            for (int j = 0; j < MAXDN; ++j)
                if (j < dn && !(HM_ABL & 8)) {
                    const int i = d0 + j;
                    const unsigned rv = (unsigned)((lb + (lp ^ ((i >> G::KEY_SHIFT) & 15))) << 4);
                    *reinterpret_cast<u32x4*>(ytok + (size_t)i * 1024 + rv) = UP[j];
                }
            // (the stores are younger than the DMA: vmcnt counts in issue order. A counted wait on a run-time count needs an immediate: one
            //  instruction per possible count)
            switch (dn) {
                case 0: break;
#define HM_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
                HM_W(1) HM_W(2) HM_W(3) HM_W(4) HM_W(5) HM_W(6) HM_W(7) HM_W(8) HM_W(9) HM_W(10) HM_W(11) HM_W(12) HM_W(13) HM_W(14)
#undef HM_W
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
            k = knext;
            continue;
        }
        // the DMA of the group's next token is waited for HERE, in front of the stores (which are never waited for)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (more) silu_prepass();
        if (!(HM_ABL & 8)) {
            int lq = lane;
            asm volatile("" : "+v"(lq));
            // row (a', k' = c) is memory row NA c + a'; this lane: columns 32 wq + 16 h .. + 15 of it
            if (QSTAGE) {
                const int cq = lq & 31;
                unsigned char* qs = qst + cq * (NA * 64) + (lq >> 5) * 8;
                if (cq < K) {
#pragma unroll
                    for (int a = 0; a < NA; ++a) *reinterpret_cast<uint2*>(qs + (((a * 4 + wq) ^ (cq & 15)) << 4)) = pk[a];
                }
                q_pending = tok;
                if (wq == 0 && lane == 0) scale_out[tok] = ((rt_flags & FQ_RATIO_POST) && vmax == 0.0f && vmin == 0.0f) ? (f16)0.0f : (f16)scale;
            } else if (QUANT) {
                uint8_t* qtok = q_out + tok * ((int64_t)M * 64) + wq * 16;   // 64 packed bytes per row
                const unsigned lane_off = (unsigned)((lq & 31) * (NA * 64) + (lq >> 5) * 8);
                if ((lq & 31) < K) {
#pragma unroll
                    for (int a = 0; a < NA; ++a) *reinterpret_cast<uint2*>(qtok + (lane_off + a * 64)) = pk[a];
                }
                if (wq == 0 && lane == 0) scale_out[tok] = ((rt_flags & FQ_RATIO_POST) && vmax == 0.0f && vmin == 0.0f) ? (f16)0.0f : (f16)scale;
            }
            if (YOUT) {
                unsigned char* ytok = reinterpret_cast<unsigned char*>(y_out) + tok * tok_bytes + wq * 64;   // 256 bytes per row
                const unsigned lane_off = (unsigned)((lq & 31) * (NA * 256) + (lq >> 5) * 32);
                if ((lq & 31) < K) {
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        *reinterpret_cast<u32x4*>(ytok + (lane_off + a * 256)) = u32x4{H[a][0], H[a][1], H[a][2], H[a][3]};
                        *reinterpret_cast<u32x4*>(ytok + (lane_off + a * 256 + 16)) = u32x4{H[a][4], H[a][5], H[a][6], H[a][7]};
                    }
                }
            }
        }
        k = knext;
    }
    if (QSTAGE && q_pending >= 0) {   // the group's last token
        HM_MEET()
        if (!(HM_ABL & 8)) q_copy_out(q_pending);
    }
}

// Returns -1000 when the shape is not one this kernel covers (n = K * 512 or n = K * 1024 with K <= 32, K % 4 == 0, hadK given).
// q_out / scale_out may be NULL (rotation only), y_out may be NULL (Quantizer output only). In place (y_out == x) is allowed: a token
// has landed in LDS completely (vmcnt(0) + the group's first meeting) before any of its rows is stored, and tokens do not overlap.
template <int NA, int GROUPS>
static int hm_launch(const f16* x, int64_t rows, int K, const f16* hadK, float ps, float sig_max, float sig_min, uint8_t* q_out,
                     f16* scale_out, f16* y_out, const f16* up, int rt_flags, int n_cu, hipStream_t stream) {
    int64_t blocks = (rows + GROUPS - 1) / GROUPS;
    if (blocks > n_cu) blocks = n_cu;   // one persistent workgroup per CU
    if (blocks < 1) blocks = 1;
    const int64_t tpb = (rows + blocks - 1) / blocks;
    constexpr int T = HmGeo<NA, GROUPS>::THREADS;
    if (up)
        hipLaunchKernelGGL((fq_had512_kernel<NA, GROUPS, true, false, true>), dim3((unsigned)blocks), dim3(T), 0, stream, x, hadK, K, rows, tpb,
                           ps, sig_max, sig_min, q_out, scale_out, y_out, up, rt_flags);
    else if (q_out && y_out)
        hipLaunchKernelGGL((fq_had512_kernel<NA, GROUPS, true, true>), dim3((unsigned)blocks), dim3(T), 0, stream, x, hadK, K, rows, tpb, ps,
                           sig_max, sig_min, q_out, scale_out, y_out, up, rt_flags);
    else if (q_out)
        hipLaunchKernelGGL((fq_had512_kernel<NA, GROUPS, true, false>), dim3((unsigned)blocks), dim3(T), 0, stream, x, hadK, K, rows, tpb, ps,
                           sig_max, sig_min, q_out, scale_out, y_out, up, rt_flags);
    else
        hipLaunchKernelGGL((fq_had512_kernel<NA, GROUPS, false, true>), dim3((unsigned)blocks), dim3(T), 0, stream, x, hadK, K, rows, tpb, ps,
                           sig_max, sig_min, q_out, scale_out, y_out, up, rt_flags);
    return (int)hipGetLastError();
}

}  // namespace

// up != NULL: x is x_gate, the rotation's input fp16(up * fp16(silu(x))) (packed output only)
int fq_launch_had_mfma(const f16* x, int64_t rows, int n, int K, const f16* hadK, float scale, float sig_max, float sig_min,
                       uint8_t* q_out, f16* scale_out, f16* y_out, int n_cu, hipStream_t stream, const f16* up, int plain_quantizer) {
    // plain_quantizer: deploy.nn.Quantizer(lac=False) — sig_max is its input_clip_ratio (FQ_RATIO_POST), sig_min unused
    const int rt_flags = plain_quantizer ? FQ_RATIO_POST : FQ_SIG_F16;
    if (up && (!q_out || y_out)) return -1000;
    if (K <= 1 || K > 32 || (K & 3) || hadK == nullptr || (n != K * 512 && n != K * 1024)) return -1000;
    if (!q_out && !y_out) return -1000;
    // the +-1 / 16 (+-1 / 32) right factor is undone here: y = (1 / sqrt(n)) H x = scale * 16 * (H / 16) x
    if (n == K * 512) return hm_launch<4, HM_NGROUPS>(x, rows, K, hadK, scale * 16.0f, sig_max, sig_min, q_out, scale_out, y_out, up, rt_flags, n_cu, stream);
    if (K > 28) return -1000;   // (the staging buffer of the two-group geometry holds 224 rows)
    return hm_launch<8, 2>(x, rows, K, hadK, scale * 32.0f, sig_max, sig_min, q_out, scale_out, y_out, up, rt_flags, n_cu, stream);
}
