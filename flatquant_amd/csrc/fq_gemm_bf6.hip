// fq_gemm_bf6.hip — the INT4 x INT4 GEMM of Linear4bit on the FP6 matrix path of gfx950 ("explore
// FP4/FP6 block-scaled MFMA"), bit-identical to fq_gemm_i4.hip.
//
// gfx950 has no INT4 MFMA. fq_gemm_i4.hip widens nibbles to bytes for v_mfma_i32_32x32x32_i8. But every integer in
// [-8, 7] is exactly representable in BF6 (E3M2: 0, +-1, 2, 3, 4, 5, 6, 7, 8), and
// v_mfma_scale_f32_32x32x64_f8f6f4 with both operands BF6 and unit block scales (E8M0 = 127) multiplies them exactly
// and accumulates in fp32: |product| <= 64 and K <= 2^18 keep every partial sum an integer below 2^24, i.e. the fp32
// accumulator IS the int32 result. Measured on MI355X with register-resident operands (tools/microbench/bf6_probe.hip):
// 5.1 Pop/s for BF6 32x32x64 against 3.6 Pop/s for i8 32x32x32 under the same sustained load, and no unpack VALU.
//
// Operands are pre-arranged as "blobs" (fq_i4_to_bf6_kernel): for a tile of 32 rows and a block of 64 k, the 64 lanes'
// MFMA operand registers — lane (kh, r) holds the 32 six-bit codes of row r, k = 64 kb + 32 kh .. + 31 as a 24-byte
// little-endian bit string — stored as three planes of 64 x 8 bytes (1536 bytes per blob, blobs of a row tile
// consecutive in k). A stage of the K loop (128 k) is then whole contiguous 3 KB segments: LDS-DMA with no address
// arithmetic, fragment reads are 3 conflict-free ds_read_b64 (consecutive lanes, consecutive 8 bytes), no swizzle.
// Weights are converted once per layer (their rows in the permuted order that leaves a lane with 16 consecutive n,
// as in fq_gemm_i4.hip); activations by one small launch per call (0.5 -> 0.75 bytes per element).
//
// Tiling: 256 x 256 per 8-wave workgroup (128 x 256 / 4 waves for launches with few tiles), wave tile 128 tokens x 64 features
// (8 accumulator tiles: 6 fragments per 8 MFMAs), three LDS stages of 48 KB filled by LDS-DMA, counted vmcnt, one barrier per
// 128-k stage. 16384 x 4096 x 4096: 159 us = 3.45 Pop/s, K = 14336: 4.3 Pop/s (profiles/r03_gemm_bf6_pipeline.txt).
//   * EXPLICIT SOFTWARE PIPELINE. A wave issues in order: a run of 18 fragment reads (or 6 LDS-DMA instructions) in front of
//     a block's MFMAs keeps the matrix pipe idle while the LDS queue — shared by the eight waves, all at the same point behind
//     the barrier — takes them. Step i of a block issues MFMA i, then one DMA instruction of the stage three ahead, then the
//     three reads of fragment i of the NEXT block. The loop is split by hand (stages that refill | the last STAGES - 1 | the
//     last) so that a stage is straight-line code between two s_barrier's.
//   * -mllvm -disable-machine-sink for this file (Makefile): LLVM's MachineSink had moved the first half-stage's eight MFMAs
//     across the barrier into the join block behind an `if (s + 1 < nk)` (legal: nothing in between reads the accumulators) —
//     all sixteen then sat BEHIND the barrier, the DMA issue and the next reads, and the fragment reads in front of the barrier
//     had nothing to hide behind (found in the ISA; it also cost 18 v_mov_b64 per stage).
//   * PERSISTENT WORKGROUPS, one per CU, walking their XCD's tile sequence: the next tile's first three stages are requested
//     BEFORE the epilogue of the current one; the tile boundary waits with a counted vmcnt that leaves the epilogue's stores
//     in flight.
//   * the sym_dequant epilogue in the float pipeline (fq_gemm_common.hpp, dequant16f): 5 - 6 VALU per output element, not 11.
// Round 6 (profiles/r06_gemm_epilogue_experiments.txt): the epilogue is 19 of 147 us (stores 12.5, arithmetic 6.7), the K loop 128; holding part
// of a tile's output under the next K loop, a start-time stagger of the workgroups, wider or single fragment reads, four waves of 128 x 128 with
// 256 AGPR accumulators (one wave per SIMD, 0.5 fragment reads per MFMA: 151.8 us), a rotated loop with C = 0 first blocks: all bit-exact, none faster.
// Built, measured and dropped (docs/DESIGN_LOG.md 10): E2M3 operands, v_mfma_scale_f32_16x16x128_f8f6f4 with its own operand image,
// 128-token tiles with two stages and two workgroups per CU, non-temporal output stores, waves of 256 x 64 / 64 x 64.
#include "fq_gemm_common.hpp"
#include <type_traits>

namespace {

using namespace fqgemm;

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_b;

constexpr int BN256 = 256;                     // feature width of the prefill tiles (the template's default BN)
constexpr int BLOB = 1536;                     // bytes: 32 rows x 64 k of BF6
constexpr int SEG = 2 * BLOB;                  // a row tile's two blobs of one stage (128 k)
constexpr int TMT = 4;                         // token tiles per wave: the wave tile is 128 tokens x 64 features
// Geometry of a workgroup tile of BM tokens x 256 features: BM / 128 x 4 waves of 128 x 64.
//   BM = 256: 8 waves, two per SIMD, 48 KB per stage — the prefill shape;
//   BM = 128: 4 waves, 36 KB per stage — twice the tiles for launches that would leave CUs without one (2048 tokens x 4096
//             features are 128 tiles of 256 x 256 on 256 CUs).
//   BM = 128, BN = 128 (round 6): 2 waves, 24 KB per stage, two workgroups per CU — DECODE-sized calls of 33 .. 128 rows, where the launch is
//             the weight stream and what counts is workgroups: a 128 x 256 tile puts 16 workgroups on a 4096-wide projection, 112 on up + gate.
template <int BM, int BN = BN256>
struct Geo {
    static constexpr int NWM = BM / 128;                      // waves along the token dimension
    static constexpr int NWN = BN / 64;                       // ... along the feature dimension (64 features each)
    static constexpr int GW = NWN * NWM, GT = GW * 64;
    static constexpr int OPB = (BN / 32) * SEG;               // the weight operand's share of a stage
    static constexpr int WDMA = (BN / 32) * 3;                // ... in 1 KB DMA instructions
    static constexpr int TILE_BYTES = OPB + (BM / 32) * SEG;  // [W tiles 0..BN/32-1][X tiles 0..BM/32-1]
    static constexpr int DPW = (TILE_BYTES / 1024) / GW;      // DMA instructions per wave and stage (6 | 9 | 12)
    static constexpr int STAGES = 3;                          // LDS stages (measured: 128-token tiles with two stages and two workgroups
                                                              // per CU, forced onto 16384 x 4096 x 4096: 210 us against 157)
    static_assert(DPW * GW * 1024 == TILE_BYTES, "whole DMA instructions per wave");
};

// ---- INT4 nibbles -> BF6 blobs ----------------------------------------------------------------------------------
// E3M2 codes of 0..8; a negative value sets bit 5
__device__ __forceinline__ unsigned bf6_code(int v) {  // v in [-8, 7]
    const unsigned mag = (unsigned)(v < 0 ? -v : v);
    const unsigned tab[9] = {0x00, 0x0C, 0x10, 0x12, 0x14, 0x15, 0x16, 0x17, 0x18};
    return tab[mag] | (v < 0 ? 0x20u : 0u);
}

// Up to four sources of the same [rows, Kb] shape in one launch (round 4: the activations of q / k / v, or of gate / up, each quantised with
// its own clip factors — at 2048 tokens a conversion launch is ~7 us of which the data are 2): their images are laid one behind the other in dst.
struct ConvSrc { const uint8_t* s[4]; };
__global__ __launch_bounds__(256) void fq_i4_to_bf6_kernel(ConvSrc srcs, int nsrc, int64_t rows, int Kb, int perm,
                                                           uint8_t* __restrict__ dst) {
    __shared__ unsigned short lut[256];  // packed byte (two nibbles, even k low) -> 12 bits of codes
    {
        const int b = threadIdx.x;
        const int lo = ((b & 15) ^ 8) - 8, hi = (((b >> 4) & 15) ^ 8) - 8;
        lut[b] = (unsigned short)(bf6_code(lo) | (bf6_code(hi) << 6));
    }
    __syncthreads();
    const int KB = Kb / 32;  // blobs per row tile (64 k = 32 packed bytes)
    const int KBP = (KB + 1) / 2;   // pairs of neighbouring blobs: a thread converts one row of one blob (both K-halves: 32 packed bytes),
                                    // lanes 0-31 the rows of blob 2 j, lanes 32-63 the same rows of blob 2 j + 1 — a wave reads 64
                                    // contiguous bytes of each of 32 rows (one blob per wave read 32 bytes per row: 16384 x 14336 78.5 -> 68.6 us)
    const int64_t n_rt = (rows + 31) / 32;
    const int64_t total = nsrc * n_rt * KBP * 64;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const int64_t pair = i >> 6;
        const int64_t rtg = pair / KBP;             // row tile in dst (all sources)
        const int sp = (int)(rtg / n_rt);           // (wave-uniform: a wave is one pair)
        const int64_t rt = rtg - sp * n_rt;         // row tile of its source
        const uint8_t* __restrict__ src = sp == 0 ? srcs.s[0] : sp == 1 ? srcs.s[1] : sp == 2 ? srcs.s[2] : srcs.s[3];
        const int kb = 2 * (int)(pair - rtg * KBP) + (lane >> 5);
        if (kb >= KB) continue;
        const int r = lane & 31;
        const int64_t row = rt * 32 + (perm ? prow(r) : r);
        const bool ok = row < rows;
        uint4 in2[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};   // (rows beyond the end are zeros: never read when !ok)
        if (ok) {
            const uint4* sp = reinterpret_cast<const uint4*>(src + row * Kb + kb * 32);
            in2[0] = sp[0];
            in2[1] = sp[1];
        }
        unsigned long long* d = reinterpret_cast<unsigned long long*>(dst + (rtg * KB + kb) * BLOB) + r;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const unsigned w[4] = {in2[kh].x, in2[kh].y, in2[kh].z, in2[kh].w};
            unsigned long long g[4];  // 48 bits each: the 8 codes of one input dword
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned long long t0 = lut[w[j] & 255], t1 = lut[(w[j] >> 8) & 255], t2 = lut[(w[j] >> 16) & 255],
                                         t3 = lut[w[j] >> 24];
                g[j] = ok ? (t0 | (t1 << 12) | (t2 << 24) | (t3 << 36)) : 0ull;
            }
            d[kh * 32] = g[0] | (g[1] << 48);
            d[kh * 32 + 64] = (g[1] >> 16) | (g[2] << 32);
            d[kh * 32 + 128] = (g[2] >> 32) | (g[3] << 16);
        }
    }
}

// ---- the GEMM ---------------------------------------------------------------------------------------------------
typedef int i32x6 __attribute__((ext_vector_type(6)));

// MULTI (round 4): up to four problems that share M and K — q / k / v, or up / gate of one layer: their own quantised activations, weights,
// scales and outputs (deploy/nn/linear.py:40-54 once per projection) — as ONE launch: problem p has tn0[p + 1] - tn0[p] column tiles, every XCD walks its share of problem 0's tiles, then of
// problem 1's, ... (next_tile), and a tile picks its problem's pointers. 2048
// tokens x 4096 features are 128 tiles of 256 x 256 on 256 CUs: three such launches leave half the chip idle three times.
template <bool MULTI> struct GemmMultiArg {};
template <> struct GemmMultiArg<true> {
    int n, tn0[5], N[4];
    const uint8_t* xb[4];
    const uint8_t* wb[4];
    GemmOut out[4];
};

// (round 4 built, measured and round 5 removed a MODE 2: x_up * silu(x_gate) in the epilogue of the gate / up pair — bit-identical to the
//  GEMMs + fq_silu_mul_f16, and no faster: the up tile's epilogue reads the gate tile's values back (~100 us per 16384 x 14336 launch) and the
//  SiLU arithmetic sits where no MFMA overlaps it; 1401 us fused against 1397 separate, a loss at 2048 tokens — profiles/r04_gate_up_epilogue.txt)
// Measurement knob (tools/variants.sh builds overlay objects with it; the product build leaves it 0) — the ablations behind
// profiles/r06_gemm_epilogue_experiments.txt: 1 the epilogue's arithmetic without its stores; 2 no epilogue; 3 no LDS reads of the weight
// fragments; 4 = 3 + no LDS-DMA of the weight operand; 5 no fragment reads at all; 6 no MFMAs (the fragments are consumed by an empty asm);
// 9 (timing only, wrong data): the token operand's LDS-DMA GATHERS 16-byte units from a token-major image (32 rows x 2 units per instruction) —
// what a transform kernel could write without a 32-token transpose (FQ_OUT_BF6): 154 -> 190 us, the address path of the gather costs twice
// what the conversion launch it would save does (profiles/r06_gemm_epilogue_experiments.txt, 5).
#ifndef GEMM_ABL
#define GEMM_ABL 0
#endif

template <int BM, int MODE = 0, int BN = BN256>
__global__ __launch_bounds__((Geo<BM, BN>::GT), 2) void fq_gemm_bf6_kernel(const uint8_t* __restrict__ XB_, const uint8_t* __restrict__ WB_,
                                                                              int M, int N_, int KB, int n_vblocks, GemmOut out_,
                                                                              GemmMultiArg<(MODE != 0)> mp) {
    constexpr bool MULTI = MODE != 0;
    const uint8_t* XB = XB_;
    const uint8_t* WB = WB_;
    int N = N_;
    GemmOut out = out_;
    // (wave-uniform selects: a dynamic index into a by-value kernel argument could send it through scratch)
#define FQ_PICK(arr, p) ((p) == 0 ? mp.arr[0] : (p) == 1 ? mp.arr[1] : (p) == 2 ? mp.arr[2] : mp.arr[3])
    using G = Geo<BM, BN>;
    constexpr int NWM = G::NWM, TILE_BYTES = G::TILE_BYTES, DPW = G::DPW, STAGES = G::STAGES, OPB = G::OPB, WDMA = G::WDMA;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % NWM, wn = wave / NWM;  // wave tile: tokens (32 TMT) wm .., features 64 wn ..
    int TNg_ = (N + BN - 1) / BN;
    if constexpr (MULTI) TNg_ = mp.tn0[4];   // (slots behind the last problem hold the total)
    const int TMg = (M + BM - 1) / BM, TNg = TNg_;
    const int nk = KB / 2;  // stages of 128 k
    const bool may_clamp = KB > 10176 / 64;   // |q| <= 64 K: beyond K = 10176 the epilogue's clamp to +-65176 (x 10) can bind
    const int mt_last = (M + 31) / 32 - 1;

    // DMA plan: instruction i = DPW wave + j: the first WDMA (24 for the 256-wide tile) the weight row tiles, then the token row tiles;
    // row tile i / 3 (resp. (i - WDMA) / 3), 1 KB part i % 3 of its 3 KB.
    // The source of an instruction is wave-uniform (SGPR base) + 16 * lane: twelve address VGPRs less than per-lane
    // pointers — those had pushed the kernel into a spill whose reload sat between the DMA instructions of a stage
    // behind an s_waitcnt vmcnt(0), i.e. every stage waited for its own loads (found in the ISA, cost ~2x).
    // (round 6) one base per ROW TILE, its three 1 KB parts through the instruction's immediate offset: DPW / 3 SGPR pairs instead of DPW
    // (12 and 18 instructions per wave in the narrow decode tiles: the pointers no longer fit the scalar file otherwise)
    static_assert(DPW % 3 == 0, "a wave's DMA share is whole row tiles");
    const unsigned char* gbase[DPW / 3];
    auto plan = [&](int mb, int nb, int p) {
        const uint8_t* WBp = WB;
        const uint8_t* XBp = XB;
        int Np = N;
        if constexpr (MULTI) {
            WBp = FQ_PICK(wb, p);
            XBp = FQ_PICK(xb, p);
            Np = FQ_PICK(N, p);
        }
        const int nt_last = (Np + 31) / 32 - 1;
#pragma unroll
        for (int j = 0; j < DPW / 3; ++j) {
            const int i = wave * DPW + 3 * j;
            const int op = i < WDMA ? 0 : 1, t = (i < WDMA ? i : i - WDMA) / 3;
            int rt = (op == 0 ? nb * BN : mb * BM) / 32 + t;
            const int last = op == 0 ? nt_last : mt_last;
            rt = rt < last ? rt : last;  // tiles beyond the matrix re-read its last tile (their outputs are never stored)
            gbase[j] = (op == 0 ? WBp : XBp) + (int64_t)rt * KB * BLOB;
        }
    };
    const unsigned voff = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)(size_t)(lds_void_b*)smem;
    auto issue_one = [&](int s, int j) {   // instruction j of this wave's share of stage s
        if (GEMM_ABL == 4 && wave * DPW + j < WDMA) return;
        // (the instruction's immediate offset moves BOTH addresses: the LDS base in M0 is the row tile's, part j % 3 comes from the offset)
        const unsigned dst = lds0 + (unsigned)((s % STAGES) * TILE_BYTES) + (unsigned)(wave * DPW + j - j % 3) * 1024u;
        const unsigned char* src = gbase[j / 3] + (int64_t)s * SEG;
        if (GEMM_ABL == 9) {
            const size_t s_ = (size_t)src;
            src = reinterpret_cast<const unsigned char*>(
                (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)s_) | ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(s_ >> 32)) << 32));
        }
        unsigned keep;
        if (GEMM_ABL == 9 && wave * DPW + j >= WDMA) {   // timing only: the token operand GATHERED from a token-major image (32 rows x 2 units per instruction)
            const unsigned vx = (unsigned)(lane & 31) * (unsigned)(KB * 48) + (unsigned)(lane >> 5) * 16u;
            const size_t sx_ = (size_t)(gbase[j / 3] + (int64_t)s * 96 + (j % 3) * 32);
            const unsigned char* srcx = reinterpret_cast<const unsigned char*>(
                (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sx_) | ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sx_ >> 32)) << 32));
            const unsigned dstx = dst + (unsigned)(j % 3) * 1024u;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vx), "s"(srcx), "s"(__builtin_amdgcn_readfirstlane((int)dstx)) : "memory");
            return;
        }
#define FQ_GLDS(OFF)                                                                                                  \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" OFF "\n\ts_mov_b32 m0, %0"        \
                     : "=&s"(keep) : "v"(voff), "s"(src), "s"(__builtin_amdgcn_readfirstlane((int)dst)) : "memory")
        if (j % 3 == 0) FQ_GLDS("");
        else if (j % 3 == 1) FQ_GLDS(" offset:1024");
        else FQ_GLDS(" offset:2048");
#undef FQ_GLDS
    };
    auto request_first_stages = [&]() {
#pragma unroll
        for (int p = 0; p < STAGES; ++p)
            if (p < nk) {
#pragma unroll
                for (int j = 0; j < DPW; ++j) issue_one(p, j);
            }
    };
    // the tile sequence of this workgroup: virtual blocks blockIdx.x, + gridDim.x, ... (gridDim.x % 8 == 0: the XCD stays)
    // MULTI (MODE 1): an XCD's share of the sequence is its share of problem 0's tiles, then of problem 1's, ...: all eight XCDs work on
    // the same problem at the same time, like a launch of its own would. Side by side in ONE sequence (first build) XCDs 0-3 streamed the
    // activations of gate_proj while 4-7 streamed those of up_proj: 16384 x 14336 x 4096 with each projection on its own packed input
    // 1237 us against 2 x 527 for two launches (tools/time_gate_up.py).
    auto next_tile = [&](int& vb, int& mb, int& nb, int& p) -> bool {
        if constexpr (MODE == 1) {
            for (; vb < n_vblocks; vb += (int)gridDim.x) {
                const int xcd = vb & 7;
                int local = vb >> 3;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int tnq = mp.tn0[q + 1] - mp.tn0[q], Tq = TMg * tnq, per = (Tq + 7) >> 3;   // (problems beyond mp.n: zero tiles)
                    if (local < per) {
                        const int L = xcd * per + local;
                        if (L < Tq) {
                            const int blk = L / (8 * TMg), rem = L - blk * 8 * TMg;
                            const int width = tnq - blk * 8 < 8 ? tnq - blk * 8 : 8;
                            mb = rem / width;
                            nb = blk * 8 + (rem - mb * width);
                            p = q;
                            return true;
                        }
                        break;   // a slot behind the end of this XCD's share of problem q
                    }
                    local -= per;
                }
            }
            return false;
        } else {
            for (; vb < n_vblocks; vb += (int)gridDim.x)
                if (xcd_tile(vb, TMg, TNg, mb, nb)) return true;
            return false;
        }
    };

    const int woff = (wn * 2) * SEG + lane * 8;             // + tn * SEG + kbl * BLOB + plane * 512
    const int xoff = OPB + (wm * TMT) * SEG + lane * 8;     // + tm * SEG + ...
    f32x16 acc[2][TMT];
    uint2 r0w[2][3], r0x[TMT][3], r1w[2][3], r1x[TMT][3];   // the fragments of the current and of the next 64-k block

// the BF6 operand is 6 registers; the builtin's type is 8 wide — the upper two lanes are left UNDEFINED (shufflevector
// index -1) so that no zeroing moves are emitted for them (24 v_mov per 128 k otherwise)
#define FQ_FRAG(R) __builtin_shufflevector(i32x6{(int)R[0].x, (int)R[0].y, (int)R[1].x, (int)R[1].y, (int)R[2].x, (int)R[2].y}, \
                                           i32x6{0, 0, 0, 0, 0, 0}, 0, 1, 2, 3, 4, 5, -1, -1)
#if GEMM_ABL == 6
#define FQ_MFMA1(RW, RX, I)                                                                                          \
    asm volatile("" : "+v"(acc[(I) / TMT][(I) % TMT]) : "v"(FQ_FRAG(RW[(I) / TMT])), "v"(FQ_FRAG(RX[(I) % TMT])));
#else
#define FQ_MFMA1(RW, RX, I)                                                                                          \
    acc[(I) / TMT][(I) % TMT] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(FQ_FRAG(RW[(I) / TMT]), FQ_FRAG(RX[(I) % TMT]), \
                                                                                acc[(I) / TMT][(I) % TMT], 3, 3, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#endif
// fragment I of a block: I < 2 the weight row tiles, then the TMT token row tiles; three conflict-free 8-byte reads
#define FQ_READ1(ST, KBL, RW, RX, I)                                                                                 \
    if ((GEMM_ABL == 3 || GEMM_ABL == 4) && (I) < 2) {                                                               \
    } else if (GEMM_ABL == 5) {                                                                                      \
    } else if ((I) < 2) {                                                                                            \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) RW[(I) < 2 ? (I) : 0][p] =                                     \
            *reinterpret_cast<const uint2*>((ST) + woff + ((I) < 2 ? (I) : 0) * SEG + (KBL) * BLOB + p * 512);       \
    } else if ((I) < 2 + TMT) {                                                                                      \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) RX[(I) >= 2 && (I) < 2 + TMT ? (I) - 2 : 0][p] =               \
            *reinterpret_cast<const uint2*>((ST) + xoff + ((I) >= 2 && (I) < 2 + TMT ? (I) - 2 : 0) * SEG + (KBL) * BLOB + p * 512); \
    }
    static_assert(2 + TMT <= 2 * TMT && DPW <= 6 * TMT, "a block has enough MFMAs to carry its successor's fragments and the DMA share");
    auto half_a = [&](const unsigned char* st) {   // block 0 of a stage from r0, block 1's fragments into r1
#pragma unroll
        for (int i = 0; i < 2 * TMT; ++i) {
            FQ_MFMA1(r0w, r0x, i)
            FQ_READ1(st, 1, r1w, r1x, i)
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto half_b = [&](const unsigned char* sn, int s_dma, auto dma_c, auto next_c) {   // block 1 from r1, the next stage's block 0 into r0
#pragma unroll
        for (int i = 0; i < 2 * TMT; ++i) {
            FQ_MFMA1(r1w, r1x, i)
            if (decltype(dma_c)::value) {   // this step's share of the DPW requests (one, sometimes two)
#pragma unroll
                for (int j = i * DPW / (2 * TMT); j < (i + 1) * DPW / (2 * TMT); ++j) issue_one(s_dma, j);
            }
            if (decltype(next_c)::value) { FQ_READ1(sn, 0, r0w, r0x, i) }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    int vb = blockIdx.x, mb = 0, nb = 0;
    int prob = 0;
    if (!next_tile(vb, mb, nb, prob)) return;
    plan(mb, nb, prob);
    request_first_stages();
    {   // the first tile starts as soon as its first stage is there
        const int younger = nk - 1 < STAGES - 1 ? nk - 1 : STAGES - 1;
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * DPW) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(DPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    for (;;) {
        if constexpr (MULTI) {
            N = FQ_PICK(N, prob);
            out = FQ_PICK(out, prob);
        }
        const int m0 = mb * BM, n0 = nb * BN;
        static_assert(TMT == 4, "the asm statement behind the K loop names sr[0..3]");
        f16 sr[TMT] = {};
        u32x4 sc[2][2] = {}, bs[2][2] = {};
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int tm = 0; tm < TMT; ++tm) acc[tn][tm] = f32x16{0};
#pragma unroll
        for (int i = 0; i < 2 + TMT; ++i) { FQ_READ1(smem, 0, r0w, r0x, i) }
        {
            int s = 0;
            for (; s + STAGES < nk; ++s) {   // stages whose buffer is refilled
                half_a(smem + (s % STAGES) * TILE_BYTES);
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"((STAGES - 2) * DPW) : "memory");   // stage s + 1 has landed (s + 2 may be in flight)
                __builtin_amdgcn_s_barrier();   // ... for every wave, and every wave holds all of stage s in registers: its buffer is free
                half_b(smem + ((s + 1) % STAGES) * TILE_BYTES, s + STAGES, std::true_type{}, std::true_type{});
            }
            for (; s + 1 < nk; ++s) {        // the last STAGES - 1 stages with a successor: nothing left to request
                half_a(smem + (s % STAGES) * TILE_BYTES);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                half_b(smem + ((s + 1) % STAGES) * TILE_BYTES, 0, std::false_type{}, std::true_type{});
            }
            half_a(smem + (s % STAGES) * TILE_BYTES);   // the last stage
            // the epilogue's scales, requested under the last eight MFMAs (the fragment registers of the next block are free)
            if (out.y != nullptr) {
#pragma unroll
                for (int tm = 0; tm < TMT; ++tm) {
                    const int m = m0 + wm * (TMT * 32) + tm * 32 + c;
                    sr[tm] = out.srow[m < M ? m : M - 1];
                }
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    int nbase = n0 + wn * 64 + tn * 32 + 16 * h;
                    nbase = nbase < N ? nbase : N - 16;   // N % 16 == 0; runs beyond N are never stored
                    sc[tn][0] = *reinterpret_cast<const u32x4*>(out.scol + nbase);
                    sc[tn][1] = *reinterpret_cast<const u32x4*>(out.scol + nbase + 8);
                    if (out.bias != nullptr) {
                        bs[tn][0] = *reinterpret_cast<const u32x4*>(out.bias + nbase);
                        bs[tn][1] = *reinterpret_cast<const u32x4*>(out.bias + nbase + 8);
                    }
                }
            }
            half_b(smem, 0, std::false_type{}, std::false_type{});
        }

        // The scales are waited for HERE (hipcc puts its s_waitcnt in front of this statement), before the next tile's requests go
        // out: the LDS-DMA instructions are inline asm, invisible to the compiler's vmcnt bookkeeping — a later "all but my last
        // seven loads" of its own would wait for eleven of the eighteen requests instead.
        asm volatile("" : "+v"(sr[0]), "+v"(sr[1]), "+v"(sr[2]), "+v"(sr[3]), "+v"(sc[0][0]), "+v"(sc[0][1]), "+v"(sc[1][0]),
                     "+v"(sc[1][1]), "+v"(bs[0][0]), "+v"(bs[0][1]), "+v"(bs[1][0]), "+v"(bs[1][1]));
        // the next tile of this workgroup: its first stages are requested NOW, in front of the epilogue
        int nvb = vb + (int)gridDim.x, nmb = 0, nnb = 0, nprob = 0;
        const bool more = next_tile(nvb, nmb, nnb, nprob);
        if (more) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // every wave has read the last stage: all three buffers are free
            plan(nmb, nnb, nprob);
            request_first_stages();
        }

        // ---- epilogue: lane (h, c) of tile (tn, tm): n = n0 + 64 wn + 32 tn + 16 h + r, token m = m0 + 128 wm + 32 tm + c
#pragma unroll
        for (int tm = 0; tm < TMT; ++tm) {
            const int m = m0 + wm * (TMT * 32) + tm * 32 + c;
            if (m >= M) continue;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int nbase = n0 + wn * 64 + tn * 32 + 16 * h;
                if (nbase >= N) continue;  // N % 16 == 0
                if (out.c != nullptr) {
                    int v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = (int)acc[tn][tm][r];  // an integer below 2^24: exact
                    int4* cp = reinterpret_cast<int4*>(out.c + (int64_t)m * N + nbase);
#pragma unroll
                    for (int g = 0; g < 4; ++g) cp[g] = make_int4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
                }
                if (GEMM_ABL == 2) {
                    asm volatile("" : : "v"(acc[tn][tm]));
                } else if (out.y != nullptr) {
                    f16x8 o0, o1;
                    if (may_clamp)
                        dequant16f<true>(acc[tn][tm], sr[tm], __builtin_bit_cast(f16x8, sc[tn][0]), __builtin_bit_cast(f16x8, sc[tn][1]),
                                         out.bias != nullptr, __builtin_bit_cast(f16x8, bs[tn][0]), __builtin_bit_cast(f16x8, bs[tn][1]), o0, o1);
                    else
                        dequant16f<false>(acc[tn][tm], sr[tm], __builtin_bit_cast(f16x8, sc[tn][0]), __builtin_bit_cast(f16x8, sc[tn][1]),
                                          out.bias != nullptr, __builtin_bit_cast(f16x8, bs[tn][0]), __builtin_bit_cast(f16x8, bs[tn][1]), o0, o1);
                    if (GEMM_ABL == 1) {
                        asm volatile("" : : "v"(o0), "v"(o1));
                        continue;
                    }
                    uint4* yp = reinterpret_cast<uint4*>(out.y + (int64_t)m * N + nbase);
                    yp[0] = __builtin_bit_cast(uint4, o0);   // (plain stores: non-temporal ones measured 163 -> 173 us)
                    yp[1] = __builtin_bit_cast(uint4, o1);
                }
            }
        }
        if (!more) break;
        vb = nvb;
        mb = nmb;
        nb = nnb;
        prob = nprob;
        // The requested stages have had the whole epilogue to land. An interior tile with the fused epilogue alone issued exactly
        // 4 TMT stores behind them (vmcnt counts loads and stores in issue order on gfx9): the wait leaves those in flight — they
        // drain under the next tile's first stages. Every other case waits for everything.
        if (out.c == nullptr && out.y != nullptr && m0 + BM <= M && n0 + BN <= N) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(4 * TMT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#undef FQ_FRAG
#undef FQ_MFMA1
#undef FQ_READ1
#undef FQ_PICK
}

}  // namespace

int64_t fq_bf6_blob_bytes(int64_t rows, int K) {
    if (rows < 0 || K <= 0 || (K & 63)) return -1;
    return ((rows + 31) / 32) * (int64_t)(K / 64) * BLOB;
}

// -1000: K % 64 != 0
int fq_launch_i4_to_bf6_multi(int nsrc, const uint8_t* const* q, int64_t rows, int K, int perm, uint8_t* blob, int n_cu, hipStream_t stream) {
    if ((K & 63) || rows < 1 || nsrc < 1 || nsrc > 4) return -1000;
    ConvSrc srcs;
    for (int p = 0; p < 4; ++p) srcs.s[p] = q[p < nsrc ? p : nsrc - 1];
    const int64_t total = nsrc * ((rows + 31) / 32) * (int64_t)((K / 64 + 1) / 2) * 64;
    int64_t blocks = (total + 255) / 256;
    if (blocks > (int64_t)n_cu * 16) blocks = (int64_t)n_cu * 16;
    hipLaunchKernelGGL(fq_i4_to_bf6_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, srcs, nsrc, rows, K / 2, perm, blob);
    return (int)hipGetLastError();
}

int fq_launch_i4_to_bf6(const uint8_t* q, int64_t rows, int K, int perm, uint8_t* blob, int n_cu, hipStream_t stream) {
    return fq_launch_i4_to_bf6_multi(1, &q, rows, K, perm, blob, n_cu, stream);
}

namespace {

int bf6_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

// Measurement knob: force the feature width of the <= 128-token tiles (0: the rule below)
#ifndef GEMM_BN_SMALL
#define GEMM_BN_SMALL 0
#endif

// The tile geometry of a launch of `cols` feature columns in total (one problem: N; several: their sum, each rounded up to the tile).
//   bm: 256-token tiles unless they would leave a quarter of the CUs (or more) without one;
//   bn: 256, except for DECODE-sized calls (M <= 128: one row of tiles), where a narrower tile is what puts workgroups on the chip —
//       128 features when 256-wide tiles are fewer than 3/4 of the CUs (a 64-wide, one-wave tile does not build: its DMA sources leave the
//       scalar file).
void bf6_geometry(int64_t M, const int* Ns, int n, int cus, int& bm, int& bn) {
    auto tiles = [&](int w) {
        int64_t t = 0;
        for (int p = 0; p < n; ++p) t += (Ns[p] + w - 1) / w;
        return t;
    };
    bn = 256;
    bm = ((M + 255) / 256) * tiles(256) * 4 < (int64_t)cus * 3 ? 128 : 256;
    if (M <= 128) {
        if (tiles(256) * 4 < (int64_t)cus * 3) bn = 128;
        if (GEMM_BN_SMALL) bn = GEMM_BN_SMALL;
    }
}

template <int BM, int MODE, int BN>
int bf6_launch(int64_t n_vblocks, int wg_per_cu, const uint8_t* xb, const uint8_t* wb, int64_t M, int N, int K, const GemmOut& o,
               const GemmMultiArg<(MODE != 0)>& mp, hipStream_t stream) {
    using G = Geo<BM, BN>;
    // persistent workgroups: wg_per_cu per CU, a multiple of 8 so that a workgroup stays on its XCD's share of the tile sequence
    int64_t blocks = (int64_t)(bf6_cus() / 8) * 8 * wg_per_cu;
    if (blocks < 8) blocks = 8;
    if (blocks > n_vblocks) blocks = n_vblocks;
    auto kern = fq_gemm_bf6_kernel<BM, MODE, BN>;
    FQ_RAISE_LDS_CAP(kern, G::STAGES * G::TILE_BYTES);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(G::GT), G::STAGES * G::TILE_BYTES, stream, xb, wb, (int)M, N, K / 64, (int)n_vblocks, o, mp);
    return (int)hipGetLastError();
}

}  // namespace

// -1000: shape not covered (K % 128 != 0, N % 16 != 0, K > 2^18): use the i8 kernel
int fq_launch_gemm_bf6(const uint8_t* xblob, const uint8_t* wblob, int64_t M, int N, int K, int32_t* c, f16* y,
                       const f16* srow, const f16* scol, const f16* bias, hipStream_t stream) {
    if ((K & 127) || (N & 15) || K > (1 << 18) || M < 1 || N < 1 || M > (1 << 30)) return -1000;
    GemmOut o;
    o.c = c;
    o.y = y;
    o.srow = srow;
    o.scol = scol;
    o.bias = bias;
    int bm, bn;
    bf6_geometry(M, &N, 1, bf6_cus(), bm, bn);
    const int64_t n_vblocks = 8 * ((((M + bm - 1) / bm) * ((N + bn - 1) / bn) + 7) / 8);  // see xcd_tile
    const GemmMultiArg<false> none{};
    if (bn == 128) return bf6_launch<128, 0, 128>(n_vblocks, 2, xblob, wblob, M, N, K, o, none, stream);
    if (bm == 128) return bf6_launch<128, 0, 256>(n_vblocks, 1, xblob, wblob, M, N, K, o, none, stream);
    return bf6_launch<256, 0, 256>(n_vblocks, 1, xblob, wblob, M, N, K, o, none, stream);
}

// Up to four problems with common M and K in one launch (fq_gemm_bf6_kernel<BM, true>). -1000: shape not covered.
int fq_launch_gemm_bf6_multi(int n, const uint8_t* const* xblob, const uint8_t* const* wblob, int64_t M, const int* Ns, int K, f16* const* y,
                             const f16* const* srow, const f16* const* scol, const f16* const* bias, hipStream_t stream) {
    if (n < 1 || n > 4 || (K & 127) || K > (1 << 18) || M < 1 || M > (1 << 30)) return -1000;
    for (int p = 0; p < n; ++p)
        if (Ns[p] < 1 || (Ns[p] & 15)) return -1000;
    int bm, bn;
    bf6_geometry(M, Ns, n, bf6_cus(), bm, bn);
    GemmMultiArg<true> mp = {};
    mp.n = n;
    int tn = 0;
    for (int p = 0; p < 4; ++p) {
        const int q = p < n ? p : n - 1;   // (unused slots repeat the last problem: the selects never read garbage)
        mp.tn0[p] = tn;
        if (p < n) tn += (Ns[q] + bn - 1) / bn;
        mp.N[p] = Ns[q];
        mp.xb[p] = xblob[q];
        mp.wb[p] = wblob[q];
        mp.out[p].c = nullptr;
        mp.out[p].y = y[q];
        mp.out[p].srow = srow[q];
        mp.out[p].scol = scol[q];
        mp.out[p].bias = bias[q];
    }
    for (int p = n; p < 5; ++p) mp.tn0[p] = tn;
    int64_t n_vblocks = 0;   // eight XCD shares of the per-problem shares (next_tile, MODE 1)
    for (int p = 0; p < n; ++p) n_vblocks += 8 * ((((M + bm - 1) / bm) * (int64_t)(mp.tn0[p + 1] - mp.tn0[p]) + 7) / 8);
    const GemmOut o0 = mp.out[0];
    if (bn == 128) return bf6_launch<128, 1, 128>(n_vblocks, 2, mp.xb[0], mp.wb[0], M, mp.N[0], K, o0, mp, stream);
    if (bm == 128) return bf6_launch<128, 1, 256>(n_vblocks, 1, mp.xb[0], mp.wb[0], M, mp.N[0], K, o0, mp, stream);
    return bf6_launch<256, 1, 256>(n_vblocks, 1, mp.xb[0], mp.wb[0], M, mp.N[0], K, o0, mp, stream);
}
