// fq_gemm_i4.hip — INT4 x INT4 -> INT32 GEMM, the consumer of the packed activations (first "next" row):
//     C[m][n] = sum_k x[m][k] * w[n][k],   x [M, K/2] and w [N, K/2] packed two's-complement nibbles (even k low)
// Replaces deploy/kernels/gemm.cu:8-47 (CUTLASS int4 tensor-op GEMM, row-major A, column-major B, int32 out) behind
// deploy.matmul (deploy/__init__.py:37-41), and — with the epilogue fused — Linear4bit.forward
// (deploy/nn/linear.py:41-56: matmul, then sym_dequant quant.cu:66-85, then bias).
//
// gfx950 has no INT4 matrix instruction; v_mfma_i32_32x32x32_i8 takes signed bytes. A nibble left in the HIGH half of
// its byte IS the signed byte 16*q, so unpacking is one AND for the odd elements and shift + AND for the even ones
// (3 VALU per packed dword, single-width: they issue in the shadow of the MFMAs), every product is 256*x*w and the
// accumulator is shifted right by 8 at the end (exact; |sum| < 2^31 for K <= 131072). No sign-extension
// arithmetic, no zero-point correction terms.
//
// Tiling: 256 x 256 output tile per 8-wave workgroup, K in steps of 128 nibbles (64 bytes per row), three LDS stages of
// 32 KB filled by LDS-DMA (16 rows x 64 bytes per instruction, 16-byte pieces XOR-swizzled by (row>>2)&3 on the source
// side so that the ds_read_b64 fragment reads are conflict-free), counted vmcnt, one barrier per K-step.
// The weight rows feed the MFMA A operand with a row permutation (prow) that leaves each lane with 16 CONSECUTIVE n of
// one output row m: 64-byte (int32) or 32-byte (fp16) contiguous stores.
#include "fq_gemm_common.hpp"

namespace {

using namespace fqgemm;

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_g;

constexpr int BM = 256, BN = 256;     // output tile (tokens x output features)
constexpr int BKB = 64;               // packed bytes of K per stage and row (128 nibbles = 4 MFMA K-steps)
#ifndef FQ_GEMM_STAGES
#define FQ_GEMM_STAGES 3   // LDS stages of 32 KB, all of them in flight or being read
#endif
constexpr int STAGES = FQ_GEMM_STAGES;
constexpr int TILE_BYTES = (BM + BN) * BKB;  // 32 KB: [W rows 0..255][X rows 0..255], 64 bytes each
#ifndef FQ_GEMM_WAVES
#define FQ_GEMM_WAVES 16  // 16: 4 x 4 waves of 64 x 64 (4 accumulator tiles, <= 128 VGPRs, 4 waves per SIMD whose
                          // unpack VALU fills the other waves' MFMA shadows); 8: 2 x 4 waves of 128 x 64
#endif
constexpr int GW = FQ_GEMM_WAVES;     // waves per workgroup
constexpr int GT = GW * 64;           // threads
constexpr int NWM = GW / 4;           // waves along the token dimension (4 along the feature dimension)
constexpr int TMT = BM / 32 / NWM;    // 32-token tiles per wave
constexpr int DPW = 32 / GW;          // DMA instructions per wave and stage

// 16-byte piece swizzles (XOR on the piece index 0..3 of a 64-byte row): the 16 lanes a ds_read_b128 serves per LDS
// cycle must hit 16 different (row & 3, piece) slots. Token rows are read in natural order (lane c -> row c), weight
// rows through prow(): lanes 0..15 read rows {0-3, 16-19, 4-7, 20-23}.
__device__ __forceinline__ int swz_x(int r) { return (r >> 2) & 3; }
__device__ __forceinline__ int swz_w(int r) { return ((r >> 2) & 1) | (((r >> 4) & 1) << 1); }


// (unpack16 — 16 nibbles -> 16 signed bytes = 16 q — lives in fq_gemm_common.hpp: shared with the fused decode launch of fq_kron64.hip)

__global__ __launch_bounds__(GT) void fq_gemm_i4_kernel(const uint8_t* __restrict__ X, const uint8_t* __restrict__ W,
                                                        int M, int N, int Kb, GemmOut out) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[STAGES * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % NWM, wn = wave / NWM;  // wave tile: TMT*32 tokens x 64 features
    int mb, nb;
#ifdef FQ_GEMM_LINEAR_ORDER
    const int nb_n = (N + BN - 1) / BN;
    mb = blockIdx.x / nb_n, nb = blockIdx.x - mb * nb_n;
    if (mb * BM >= M) return;
#else
    if (!xcd_tile(blockIdx.x, (M + BM - 1) / BM, (N + BN - 1) / BN, mb, nb)) return;
#endif
    const int m0 = mb * BM, n0 = nb * BN;
    const int nk = Kb / BKB;

    // ---- DMA plan: a stage is 32 instructions of 1 KB (16 rows x 64 B); wave w issues instructions 4w .. 4w+3.
    // instruction i < 16 -> weight rows 16 i .., else token rows 16 (i - 16) .. ; lane l: row + (l >> 2), physical
    // 16-byte piece l & 3 holds logical piece (l & 3) ^ swz(row)
    const unsigned char* gsrc[DPW];
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int i = wave * DPW + j;
        const int r = (i & 15) * 16 + (lane >> 2);  // row inside the 256-row half
        const int piece = (lane & 3) ^ (i < 16 ? swz_w(r) : swz_x(r));
        int64_t grow;
        const unsigned char* base;
        if (i < 16) {
            grow = n0 + r < N ? n0 + r : N - 1;
            base = W;
        } else {
            grow = m0 + r < M ? m0 + r : M - 1;
            base = X;
        }
        gsrc[j] = base + grow * Kb + piece * 16;
    }
    const unsigned lds0 = (unsigned)(size_t)(lds_void_g*)smem;
    auto issue_stage = [&](int kb) {
        const unsigned dst = lds0 + (unsigned)((kb % STAGES) * TILE_BYTES) + (unsigned)(wave * DPW) * 1024u;
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const unsigned char* src = gsrc[j] + (int64_t)kb * BKB;
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %2\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, off\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(src), "s"(__builtin_amdgcn_readfirstlane((int)(dst + (unsigned)j * 1024u)))
                : "memory");
        }
    };

    // ---- fragment read offsets (bytes inside a stage). A lane reads whole 16-byte pieces (ds_read_b128: one LDS
    // instruction per fragment and PAIR of K-steps, 64-bank addressing; two ds_read_b64 get merged by hipcc into
    // ds_read2st64_b64, which banks modulo 32 and measured 16 conflict cycles per instruction): logical piece 2 p + h of
    // row r sits at r*64 + ((2 p + h) ^ swz(r))*16; its first 8 bytes feed K-step 2p, the second 8 bytes K-step 2p+1
    // (any split of k over the steps is fine as long as both operands use the same one).
    int woff[2], xoff[TMT];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) woff[tn] = (wn * 64 + tn * 32 + prow(c)) * BKB;
#pragma unroll
    for (int tm = 0; tm < TMT; ++tm) xoff[tm] = BN * BKB + (wm * (TMT * 32) + tm * 32 + c) * BKB;
    const int wsw = swz_w(prow(c));  // tile bases are multiples of 32 rows: the swizzle only sees prow(c) / c
    const int xsw = swz_x(c);

    i32x16 acc[2][TMT];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int tm = 0; tm < TMT; ++tm) acc[tn][tm] = i32x16{0};

    // Pipeline. Per stage two pairs of K-steps; the pieces of pair 1 are read BEFORE the stage hand-over (wait for
    // stage kb+1, barrier, DMA of stage kb+STAGES into the buffer every wave has just finished reading) and the pieces
    // of the next stage's pair 0 right after it, so the MFMAs of pair 1 start the moment the barrier releases and
    // cover those reads: no LDS round trip between a barrier and the first MFMA behind it.
    uint4 r0w[2], r0x[TMT], r1w[2], r1x[TMT];
#define FQ_READ(ST, P, RW, RX)                                                                                      \
    {                                                                                                                \
        _Pragma("unroll") for (int tn = 0; tn < 2; ++tn) RW[tn] =                                                    \
            *reinterpret_cast<const uint4*>((ST) + woff[tn] + (((2 * (P) + h) ^ wsw) << 4));                         \
        _Pragma("unroll") for (int tm = 0; tm < TMT; ++tm) RX[tm] =                                                  \
            *reinterpret_cast<const uint4*>((ST) + xoff[tm] + (((2 * (P) + h) ^ xsw) << 4));                         \
    }
#define FQ_COMPUTE(RW, RX)                                                                                          \
    _Pragma("unroll") for (int half = 0; half < 2; ++half) {                                                         \
        i32x4 wf[2], xf[TMT];                                                                                        \
        _Pragma("unroll") for (int tn = 0; tn < 2; ++tn) wf[tn] =                                                    \
            unpack16(half ? make_uint2(RW[tn].z, RW[tn].w) : make_uint2(RW[tn].x, RW[tn].y));                        \
        _Pragma("unroll") for (int tm = 0; tm < TMT; ++tm) xf[tm] =                                                  \
            unpack16(half ? make_uint2(RX[tm].z, RX[tm].w) : make_uint2(RX[tm].x, RX[tm].y));                        \
        _Pragma("unroll") for (int tn = 0; tn < 2; ++tn) _Pragma("unroll") for (int tm = 0; tm < TMT; ++tm)          \
            acc[tn][tm] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[tn], xf[tm], acc[tn][tm], 0, 0, 0);               \
    }
#pragma unroll
    for (int p = 0; p < STAGES; ++p)
        if (p < nk) issue_stage(p);
    {   // stage 0 landed: up to STAGES - 1 younger stages of this wave in flight
        const int younger = nk - 1 < STAGES - 1 ? nk - 1 : STAGES - 1;
        if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(3 * DPW) : "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * DPW) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(DPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    FQ_READ(smem, 0, r0w, r0x)
    for (int kb = 0; kb < nk; ++kb) {
        const unsigned char* st = smem + (kb % STAGES) * TILE_BYTES;
        FQ_READ(st, 1, r1w, r1x)
        FQ_COMPUTE(r0w, r0x)
        if (kb + 1 < nk) {
            // stage kb+1 landed: the stages kb+2 .. kb+STAGES-1 of this wave may still be in flight
            const int younger = nk - 2 - kb < STAGES - 2 ? nk - 2 - kb : STAGES - 2;
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(2 * DPW) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(DPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // every wave holds all of stage kb in registers: its buffer is free
            if (kb + STAGES < nk) issue_stage(kb + STAGES);
            const unsigned char* sn = smem + ((kb + 1) % STAGES) * TILE_BYTES;
            FQ_READ(sn, 0, r0w, r0x)
        }
        FQ_COMPUTE(r1w, r1x)
    }
#undef FQ_READ
#undef FQ_COMPUTE

    // ---- epilogue: lane (h, c) of tile (tn, tm) holds n = n0 + wn*64 + tn*32 + 16 h + r (r = 0..15) of token
    //      m = m0 + wm*128 + tm*32 + c; products carry a factor 256 ----
#pragma unroll
    for (int tm = 0; tm < TMT; ++tm) {
        const int m = m0 + wm * (TMT * 32) + tm * 32 + c;
        if (m >= M) continue;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int nbase = n0 + wn * 64 + tn * 32 + 16 * h;
            if (nbase >= N) continue;  // N % 16 == 0 (checked by the launcher)
            int v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[tn][tm][r] >> 8;
            if (out.c != nullptr) {
                int4* cp = reinterpret_cast<int4*>(out.c + (int64_t)m * N + nbase);
#pragma unroll
                for (int g = 0; g < 4; ++g) cp[g] = make_int4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
            }
            if (out.y != nullptr) {
                f16x8 o0, o1;
                // the float-pipeline epilogue of the FP6 path (dequant16f: 5 - 6 VALU per element instead of 11): |v| < 2^24 is exact in fp32
                f32x16 vf;
#pragma unroll
                for (int r = 0; r < 16; ++r) vf[r] = (float)v[r];
                const f16x8 s0 = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(out.scol + nbase));
                const f16x8 s1 = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(out.scol + nbase + 8));
                f16x8 b0 = {}, b1 = {};
                if (out.bias != nullptr) {
                    b0 = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(out.bias + nbase));
                    b1 = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(out.bias + nbase + 8));
                }
                if (Kb > 10176 / 2) dequant16f<true>(vf, out.srow[m], s0, s1, out.bias != nullptr, b0, b1, o0, o1);   // (|q| <= 64 K)
                else dequant16f<false>(vf, out.srow[m], s0, s1, out.bias != nullptr, b0, b1, o0, o1);
                uint4* yp = reinterpret_cast<uint4*>(out.y + (int64_t)m * N + nbase);
                yp[0] = __builtin_bit_cast(uint4, o0);
                yp[1] = __builtin_bit_cast(uint4, o1);
            }
        }
    }
}

// Any other shape the reference accepts (K % 32 == 0, deploy/__init__.py:38): one thread per output element,
// 32-bit loads of 8 nibbles, sign extension by shifts. Correct, not fast; the model shapes never get here.
__global__ __launch_bounds__(256) void fq_gemm_i4_simple_kernel(const uint8_t* __restrict__ X, const uint8_t* __restrict__ W,
                                                               int M, int N, int Kb, GemmOut out) {
    const int64_t total = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i - (int64_t)m * N);
        const uint32_t* xp = reinterpret_cast<const uint32_t*>(X + (int64_t)m * Kb);
        const uint32_t* wp = reinterpret_cast<const uint32_t*>(W + (int64_t)n * Kb);
        int acc = 0;
        for (int k = 0; k < Kb / 4; ++k) {
            const uint32_t a = xp[k], b = wp[k];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += (((int)(a << (28 - 4 * e))) >> 28) * (((int)(b << (28 - 4 * e))) >> 28);
        }
        if (out.c != nullptr) out.c[i] = acc;
        if (out.y != nullptr) {
            f16 v = dequant1(acc, out.srow[m], out.scol[n]);
            if (out.bias != nullptr) v = v + out.bias[n];
            out.y[i] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Skinny GEMM for decode-sized batches (M <= 128): the work is streaming the weights once (N K / 2 bytes), so
//  * the weights come from an IMAGE in MFMA fragment order (fq_i4_to_frag_kernel, built once per layer): blob
//    (32-feature tile rt, 64-k block kb) = 64 lanes x 16 bytes, lane (h, c) = row prow(c), packed bytes
//    32 kb + 16 h .. + 15 -> every load of a wave is 1 KB contiguous and a wave walks memory sequentially;
//  * a 16-wave workgroup owns one feature tile and splits K between its waves (enough loads in flight with only
//    N / 32 workgroups), each wave reusing its weight fragment for all token tiles; the 16 partial 32 x 32 tiles meet
//    in LDS through ds_add_u32;
//  * the activations (M x K / 2 bytes, L2-resident) are read directly as B fragments.
// The big-tile kernel above spends ~46 us on any M <= 128 (it streams a 256-token tile that is mostly padding).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fq_i4_to_frag_kernel(const uint8_t* __restrict__ W, int N, int Kb, uint4* __restrict__ img) {
    const int KB = Kb / 32;
    const int64_t total = (int64_t)((N + 31) / 32) * KB * 64;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const int64_t blob = i >> 6;
        const int64_t rt = blob / KB;
        const int kb = (int)(blob - rt * KB);
        const int64_t row = rt * 32 + prow(lane & 31);
        img[i] = row < N ? *reinterpret_cast<const uint4*>(W + row * Kb + kb * 32 + (lane >> 5) * 16) : make_uint4(0, 0, 0, 0);
    }
}

constexpr int SK_WAVES = 16;

// Up to four problems that share M and K (q / k / v or up / gate of a decoder layer, each with its own activations, image, scales, bias,
// output) as ONE launch: the feature tiles of the problems side by side in the grid. A decode-sized launch sits on the ~4 us floor of a
// small dispatch whatever it streams (4.4 us for the 8 MB of a 4096 x 4096 projection, 4.0 us for the 2 MB of a 1024-wide one:
// profiles/r05_skinny_prefetch.txt), so three launches cost three floors.
struct SkinnyProblems {
    const uint8_t* X[4];
    const uint4* Wimg[4];
    GemmOut out[4];
    int N[4];
    int tile0[4];   // first feature tile (workgroup) of problem p
    int n;
};

// KSP (round 6, third session): THE K RANGE OF A FEATURE TILE OVER gridDim.y WORKGROUPS. A lone 4096-wide projection is 128 feature tiles — half the
// CUs — and at 33 .. 128 rows its workgroups are busy with the int8 MFMAs and the nibble unpacking of two to four token tiles (down_proj, K = 14336:
// 14.6 us at 64 rows, 26.3 at 128: 2.0 / 1.1 TB/s of weights). With KSP every tile's 64-k blobs are split over 2 (4) workgroups; a workgroup leaves
// its int32 partial tile in `kws` (agent-scope stores, as the split decode attention's states), takes a ticket on the tile's counter, and the LAST to
// arrive adds the others' partials to its own and runs the epilogue. Integer sums: the result is the unsplit launch's bit for bit, whatever the order.
// kws: [tiles] counters (left at zero), then [gridDim.y][tiles][MT * 1024] partial sums.
template <int MT, bool MULTI, int U = 1, bool KSP = false>
__global__ __launch_bounds__(SK_WAVES * 64) void fq_gemm_i4_skinny_kernel(const uint8_t* __restrict__ X_,
                                                                         const uint4* __restrict__ Wimg_, int M, int N_,
                                                                         int Kb, GemmOut out_, SkinnyProblems pr, int* __restrict__ kws = nullptr) {
    __shared__ int tile[MT][32][33];  // [token tile][token][feature], +1 padding
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int rt = blockIdx.x;
    const int KB = Kb / 32;
    const uint8_t* __restrict__ X = X_;
    const uint4* __restrict__ Wimg = Wimg_;
    GemmOut out = out_;
    int N = N_;
    if (MULTI) {   // (workgroup-uniform selects)
        int pi = 0;
#pragma unroll
        for (int q = 1; q < 4; ++q)
            if (q < pr.n && rt >= pr.tile0[q]) pi = q;
        X = pr.X[0], Wimg = pr.Wimg[0], out = pr.out[0], N = pr.N[0];
#pragma unroll
        for (int q = 1; q < 4; ++q)
            if (pi == q) X = pr.X[q], Wimg = pr.Wimg[q], out = pr.out[q], N = pr.N[q], rt -= pr.tile0[q];
    }
    for (int i = tid; i < MT * 32 * 33; i += SK_WAVES * 64) (&tile[0][0][0])[i] = 0;
    __syncthreads();

    i32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = i32x16{0};
    const uint4* wp = Wimg + (size_t)rt * KB * 64 + lane;
    const uint8_t* xrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 32 + c;
        xrow[mt] = X + (size_t)(m < M ? m : M - 1) * Kb + 16 * h;
    }
    auto block = [&](int kb, const u32x4 a_) {
        const uint4 a = make_uint4(a_[0], a_[1], a_[2], a_[3]);
        const i32x4 a0 = unpack16(make_uint2(a.x, a.y)), a1 = unpack16(make_uint2(a.z, a.w));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const uint4 b = *reinterpret_cast<const uint4*>(xrow[mt] + (size_t)kb * 32);
            acc[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, unpack16(make_uint2(b.x, b.y)), acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, unpack16(make_uint2(b.z, b.w)), acc[mt], 0, 0, 0);
        }
    };
    // U weight blobs of a wave requested at a time. U = 1 keeps two workgroups per CU (41 VGPRs) — right for the wide launches (N = 14336:
    // 448 workgroups, the q/k/v and up/gate groups); a 4096-wide projection alone is 128 workgroups with ONE 1 KB load per wave in
    // flight, 2 MB on the whole chip: latency-bound at 1.9-3.3 TB/s — there U = 2 (measured with the knob on every launch, per-dispatch
    // durations: 7.56 -> 5.8 us on the N = 4096 launches, slower on N = 1024 and 14336: profiles/r05_skinny_prefetch.txt).
    int kb = wave, kb_end = KB;
    if constexpr (KSP) {       // this workgroup's share of the tile's blobs
        const int per = (KB + (int)gridDim.y - 1) / (int)gridDim.y;
        kb = (int)blockIdx.y * per + wave;
        kb_end = ((int)blockIdx.y + 1) * per < KB ? ((int)blockIdx.y + 1) * per : KB;
    }
    const int KB_ = kb_end;
    for (; kb + (U - 1) * SK_WAVES < KB_; kb += U * SK_WAVES) {
        u32x4 a_[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a_[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + (size_t)(kb + u * SK_WAVES) * 64));
#pragma unroll
        for (int u = 0; u < U; ++u) block(kb + u * SK_WAVES, a_[u]);
    }
    for (; kb < KB_; kb += SK_WAVES) block(kb, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + (size_t)kb * 64)));
    // lane (h, c) of tile mt: token 32 mt + c, features 16 h + r
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) atomicAdd(&tile[mt][c][16 * h + r], acc[mt][r]);
    __syncthreads();
    if constexpr (KSP) {
        const int S = (int)gridDim.y, tiles = (int)gridDim.x;
        int* part = kws + tiles + ((size_t)blockIdx.y * tiles + rt) * (MT * 1024);
        for (int i = tid; i < MT * 1024; i += SK_WAVES * 64)
            __hip_atomic_store(part + i, tile[i >> 10][(i >> 5) & 31][i & 31], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __shared__ unsigned s_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the partial tile has reached memory (gfx9 counts stores in vmcnt) ...
        __syncthreads();
        if (tid == 0) s_last = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(kws) + rt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(S - 1);   // ... before it is counted
        __syncthreads();
        if (!s_last) return;
        for (int i = tid; i < MT * 1024; i += SK_WAVES * 64) {
            int v = 0;
            for (int z = 0; z < S; ++z)
                if (z != (int)blockIdx.y) v += __hip_atomic_load(kws + tiles + ((size_t)z * tiles + rt) * (MT * 1024) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tile[i >> 10][(i >> 5) & 31][i & 31] += v;
        }
        if (tid == 0) __hip_atomic_store(reinterpret_cast<unsigned*>(kws) + rt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next launch finds the counter as this one did
        // (each thread adds into and then reads its OWN elements: the element loop below walks the same indices)
    }
    for (int i = tid; i < MT * 32 * 32; i += SK_WAVES * 64) {
        const int mt = i >> 10, tok = (i >> 5) & 31, nl = i & 31;
        const int m = mt * 32 + tok, n = rt * 32 + nl;
        if (m >= M || n >= N) continue;
        const int v = tile[mt][tok][nl] >> 8;  // products carry 256 (see unpack16)
        if (out.c != nullptr) out.c[(int64_t)m * N + n] = v;
        if (out.y != nullptr) {
            f16 y = dequant1(v, out.srow[m], out.scol[n]);
            if (out.bias != nullptr) y = y + out.bias[n];
            out.y[(int64_t)m * N + n] = y;
        }
    }
}

}  // namespace

// -1000: shape not accepted (K must be a multiple of 32, as deploy.matmul asserts).
int fq_launch_gemm_i4(const uint8_t* X, const uint8_t* W, int64_t M, int N, int K, int32_t* c, f16* y, const f16* srow,
                      const f16* scol, const f16* bias, hipStream_t stream) {
    if ((K & 31) || K < 32 || M < 1 || N < 1 || M > (1 << 30)) return -1000;
    GemmOut o;
    o.c = c;
    o.y = y;
    o.srow = srow;
    o.scol = scol;
    o.bias = bias;
    if ((K & 127) || (N & 15)) {
        int64_t blocks = (M * (int64_t)N + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(fq_gemm_i4_simple_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, X, W, (int)M, N, K / 2, o);
        return (int)hipGetLastError();
    }
    const int64_t blocks = 8 * ((((M + BM - 1) / BM) * ((N + BN - 1) / BN) + 7) / 8);  // see xcd_tile
    hipLaunchKernelGGL(fq_gemm_i4_kernel, dim3((unsigned)blocks), dim3(GT), 0, stream, X, W, (int)M, N, K / 2, o);
    return (int)hipGetLastError();
}

int64_t fq_i4_frag_bytes(int N, int K) {
    if (N < 1 || K < 64 || (K & 63)) return -1;
    return (int64_t)((N + 31) / 32) * (K / 64) * 1024;
}

int fq_launch_i4_to_frag(const uint8_t* W, int N, int K, void* img, int n_cu, hipStream_t stream) {
    if (fq_i4_frag_bytes(N, K) < 0) return -1000;
    const int64_t total = (int64_t)((N + 31) / 32) * (K / 64) * 64;
    int64_t blocks = (total + 255) / 256;
    if (blocks > (int64_t)n_cu * 16) blocks = (int64_t)n_cu * 16;
    hipLaunchKernelGGL(fq_i4_to_frag_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, W, N, K / 2, reinterpret_cast<uint4*>(img));
    return (int)hipGetLastError();
}

// -1000: M > 128 or K % 64 != 0 (use the tile kernel)
int fq_launch_gemm_i4_skinny(const uint8_t* X, const void* wimg, int64_t M, int N, int K, int32_t* c, f16* y, const f16* srow,
                             const f16* scol, const f16* bias, hipStream_t stream) {
    if (M < 1 || M > 128 || N < 1 || (K & 63) || K < 64 || K > 131072) return -1000;
    GemmOut o;
    o.c = c;
    o.y = y;
    o.srow = srow;
    o.scol = scol;
    o.bias = bias;
    const dim3 grid((unsigned)((N + 31) / 32));
    const uint4* img = reinterpret_cast<const uint4*>(wimg);
    const SkinnyProblems none = {};
    const bool deep = M <= 32 && (N + 31) / 32 <= 256 && (N + 31) / 32 >= 64 && K / 64 >= 2 * SK_WAVES;   // a lone 2048- to 8192-wide projection of <= 32 rows
                                                                                                        // (64 / 128 rows: no gain / 31.5 vs 28.5 us): see the kernel
    if (deep) hipLaunchKernelGGL((fq_gemm_i4_skinny_kernel<1, false, 2>), grid, dim3(SK_WAVES * 64), 0, stream, X, img, (int)M, N, K / 2, o, none);
    else if (M <= 32) hipLaunchKernelGGL((fq_gemm_i4_skinny_kernel<1, false>), grid, dim3(SK_WAVES * 64), 0, stream, X, img, (int)M, N, K / 2, o, none);
    else if (M <= 64) hipLaunchKernelGGL((fq_gemm_i4_skinny_kernel<2, false>), grid, dim3(SK_WAVES * 64), 0, stream, X, img, (int)M, N, K / 2, o, none);
    else hipLaunchKernelGGL((fq_gemm_i4_skinny_kernel<4, false>), grid, dim3(SK_WAVES * 64), 0, stream, X, img, (int)M, N, K / 2, o, none);
    return (int)hipGetLastError();
}

// How many workgroups share a feature tile's K range (1: the launch is not split), and the workspace the split launch needs. Split where the
// unsplit launch leaves CUs idle (at most 128 feature tiles) AND either its workgroups are busy with more than one token tile (33 rows or more) or the K
// range is long; at least two rounds of blobs per workgroup of a split (measured: profiles/r06_skinny_ksplit.txt).
int fq_gemm_i4_skinny_splits(int64_t M, int N, int K) {
    const int tiles = (N + 31) / 32, KB = K / 64;
    if (M < 1 || M > 128 || tiles > 128 || (K & 63)) return 1;
    // up to 32 rows (one token tile: the launch is latency-, not arithmetic-bound) only a long K range pays for the hand-over: measured on a 4096-wide
    // projection (r06c69) K = 14336: 8.5 -> 8.8 us at one row, 11.2 -> 9.4 at 16, 15.8 -> 11.3 at 32; K = 28672: 16.6 -> 13.3 at one row, 30.6 -> 18.5 at 32
    if (M < 33 && (K < 8192 || (M < 9 && K < 16384))) return 1;
    int s = tiles <= 64 ? 4 : 2;
    while (s > 1 && KB / s < 2 * SK_WAVES) s >>= 1;
    return s;
}
int64_t fq_gemm_i4_skinny_split_ws_bytes(int64_t M, int N, int K) {
    const int s = fq_gemm_i4_skinny_splits(M, N, K);
    if (s <= 1) return 0;
    const int64_t tiles = (N + 31) / 32, mt = M <= 32 ? 1 : M <= 64 ? 2 : 4;
    return (((tiles + 3) & ~(int64_t)3) + (int64_t)s * tiles * mt * 1024) * 4 + 64;
}
// fq_launch_gemm_i4_skinny with the K range of every feature tile split over fq_gemm_i4_skinny_splits workgroups (kws: zeroed counters in front,
// fq_gemm_i4_skinny_split_ws_bytes bytes); kws == nullptr or a geometry that is not split: the plain launch. Same results bit for bit.
int fq_launch_gemm_i4_skinny_split(const uint8_t* X, const void* wimg, int64_t M, int N, int K, f16* y, const f16* srow, const f16* scol,
                                   const f16* bias, int* kws, hipStream_t stream) {
    const int s = kws != nullptr ? fq_gemm_i4_skinny_splits(M, N, K) : 1;
    if (s <= 1) return fq_launch_gemm_i4_skinny(X, wimg, M, N, K, nullptr, y, srow, scol, bias, stream);
    GemmOut o;
    o.c = nullptr;
    o.y = y;
    o.srow = srow;
    o.scol = scol;
    o.bias = bias;
    const int tiles = (N + 31) / 32;
    const dim3 grid((unsigned)tiles, (unsigned)s);
    const uint4* img = reinterpret_cast<const uint4*>(wimg);
    const SkinnyProblems none = {};
    int* ws = kws;      // counters [tiles] (the partial sums start at the next multiple of four ints: see the bytes function; the kernel uses `tiles` ints)
    if (M <= 32) hipLaunchKernelGGL((fq_gemm_i4_skinny_kernel<1, false, 2, true>), grid, dim3(SK_WAVES * 64), 0, stream, X, img, (int)M, N, K / 2, o, none, ws);
    else if (M <= 64) hipLaunchKernelGGL((fq_gemm_i4_skinny_kernel<2, false, 1, true>), grid, dim3(SK_WAVES * 64), 0, stream, X, img, (int)M, N, K / 2, o, none, ws);
    else hipLaunchKernelGGL((fq_gemm_i4_skinny_kernel<4, false, 1, true>), grid, dim3(SK_WAVES * 64), 0, stream, X, img, (int)M, N, K / 2, o, none, ws);
    return (int)hipGetLastError();
}

// -1000: n not in 1..4, M > 128 or K % 64 != 0
int fq_launch_gemm_i4_skinny_multi(int n, const uint8_t* const* X, const void* const* wimg, int64_t M, const int* N, int K, f16* const* y,
                                   const f16* const* srow, const f16* const* scol, const f16* const* bias, hipStream_t stream) {
    if (n < 1 || n > 4 || M < 1 || M > 128 || (K & 63) || K < 64 || K > 131072) return -1000;
    SkinnyProblems pr = {};
    pr.n = n;
    int tiles = 0;
    for (int p = 0; p < n; ++p) {
        if (N[p] < 1) return -1000;
        pr.X[p] = X[p];
        pr.Wimg[p] = reinterpret_cast<const uint4*>(wimg[p]);
        pr.out[p].c = nullptr;
        pr.out[p].y = y[p];
        pr.out[p].srow = srow[p];
        pr.out[p].scol = scol[p];
        pr.out[p].bias = bias ? bias[p] : nullptr;
        pr.N[p] = N[p];
        pr.tile0[p] = tiles;
        tiles += (N[p] + 31) / 32;
    }
    const dim3 grid((unsigned)tiles);
    GemmOut o = {};
    if (M <= 32) hipLaunchKernelGGL((fq_gemm_i4_skinny_kernel<1, true>), grid, dim3(SK_WAVES * 64), 0, stream, nullptr, nullptr, (int)M, 0, K / 2, o, pr);
    else if (M <= 64) hipLaunchKernelGGL((fq_gemm_i4_skinny_kernel<2, true>), grid, dim3(SK_WAVES * 64), 0, stream, nullptr, nullptr, (int)M, 0, K / 2, o, pr);
    else hipLaunchKernelGGL((fq_gemm_i4_skinny_kernel<4, true>), grid, dim3(SK_WAVES * 64), 0, stream, nullptr, nullptr, (int)M, 0, K / 2, o, pr);
    return (int)hipGetLastError();
}
