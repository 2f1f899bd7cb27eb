// fq_gemm_bf6.hip — the INT4 x INT4 GEMM of Linear4bit on the FP6 matrix path of gfx950 ("explore
// FP4/FP6 block-scaled MFMA"), bit-identical to fq_gemm_i4.hip.
//
// gfx950 has no INT4 MFMA. fq_gemm_i4.hip widens nibbles to bytes for v_mfma_i32_32x32x32_i8. But every integer in
// [-8, 7] is exactly representable in BF6 (E3M2: 0, +-1, 2, 3, 4, 5, 6, 7, 8), and
// v_mfma_scale_f32_32x32x64_f8f6f4 with both operands BF6 and unit block scales (E8M0 = 127) multiplies them exactly
// and accumulates in fp32: |product| <= 64 and K <= 2^18 keep every partial sum an integer below 2^24, i.e. the fp32
// accumulator IS the int32 result. Measured on MI355X with register-resident operands (tools/scratch/bf6_probe.hip):
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
// Tiling: 256 x 256 per 8-wave workgroup, wave tile 128 tokens x 64 features (8 accumulator tiles: 6 fragments per 8
// MFMAs), three LDS stages of 48 KB, counted vmcnt, one barrier per stage — the pipeline of fq_gemm_i4.hip.
#include "fq_gemm_common.hpp"
#include <stdlib.h>

namespace {

using namespace fqgemm;

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_b;

constexpr int BM = 256, BN = 256;
constexpr int BLOB = 1536;                     // bytes: 32 rows x 64 k of BF6
constexpr int SEG = 2 * BLOB;                  // a row tile's two blobs of one stage (128 k)
constexpr int OPB = 8 * SEG;                   // one operand's share of a stage: 8 row tiles
constexpr int TILE_BYTES = 2 * OPB;            // 48 KB: [W tiles 0..7][X tiles 0..7]
constexpr int STAGES = 3;
#ifndef FQ_BF6_WAVES
#define FQ_BF6_WAVES 8   // 8: 2 x 4 waves of 128 x 64 (2 per SIMD); 16: 4 x 4 waves of 64 x 64 (4 per SIMD, <= 128 VGPRs)
#endif
constexpr int GW = FQ_BF6_WAVES, GT = GW * 64;
constexpr int NWM = GW / 4;                    // waves along the token dimension (4 along the feature dimension)
constexpr int TMT = BM / 32 / NWM;             // token tiles per wave
#ifndef FQ_BF6_LATE_DMA
#define FQ_BF6_LATE_DMA 0   // 1: refill after the second half's MFMAs instead of right behind the barrier (measured neutral: 227 vs 224 us)
#endif
#ifndef FQ_BF6_DMA_WAVES
#define FQ_BF6_DMA_WAVES 8   // waves that issue the DMA. (4 = one per SIMD, so that the two waves of a SIMD leave each barrier
                             // differently loaded and stop marching in step: measured no difference, 226 vs 224 us)
#endif
constexpr int DW = FQ_BF6_DMA_WAVES;
constexpr int DPW = (TILE_BYTES / 1024) / DW;  // DMA instructions per issuing wave and stage

// ---- INT4 nibbles -> BF6 blobs ----------------------------------------------------------------------------------
// E3M2 codes of 0..8; a negative value sets bit 5
__device__ __forceinline__ unsigned bf6_code(int v) {  // v in [-8, 7]
    const unsigned mag = (unsigned)(v < 0 ? -v : v);
    const unsigned tab[9] = {0x00, 0x0C, 0x10, 0x12, 0x14, 0x15, 0x16, 0x17, 0x18};
    return tab[mag] | (v < 0 ? 0x20u : 0u);
}

__global__ __launch_bounds__(256) void fq_i4_to_bf6_kernel(const uint8_t* __restrict__ src, int64_t rows, int Kb, int perm,
                                                           uint8_t* __restrict__ dst) {
    __shared__ unsigned short lut[256];  // packed byte (two nibbles, even k low) -> 12 bits of codes
    {
        const int b = threadIdx.x;
        const int lo = ((b & 15) ^ 8) - 8, hi = (((b >> 4) & 15) ^ 8) - 8;
        lut[b] = (unsigned short)(bf6_code(lo) | (bf6_code(hi) << 6));
    }
    __syncthreads();
    const int KB = Kb / 32;  // blobs per row tile (64 k = 32 packed bytes)
    const int64_t n_rt = (rows + 31) / 32;
    const int64_t total = n_rt * KB * 64;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const int64_t blob = i >> 6;
        const int64_t rt = blob / KB;
        const int kb = (int)(blob - rt * KB);
        const int kh = lane >> 5, r = lane & 31;
        const int64_t row = rt * 32 + (perm ? prow(r) : r);
        uint4 in = make_uint4(0x88888888u, 0x88888888u, 0x88888888u, 0x88888888u);  // never used: rows beyond the end are zeros
        const bool ok = row < rows;
        if (ok) in = *reinterpret_cast<const uint4*>(src + row * Kb + kb * 32 + kh * 16);
        const unsigned w[4] = {in.x, in.y, in.z, in.w};
        unsigned long long g[4];  // 48 bits each: the 8 codes of one input dword
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned long long t0 = lut[w[j] & 255], t1 = lut[(w[j] >> 8) & 255], t2 = lut[(w[j] >> 16) & 255],
                                     t3 = lut[w[j] >> 24];
            g[j] = ok ? (t0 | (t1 << 12) | (t2 << 24) | (t3 << 36)) : 0ull;
        }
        const unsigned long long o0 = g[0] | (g[1] << 48);
        const unsigned long long o1 = (g[1] >> 16) | (g[2] << 32);
        const unsigned long long o2 = (g[2] >> 32) | (g[3] << 16);
        unsigned long long* d = reinterpret_cast<unsigned long long*>(dst + blob * BLOB) + lane;
        d[0] = o0;
        d[64] = o1;
        d[128] = o2;
    }
}

// ---- the GEMM ---------------------------------------------------------------------------------------------------
template <int ABL>  // measurement builds (wrong results): 1 = no MFMA, 2 = fragment reads of the first stage only, 4 = no DMA after the prologue,
                    // 8 = no epilogue stores, 16 = one output row in eight dequantised and stored
__global__ __launch_bounds__(GT, GW / 4) void fq_gemm_bf6_kernel(const uint8_t* __restrict__ XB, const uint8_t* __restrict__ WB,
                                                            int M, int N, int KB, GemmOut out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % NWM, wn = wave / NWM;  // wave tile: tokens (32 TMT) wm .., features 64 wn ..
    int mb, nb;
    if (!xcd_tile(blockIdx.x, (M + BM - 1) / BM, (N + BN - 1) / BN, mb, nb)) return;
    const int m0 = mb * BM, n0 = nb * BN;
    const int nk = KB / 2;  // stages of 128 k
    const int mt_last = (M + 31) / 32 - 1, nt_last = (N + 31) / 32 - 1;

    // DMA plan: instruction i = 6 wave + j: operand i / 24 (0 = W), row tile (i % 24) / 3, 1 KB part i % 3 of its 3 KB.
    // The source of an instruction is wave-uniform (SGPR base) + 16 * lane: twelve address VGPRs less than per-lane
    // pointers — those had pushed the kernel into a spill whose reload sat between the DMA instructions of a stage
    // behind an s_waitcnt vmcnt(0), i.e. every stage waited for its own loads (found in the ISA, cost ~2x).
    const unsigned char* gbase[DPW];
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int i = (wave < DW ? wave : 0) * DPW + j;
        const int op = i / 24, t = (i % 24) / 3, part = i % 3;
        int rt = (op == 0 ? n0 : m0) / 32 + t;
        const int last = op == 0 ? nt_last : mt_last;
        rt = rt < last ? rt : last;  // tiles beyond the matrix re-read its last tile (their outputs are never stored)
        gbase[j] = (op == 0 ? WB : XB) + (int64_t)rt * KB * BLOB + part * 1024;
    }
    const unsigned voff = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)(size_t)(lds_void_b*)smem;
    auto issue_stage = [&](int s) {
        if (wave >= DW) return;  // (their vmcnt waits below find nothing outstanding and fall through)
        const unsigned dst = lds0 + (unsigned)((s % STAGES) * TILE_BYTES) + (unsigned)(wave * DPW) * 1024u;
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const unsigned char* src = gbase[j] + (int64_t)s * SEG;
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %3\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, %2\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(voff), "s"(src), "s"(__builtin_amdgcn_readfirstlane((int)(dst + (unsigned)j * 1024u)))
                : "memory");
        }
    };

    const int woff = (wn * 2) * SEG + lane * 8;             // + tn * SEG + kbl * BLOB + plane * 512
    const int xoff = OPB + (wm * TMT) * SEG + lane * 8;     // + tm * SEG + ...

    f32x16 acc[2][TMT];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int tm = 0; tm < TMT; ++tm) acc[tn][tm] = f32x16{0};

#if FQ_BF6_WAVES == 16
    uint2 r0w[2][3], r0x[TMT][3];
#else
    uint2 r0w[2][3], r0x[TMT][3], r1w[2][3], r1x[TMT][3];
#endif
#define FQ_READ(ST, KBL, RW, RX)                                                                                    \
    {                                                                                                                \
        _Pragma("unroll") for (int tn = 0; tn < 2; ++tn) _Pragma("unroll") for (int p = 0; p < 3; ++p) RW[tn][p] =   \
            *reinterpret_cast<const uint2*>((ST) + woff + tn * SEG + (KBL) * BLOB + p * 512);                        \
        _Pragma("unroll") for (int tm = 0; tm < TMT; ++tm) _Pragma("unroll") for (int p = 0; p < 3; ++p) RX[tm][p] = \
            *reinterpret_cast<const uint2*>((ST) + xoff + tm * SEG + (KBL) * BLOB + p * 512);                        \
    }
// the BF6 operand is 6 registers; the builtin's type is 8 wide — the upper two lanes are left UNDEFINED (shufflevector
// index -1) so that no zeroing moves are emitted for them (24 v_mov per 128 k otherwise)
typedef int i32x6 __attribute__((ext_vector_type(6)));
#define FQ_FRAG(R) __builtin_shufflevector(i32x6{(int)R[0].x, (int)R[0].y, (int)R[1].x, (int)R[1].y, (int)R[2].x, (int)R[2].y}, \
                                           i32x6{0, 0, 0, 0, 0, 0}, 0, 1, 2, 3, 4, 5, -1, -1)
#define FQ_COMPUTE(RW, RX)                                                                                          \
    {                                                                                                                \
        i32x8 wf[2], xf[TMT];                                                                                        \
        _Pragma("unroll") for (int tn = 0; tn < 2; ++tn) wf[tn] = FQ_FRAG(RW[tn]);                                   \
        _Pragma("unroll") for (int tm = 0; tm < TMT; ++tm) xf[tm] = FQ_FRAG(RX[tm]);                                 \
        if (ABL & 1) {                                                                                               \
            _Pragma("unroll") for (int tn = 0; tn < 2; ++tn) _Pragma("unroll") for (int tm = 0; tm < TMT; ++tm)      \
                acc[tn][tm][0] += (float)(wf[tn][0] ^ xf[tm][5]);                                                    \
        } else {                                                                                                     \
        _Pragma("unroll") for (int tn = 0; tn < 2; ++tn) _Pragma("unroll") for (int tm = 0; tm < TMT; ++tm)          \
            acc[tn][tm] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[tn], xf[tm], acc[tn][tm], 3, 3, 0,       \
                                                                          0x7f7f7f7f, 0, 0x7f7f7f7f);               \
        }                                                                                                            \
    }
#pragma unroll
    for (int p = 0; p < STAGES; ++p)
        if (p < nk) issue_stage(p);
    {
        const int younger = nk - 1 < STAGES - 1 ? nk - 1 : STAGES - 1;
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * DPW) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(DPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
#if FQ_BF6_WAVES == 16
    // 16-wave build: four waves per SIMD hide the fragment reads of each other, so a wave reads and multiplies one 64-k
    // block at a time with a single fragment buffer (<= 128 VGPRs)
    for (int s = 0; s < nk; ++s) {
        const unsigned char* st = smem + (s % STAGES) * TILE_BYTES;
        FQ_READ(st, 0, r0w, r0x)
        FQ_COMPUTE(r0w, r0x)
        __builtin_amdgcn_sched_barrier(0);
        FQ_READ(st, 1, r0w, r0x)
        FQ_COMPUTE(r0w, r0x)
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nk) {
            const int younger = nk - 2 - s < STAGES - 2 ? nk - 2 - s : STAGES - 2;
            if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(DPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s + STAGES < nk) issue_stage(s + STAGES);
        }
    }
#else
    FQ_READ(smem, 0, r0w, r0x)
    for (int s = 0; s < nk; ++s) {
        const unsigned char* st = smem + (s % STAGES) * TILE_BYTES;
        if (!(ABL & 2) || s == 0) FQ_READ(st, 1, r1w, r1x)
        FQ_COMPUTE(r0w, r0x)
        __builtin_amdgcn_sched_barrier(0);  // these MFMAs cover the reads above; hipcc otherwise sinks them below the barrier
        if (s + 1 < nk) {
            const int younger = nk - 2 - s < STAGES - 2 ? nk - 2 - s : STAGES - 2;
            if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(DPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // every wave holds all of stage s in registers: its buffer is free
#if !FQ_BF6_LATE_DMA
            if (s + STAGES < nk && !(ABL & 4)) issue_stage(s + STAGES);
#endif
            const unsigned char* sn = smem + ((s + 1) % STAGES) * TILE_BYTES;
            if (!(ABL & 2)) FQ_READ(sn, 0, r0w, r0x)
        }
        __builtin_amdgcn_sched_barrier(0);
        FQ_COMPUTE(r1w, r1x)
        __builtin_amdgcn_sched_barrier(0);
#if FQ_BF6_LATE_DMA   // refill the freed buffer AFTER the second half's MFMAs are queued: shortens barrier -> first MFMA
        if (s + 1 < nk && s + STAGES < nk && !(ABL & 4)) issue_stage(s + STAGES);
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
#endif
#undef FQ_READ
#undef FQ_FRAG
#undef FQ_COMPUTE

    // ---- epilogue: lane (h, c) of tile (tn, tm): n = n0 + 64 wn + 32 tn + 16 h + r, token m = m0 + 128 wm + 32 tm + c
#pragma unroll
    for (int tm = 0; tm < TMT; ++tm) {
        const int m = m0 + wm * (TMT * 32) + tm * 32 + c;
        if (m >= M) continue;
        if ((ABL & 16) && (c & 7)) continue;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int nbase = n0 + wn * 64 + tn * 32 + 16 * h;
            if (nbase >= N) continue;  // N % 16 == 0
            int v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = (int)acc[tn][tm][r];  // an integer below 2^24: exact
            if (out.c != nullptr) {
                int4* cp = reinterpret_cast<int4*>(out.c + (int64_t)m * N + nbase);
#pragma unroll
                for (int g = 0; g < 4; ++g) cp[g] = make_int4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
            }
            if (out.y != nullptr) {
                f16x8 o0, o1;
                dequant16(v, out.srow[m], out.scol + nbase, out.bias != nullptr ? out.bias + nbase : nullptr, o0, o1);
                uint4* yp = reinterpret_cast<uint4*>(out.y + (int64_t)m * N + nbase);
                if (!(ABL & 8) || o0[0] == (f16)12345.0f) {
                    yp[0] = __builtin_bit_cast(uint4, o0);
                    yp[1] = __builtin_bit_cast(uint4, o1);
                }
            }
        }
    }
}

}  // namespace

int64_t fq_bf6_blob_bytes(int64_t rows, int K) {
    if (rows < 0 || K <= 0 || (K & 63)) return -1;
    return ((rows + 31) / 32) * (int64_t)(K / 64) * BLOB;
}

// -1000: K % 64 != 0
int fq_launch_i4_to_bf6(const uint8_t* q, int64_t rows, int K, int perm, uint8_t* blob, int n_cu, hipStream_t stream) {
    if ((K & 63) || rows < 1) return -1000;
    const int64_t total = ((rows + 31) / 32) * (int64_t)(K / 64) * 64;
    int64_t blocks = (total + 255) / 256;
    if (blocks > (int64_t)n_cu * 16) blocks = (int64_t)n_cu * 16;
    hipLaunchKernelGGL(fq_i4_to_bf6_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, q, rows, K / 2, perm, blob);
    return (int)hipGetLastError();
}

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
    const int64_t blocks = 8 * ((((M + BM - 1) / BM) * ((N + BN - 1) / BN) + 7) / 8);  // see xcd_tile
#ifdef FQ_MEASURE  // measurement builds only (tools/scratch/gemm_ablate.sh): ablation selected from the environment
    const char* e = getenv("FQ_GEMM_ABLATE");
    const int abl = e ? atoi(e) : 0;
#else
    constexpr int abl = 0;
#endif
#define FQ_LAUNCH(A)                                                                                                 \
    {                                                                                                                \
        FQ_RAISE_LDS_CAP(fq_gemm_bf6_kernel<A>, STAGES * TILE_BYTES);                                                \
        hipLaunchKernelGGL(fq_gemm_bf6_kernel<A>, dim3((unsigned)blocks), dim3(GT), STAGES * TILE_BYTES, stream, xblob, \
                           wblob, (int)M, N, K / 64, o);                                                             \
    }
    if (abl == 1) FQ_LAUNCH(1)
    else if (abl == 2) FQ_LAUNCH(2)
    else if (abl == 4) FQ_LAUNCH(4)
    else if (abl == 6) FQ_LAUNCH(6)
    else if (abl == 8) FQ_LAUNCH(8)
    else if (abl == 16) FQ_LAUNCH(16)
    else FQ_LAUNCH(0)
#undef FQ_LAUNCH
    return (int)hipGetLastError();
}
