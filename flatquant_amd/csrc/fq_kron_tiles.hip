// fq_kron_tiles.hip — fused Kronecker transform + per-token INT4 quantisation (packed output) for the factor pairs whose token
// is not four 32-column n'-tiles wide, or more than four row tiles tall: 80 x 112 (8960, Qwen2.5-1.5B ffn), 86 x 128 (11008, Llama-2-7B ffn),
// 128 x 144 (18432, DeepSeek-V3 dense ffn: deepseekv3_utils.py:343-348), 128 x 148 (18944, Qwen2.5-7B ffn), 144 x 192 (27648, Qwen2.5-32B
// ffn), 168 x 176 (29568, Qwen2.5-72B ffn) — function_utils.py:11-21 pairs that until round 4 ran the workgroup-per-token kernel with ONE
// token resident per CU (0.21-0.32 of the HBM roofline; 86 x 128 the trio kernel) — on fp16 AND bf16 activations (bf16 also 112 x 128).
//
// The structure is fq_kron_trio.hip's, with the geometry as template parameters <MT, NT, N, GROUPS, LKS, RS, T>: one persistent workgroup
// per CU holds GROUPS token groups of NT waves (a wave per n'-tile), the waves of a group meet on LDS counters and the groups drift apart,
// tokens are claimed from a workgroup counter one ahead, staged by LDS-DMA, and never leave registers between GEMM 1 and the packed stores.
// What is new:
//   * N = 16 KS1 for any KS1 (7, 8, 9, 11, 12 here): CPR = N / 8 chunks per row, unpadded rows; the bank rotation of a row is a property
//     of CPR (tl_swz: chunk bit 0 flipped in rows 8..15 for CPR = 14, 18, 22, an XOR of the row's bits 1..3 for CPR = 24, of bits 0..3 for
//     16), and the per-lane DMA source offsets follow from it — a table in LDS, DMP x 64 dwords, filled once per workgroup;
//   * N % 16 != 0 (148): the token staged linearly (row pitch 296 bytes), fragments as two 8-byte reads, 32 zero bytes behind the token
//     for the last K-step of the last row, a last half-tile with 4 valid columns (TilesGeom::ODDN);
//   * a last n'-tile that is half padding (N % 32 = 16): its upper half-wave neither contributes extrema nor stores;
//   * NO zero rows under the token: the row index of an A fragment is clamped to M - 1 instead (a padding row of U then holds finite
//     copies, and meets zero rows of L in GEMM 2), so a buffer is exactly 16 LKS rows and 144 x 192 fits two tokens + its L image
//     (trimmed to the K-steps that hold rows of L: LKS x MT KB) in 160 KB;
//   * RS: the wave's R fragments (KS1 x 4 registers: 48 at N = 192) stream from the L2-resident image through a ring instead of
//     living in registers, as in fq_kron_duo.hip;
//   * group counts follow the register file and the LDS: 4 x 4 waves (80 x 112, 127 VGPRs), 3 x 4 (86 x 128), 2 x 5 (128 x 144 / 148), 2 x 6
//     (144 x 192), 1 x 6 (168 x 176: two tokens + the L image would be 184 KB).
// Same mathematics, rounding points, fragment image (fq_kron_prepare_kernel) and quantiser helpers as every other Kronecker kernel;
// bit-identical to the workgroup-per-token kernel (tests/test_gpu_kron_tiles.py). Measurements: profiles/r04_tiles_timing.txt,
// r04_kron_128x144_pmc.txt.
#include "fq_common.hpp"
#ifndef FQ_PRIO_MFMA
#define FQ_PRIO_MFMA 2   // s_setprio level of a wave inside its GEMM phases (0: off); see fq_kron_duo.hip
#endif
#include "fq_dma.hpp"

namespace {

// measurement knobs (tools/variants.sh): groups of the 80 x 112 launch, A fragments in flight, streamed R at N = 144
#ifndef TILES_G112
#define TILES_G112 4   // (measured: 3 groups 113 us, 4 groups 107.5 us per 16384 tokens of 80 x 112; 121-127 VGPRs with four A fragments in flight)
#endif
#ifndef TILES_DA
#define TILES_DA 0   // 0: the per-geometry default below
#endif
#ifndef TILES_DB
#define TILES_DB 0   // 0: the default below (2)
#endif
#ifndef TILES_G144
#define TILES_G144 2
#endif
#ifndef TILES_RS144
#define TILES_RS144 false
#endif
#ifndef TILES_ABL
#define TILES_ABL 0   // measurement builds: 1 no quantiser, 2 no GEMM 1, 4 no GEMM 2, 8 no stores, 16 no DMA after the first
#endif

// bank rotation of row r: the 16 lanes of a ds_read_b128 group — 16 CONSECUTIVE rows of one K-half — must hit the 16 different
// 16-byte bank groups; lane c reads group (CPR c + p) mod 16, p the chunk. CPR = 16: all rows alias -> XOR the row's four low bits;
// CPR = 8 (mod 16): rows alias in two classes -> XOR bits 1..3; CPR = 4 (mod 8): four classes -> XOR bits 2..3; CPR = 2 (mod 4)
// (14, 18, 22): CPR c mod 16 runs through the eight EVEN groups twice -> rows 8..15 flip chunk bit 0. (The first build had no
// rotation for that class — "eight rows, eight groups" — and PMC showed SQ_LDS_BANK_CONFLICT = 11.8 M cycles per launch at 128 x 144,
// more than the LDS was otherwise active: profiles/r04_kron_128x144_pmc.txt.)
template <int CPR>
__device__ __forceinline__ int tl_swz(int r) {
    return CPR % 16 == 0 ? (r & 15) : CPR % 16 == 8 ? ((r >> 1) & 7) : CPR % 8 == 4 ? ((r >> 2) & 3) : ((r >> 3) & 1);
}
template <int CPR>
constexpr bool tl_has_swz() { return true; }
typedef uint2 tl_uint2_a2 __attribute__((aligned(2)));

template <int MT, int NT, int N, int GROUPS, int LKS>
struct TilesGeom {
    // ODDN (N % 16 != 0, N % 4 == 0: 148): a row is not a whole number of 16-byte chunks. The token is staged LINEARLY (row pitch 2 N
    // bytes: 296 = 74 dwords, sixteen consecutive rows start in sixteen different bank pairs), a fragment is two 8-byte reads, the
    // last K-step reads past the row's end into the next row (finite values against the zero rows of R) and the last row into 32
    // zero bytes behind the buffer; the last n'-tile holds N - 16 (2 NT - 1) valid columns in its upper half-wave.
    static constexpr bool ODDN = N % 16 != 0;
    static constexpr int KS1 = (N + 15) / 16, CPR = ODDN ? 16 : N / 8, THREADS = GROUPS * NT * 64;   // (ODDN: CPR only names the "no rotation" class)
    static constexpr int LFR = LKS * MT * 64;                 // uint4: the K-steps of the L image that hold rows of L
    static constexpr int TOKBUF = LKS * 16 * N * 2 + (ODDN ? 32 : 0);   // bytes: 16 LKS >= M rows (+ the zero tail)
    static constexpr int RED = LFR * 16 + GROUPS * TOKBUF;    // [max x8][min x8] floats per group
    static constexpr int CTL = RED + GROUPS * 64;             // [meet x4][next][claim x4]
    // the per-lane DMA source offsets repeat every DMP instructions (16 rows = 16 CPR slots, an instruction fills 64): a table of
    // DMP x 64 dwords, filled once per workgroup — computed per instruction they cost ~15 VALU each (+24 % VALU at 128 x 144, measured)
    static constexpr int DMP = ODDN ? 1 : CPR / (CPR % 4 == 0 ? 4 : CPR % 2 == 0 ? 2 : 1);
    static constexpr int DMT = CTL + 48;
    static constexpr int LDS = DMT + DMP * 256;
    static_assert(ODDN || (DMP * 64) % (16 * CPR) == 0, "table period: a whole number of 16-row blocks");
    static_assert(N % 4 == 0 && (LKS * 16 * N * 2) % 16 == 0 && NT == (N + 31) / 32 && LKS <= 2 * MT && LKS > 2 * MT - 2 && GROUPS <= 4 && NT <= 8, "geometry");
};

__device__ __forceinline__ unsigned tl_lds_read(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ unsigned tl_lds_add_rtn(unsigned addr, unsigned val) {
    unsigned v;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "v"(val) : "memory");
    return v;
}
__device__ __forceinline__ void tl_lds_write(unsigned addr, unsigned val) {
    asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(val) : "memory");
}
// group meeting on a counter in LDS (fq_kron_trio.hip: trio_meet)
__device__ __forceinline__ void tl_meet(unsigned cnt_lds, unsigned target, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(cnt_lds), "v"(1u) : "memory");
    for (;;) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(cnt_lds) : "memory");
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)v) >= target) break;
        __builtin_amdgcn_s_sleep(1);
    }
}
#define TILES_MEET() { meet_n += NT; tl_meet(meet, meet_n, lane); }

// per-lane source offset of DMA instruction i (which fills the LDS slots [64 i, 64 i + 64) linearly), relative to the
// instruction's own KB and biased by +128 bytes (an XOR moves a lane at most 7 chunks back: never negative)
template <int CPR>
__device__ __forceinline__ unsigned tl_dma_off(int i, int lane) {
    if (!tl_has_swz<CPR>()) return (unsigned)(lane * 16 + 128);
    const int q = i * 64 + lane, r = q / CPR, pch = q - r * CPR;
    return (unsigned)((lane + ((pch ^ tl_swz<CPR>(r)) - pch)) * 16 + 128);
}

// H16 (round 5, N = 64 — the Hadamard rotations of 11008 = 172 x 64 (Llama-2-7B ffn), 8960 = 140 x 64, 5120 = 80 x 64 run as ONE Kronecker
// launch in front of deploy.nn.Quantizer: hadamard_utils.py:132-141 + deploy/nn/quantization.py:13-36): the output arithmetic of
// fq_kron_tall.hip's H16 instantiations — the transform times the post-scale rounded to fp16 PAIRS, extrema on the pairs, the Quantizer's
// fp16 scale and the two-operation exact fp16 quotient (fq_quant8_h16), FQ_RATIO_POST's zero-token rule — and, when out.y is given, the
// rounded transform itself (kronecker_matmul / the rotation alone: no clip set). A token of N = 64 is two n'-tiles: TWO waves per
// token group and four groups per CU, each wave running both GEMMs of its 32 columns without meeting anybody (the contraction of GEMM 2
// is over rows, inside the wave) — where fq_kron_tall.hip splits the token by ROW tile and pays two workgroup barriers per token over
// 6 waves on 4 SIMDs (163 us per 16384 tokens of 172 x 64 with or without its loads: profiles/r05_tall_prefetch.txt).
template <int MT, int NT, int N, int GROUPS, int LKS, bool RS, typename T, bool H16 = false>
__global__ __launch_bounds__(GROUPS * NT * 64) void fq_kron_tiles_kernel(const T* __restrict__ x, const uint4* __restrict__ ws,
                                                                       int64_t rows, int64_t tpb, int M, FqQuantOut out) {
    typedef TilesGeom<MT, NT, N, GROUPS, LKS> G;
    typedef typename FqVec<T>::x8 X8;
    static_assert(!H16 || (FqVec<T>::is_f16 && N % 32 == 0), "the fp16 Quantizer epilogue: fp16 activations, whole n'-tiles");
    constexpr int KS1 = G::KS1, CPR = G::CPR, THREADS = G::THREADS;
    constexpr bool ODDN = G::ODDN;
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave / NT, wq = wave - grp * NT;   // token group, n'-tile of this wave
    uint4* lfr = reinterpret_cast<uint4*>(smem);
    unsigned char* tokbuf = smem + G::LFR * 16 + grp * G::TOKBUF;
    float* red = reinterpret_cast<float*>(smem + G::RED) + grp * 16;   // [max x8][min x8]
    unsigned* ctl = reinterpret_cast<unsigned*>(smem + G::CTL);
    const unsigned ctl_lds = (unsigned)(size_t)(lds_void*)ctl, meet = ctl_lds + grp * 4;
    const unsigned tok_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)tokbuf);
    const int64_t tok_bytes = (int64_t)M * (N * 2);
    const int n_slots = (M * N) >> 3, n_dma = (n_slots + 63) >> 6;   // 16-byte slots; 1 KB instructions per token, the last one may end inside its KB
    const int per = (n_dma + NT - 1) / NT;                      // this wave stages instructions [d0, d0 + dn)
    const int d0 = wq * per;
    const int dn = n_dma - d0 < per ? (n_dma - d0 < 0 ? 0 : n_dma - d0) : per;
    const int tail_lanes = n_slots & 63;
    const bool own_tail = tail_lanes != 0 && dn > 0 && d0 + dn == n_dma;
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(x) - 128;   // (tl_dma_off's bias)
    // N % 32 = 16: the upper half-wave of the last tile holds n' >= N (zero columns of R: its Y is 0 and belongs to nobody)
    const bool nvalid = N % 32 == 0 || (h * NT * 16 + wq * 16) < N;
    // ODDN: how many of the lane's 16 columns exist (16 everywhere but in the upper half-wave of the last tile: N - 16 (2 NT - 1) = 4 at N = 148)
    const int nval = !ODDN ? 16 : (N - (h * NT * 16 + wq * 16) >= 16 ? 16 : N - (h * NT * 16 + wq * 16) > 0 ? N - (h * NT * 16 + wq * 16) : 0);

    const int64_t blk_base = (int64_t)blockIdx.x * tpb;
    const int blk_cnt = (int)(rows - blk_base < tpb ? (rows - blk_base < 0 ? 0 : rows - blk_base) : tpb);

    // ---- once per workgroup: L image, control words, this wave's R fragments, first DMA ----
    {
        const uint4* lsrc = ws + NT * KS1 * 64;
        for (int i = tid; i < G::LFR; i += THREADS) lfr[i] = lsrc[i];
        if (tid < 12) ctl[tid] = tid == 4 ? GROUPS : 0;   // meeting counters, the next unclaimed token, (published claims)
        unsigned* dmt_w = reinterpret_cast<unsigned*>(smem + G::DMT);
        for (int e = tid; e < G::DMP * 64; e += THREADS) dmt_w[e] = ODDN ? (unsigned)((e & 63) * 16 + 128) : tl_dma_off<CPR>(e >> 6, e & 63);
        if (ODDN && tid < GROUPS * 8)   // 32 zero bytes behind every token (the DMA never writes them: its last instruction is lane-masked)
            *reinterpret_cast<unsigned*>(smem + G::LFR * 16 + (tid >> 3) * G::TOKBUF + M * (N * 2) + (tid & 7) * 4) = 0u;
    }
    const unsigned* dmt = reinterpret_cast<const unsigned*>(smem + G::DMT);
    const uint4* rsrc = ws + (size_t)wq * KS1 * 64;       // this wave's R fragments in the image (wave-uniform base)
    constexpr int NRF = RS ? 1 : KS1, DR = 4;
    X8 RF[NRF];
    X8 RB[DR];
    if (!RS) {
#pragma unroll
        for (int s = 0; s < NRF; ++s) RF[s] = __builtin_bit_cast(X8, rsrc[s * 64 + lane]);
    }
    __syncthreads();   // (the R fragments have arrived: vmcnt(0))
    if (!RS) {
#pragma unroll
        for (int s = 0; s < NRF; ++s) asm volatile("" : "+v"(RF[s]));
    }
    // this wave's share of token k's DMA, four instructions per M0 / base pair; a partial last instruction runs with the lanes
    // beyond the token masked off
    auto stage_token = [&](int k) {
        const unsigned char* src = xb + (blk_base + k) * tok_bytes + (int64_t)d0 * 1024;
        const int nfull = own_tail ? dn - 1 : dn;
        int ln = lane;
        asm volatile("" : "+v"(ln));   // (the offsets are recomputed here: hoisted, they are registers held across the GEMMs)
        for (int g = 0; g < nfull; g += 4) {
            unsigned rv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) rv[j] = dmt[((d0 + g + j) % G::DMP) * 64 + ln];
            dma_span(src + (int64_t)g * 1024, nfull - g < 4 ? nfull - g : 4, tok_lds + (unsigned)(d0 + g) * 1024, rv);
        }
        if (own_tail && lane < tail_lanes) {
            const unsigned r1[4] = {dmt[((d0 + nfull) % G::DMP) * 64 + ln], 0u, 0u, 0u};
            dma_span(src + (int64_t)nfull * 1024, 1, tok_lds + (unsigned)(d0 + nfull) * 1024, r1);
        }
    };
    if (grp < blk_cnt && dn > 0) stage_token(grp);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    auto prime_r = [&]() {
        if (RS) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int i = 0; i < DR - 1; ++i) RB[i] = __builtin_bit_cast(X8, rsrc[i * 64 + ln]);
        }
    };
    prime_r();

    FqGroupCursor gcur;
    const float ps = out.post_scale != 0.0f ? out.post_scale : 1.0f;
    const int ks_n = (M + 15) >> 4;   // K-steps of GEMM 2 that hold rows of L
    unsigned meet_n = 0;

    for (int k = grp; k < blk_cnt;) {   // k: the group's current token (of this workgroup's range), claimed one token ahead
        const int64_t tok = blk_base + k;

        // ================= phase A: GEMM 1 (U = X . R for this wave's n'-tile), fp16 rounding =================
        TILES_MEET()   // C|A: every wave of the group waited for its share of the DMA before its stores of phase C
        if (FQ_PRIO_MFMA) __builtin_amdgcn_s_setprio(FQ_PRIO_MFMA);
        X8 Uh[MT][2];
        {
            int cl = c;
            asm volatile("" : "+v"(cl));   // keep the address arithmetic inside the loop
            // rows of the last tile beyond the token read row M - 1 (finite; they meet zero rows of L)
            const int rl = (MT - 1) * 32 + cl < M ? (MT - 1) * 32 + cl : M - 1;
            const int swa = tl_swz<CPR>(cl), swl = tl_swz<CPR>(rl);   // (rows 32 mt + c rotate like row c)
            // CPR = 2 (mod 4): the rotation only flips chunk bit 0, i.e. the K-half — (2 s + h) ^ sw = 2 s + (h ^ sw): folded into the
            // base pointer, and 2 s stays in the instruction's offset field (an XOR per read would be three VALU per fragment)
            constexpr bool FOLD = CPR % 4 == 2;
            const uint4* tb = reinterpret_cast<const uint4*>(tokbuf) + cl * CPR + (FOLD ? (h ^ swa) : 0);
            const uint4* tl = reinterpret_cast<const uint4*>(tokbuf) + rl * CPR + (FOLD ? (h ^ swl) : 0);
            const unsigned char* ob = tokbuf + cl * (N * 2) + h * 16;   // ODDN: byte addresses, 8-byte aligned
            const unsigned char* ol = tokbuf + rl * (N * 2) + h * 16;
            auto afrag = [&](int i) -> X8 {   // i = s * MT + mt
                const int s = i / MT, mt = i % MT;
                if (ODDN) {
                    const unsigned char* pp = (mt == MT - 1 ? ol : ob + mt * 32 * (N * 2)) + s * 32;
                    const uint2 a = *reinterpret_cast<const uint2*>(pp), b = *reinterpret_cast<const uint2*>(pp + 8);
                    return __builtin_bit_cast(X8, uint4{a.x, a.y, b.x, b.y});
                }
                if (FOLD) return mt == MT - 1 ? __builtin_bit_cast(X8, tl[s * 2]) : __builtin_bit_cast(X8, tb[mt * 32 * CPR + s * 2]);
                return mt == MT - 1 ? __builtin_bit_cast(X8, tl[(s * 2 + h) ^ swl])
                                    : __builtin_bit_cast(X8, tb[mt * 32 * CPR + ((s * 2 + h) ^ swa)]);
            };
            f32x16 U[MT];
            constexpr int DA = TILES_DA ? TILES_DA : RS ? 4 : (KS1 > 8 || GROUPS == 4 ? 4 : 8), NA = KS1 * MT;   // fragment reads in flight (x 4 VGPRs)
            X8 A[DA];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) U[mt] = f32x16{0};
#pragma unroll
            for (int i = 0; i < DA - 1; ++i) A[i] = afrag(i);
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int s = i / MT;
                if (RS && i % MT == 0 && s + DR - 1 < KS1) RB[(s + DR - 1) % DR] = __builtin_bit_cast(X8, rsrc[(s + DR - 1) * 64 + ln]);
                if (i + DA - 1 < NA) A[(i + DA - 1) % DA] = afrag(i + DA - 1);
                if (!(TILES_ABL & 2)) U[i % MT] = fq_mfma32<T>(A[i % DA], RS ? RB[s % DR] : RF[RS ? 0 : s], U[i % MT]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int j = 0; j < 8; ++j) Uh[mt][p][j] = (T)U[mt][p * 8 + j];
        }

        // ================= phase B: next token's DMA, GEMM 2 (Y^T = U^T . L), extrema =================
        if (wq == 0 && lane == 0) tl_lds_write(ctl_lds + 20 + grp * 4, tl_lds_add_rtn(ctl_lds + 16, 1u));   // claim the group's next token
        TILES_MEET()   // A|B: the group has read its token buffer
        const int knext = __builtin_amdgcn_readfirstlane((int)tl_lds_read(ctl_lds + 20 + grp * 4));
        const bool more = !(TILES_ABL & 16) && knext < blk_cnt && dn > 0;
        if (more) stage_token(knext);
        if (knext < blk_cnt) prime_r();   // (behind the DMA in the queue; first used after phase C's vmcnt(0))
        f32x16 Y[MT];   // Y^T of tile (wq, mo): rows n' = h*NT*16 + wq*16 + r, col m' = 32 mo + c
        {
            int loff = lane;
            asm volatile("" : "+v"(loff));
            const uint4* mylfr = lfr + loff;
            // L fragments in flight (x 4 VGPRs). N = 64 (two waves per SIMD, 66 fragment reads per token and wave at 172 x 64): four —
            // measured 141.8 -> 135.8 us (172 x 64, fp16 Quantizer epilogue), 111.4 -> 106.3 (140 x 64); profiles/r05_tall_on_tiles.txt
            constexpr int DB = TILES_DB ? TILES_DB : (N == 64 ? 4 : 2), NB = LKS * MT;
            X8 B[DB];
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) Y[mo] = f32x16{0};
#pragma unroll
            for (int i = 0; i < DB - 1; ++i) B[i] = __builtin_bit_cast(X8, mylfr[i * 64]);
#pragma unroll
            for (int i = 0; i < NB; ++i) {   // i = ks * MT + mo
                const int ks = i / MT, mo = i % MT;
                if (i + DB - 1 < NB) B[(i + DB - 1) % DB] = __builtin_bit_cast(X8, mylfr[(i + DB - 1) * 64]);
                if (!(TILES_ABL & 4) && (ks < 2 * MT - 2 || ks < ks_n)) Y[mo] = fq_mfma32<T>(Uh[ks >> 1][ks & 1], B[i % DB], Y[mo]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (FQ_PRIO_MFMA) __builtin_amdgcn_s_setprio(0);
        float vmax = -INFINITY, vmin = INFINITY;
        uint32_t H[H16 ? MT : 1][8];   // H16: the fp16 pairs the deploy Quantizer sees (Y is dead from here on)
        if constexpr (H16) {
            f16x2 pmax = {(f16)-INFINITY, (f16)-INFINITY}, pmin = {(f16)INFINITY, (f16)INFINITY};
            f16x2 lmax = pmax, lmin = pmin;   // the last row tile on its own: only it can hold padding rows
#pragma unroll
            for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f16x2 pr = fq_mul_to_f16x2(Y[mo][2 * j], Y[mo][2 * j + 1], f32x2{ps, ps});
                    H[H16 ? mo : 0][j] = __builtin_bit_cast(uint32_t, pr);
                    if (mo < MT - 1) {
                        pmax = fq_pk_max(pmax, pr);
                        pmin = fq_pk_min(pmin, pr);
                    } else {
                        lmax = fq_pk_max(lmax, pr);
                        lmin = fq_pk_min(lmin, pr);
                    }
                }
            if (((MT - 1) * 32 + c) < M) {
                pmax = fq_pk_max(pmax, lmax);
                pmin = fq_pk_min(pmin, lmin);
            }
            if (MT > 1 || ((MT - 1) * 32 + c) < M) {
                vmax = fmaxf((float)pmax[0], (float)pmax[1]);
                vmin = fminf((float)pmin[0], (float)pmin[1]);
            }
        } else {
            if (out.post_scale != 0.0f) {
#pragma unroll
                for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float p = Y[mo][r] * ps;
                        asm volatile("" : "+v"(p));   // an fp32 VALUE (no fusion with a later rounding)
                        Y[mo][r] = p;
                    }
            }
            if (out.rt_flags & FQ_ROUND_Y_F16) {
#pragma unroll
                for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Y[mo][r] = (float)(T)Y[mo][r];
            }
            float pmx[MT], pmn[MT];   // one independent max3 / min3 chain per tile
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) {
                const f32x16& t = Y[mo];
                float a = FqMaxOp()(t[0], t[1]), b = FqMinOp()(t[0], t[1]);
#pragma unroll
                for (int r = 2; r < 16; r += 2) {
                    a = fq_max3(a, t[r], t[r + 1]);
                    b = fq_min3(b, t[r], t[r + 1]);
                }
                if (ODDN && nval < 16) {   // the half-wave N cuts: its first nval columns only (N % 4 == 0: whole pairs)
                    a = -INFINITY;
                    b = INFINITY;
#pragma unroll
                    for (int r = 0; r < 16; r += 2)
                        if (r < nval) {
                            a = fq_max3(a, t[r], t[r + 1]);
                            b = fq_min3(b, t[r], t[r + 1]);
                        }
                }
                const bool ok = nvalid && (mo < MT - 1 || (mo * 32 + c) < M);   // (only the last row tile can hold padding rows)
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
            red[8 + wq] = vmin;
        }

        // ================= phase C: the token's extrema, scale, quantiser, pack, stores =================
        TILES_MEET()   // B|C: the partial extrema are in LDS
        {
            float a = red[0], b = red[8];
#pragma unroll
            for (int w = 1; w < NT; ++w) {
                a = fmaxf(a, red[w]);
                b = fminf(b, red[8 + w]);
            }
            vmax = fq_uniform_f32(a);   // (wave-uniform by construction; told to the compiler: fq_kron_duo.hip)
            vmin = fq_uniform_f32(b);
        }
        bool waited = false;
        if constexpr (H16) {
            if (out.y != nullptr) {   // the rounded transform itself (wave-uniform): row m' = 32 mo + c, 16 consecutive n' = 32 bytes per lane
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the next token's DMA: see below — with no clip set this is the only wait)
                waited = true;
                if (!(TILES_ABL & 8)) {
                    T* ytok = reinterpret_cast<T*>(out.y) + tok * ((int64_t)M * N) + (h * NT * 16 + wq * 16);
#pragma unroll
                    for (int mo = 0; mo < MT; ++mo)
                        if ((mo * 32 + c) < M) {
                            u32x4* dst = reinterpret_cast<u32x4*>(ytok + (mo * 32 + c) * N);
                            const uint32_t(&hv)[8] = H[H16 ? mo : 0];
                            dst[0] = u32x4{hv[0], hv[1], hv[2], hv[3]};
                            dst[1] = u32x4{hv[4], hv[5], hv[6], hv[7]};
                        }
                }
            }
        }
        for (int ci = 0; ci < out.n_clips; ++ci) {
            float sig_max, sig_min;
            fq_token_sigs(out, ci, tok, gcur, sig_max, sig_min);
            const float scale = H16 ? fq_token_scale<FQ_QUANT_F16>(vmax, vmin, sig_max, sig_min, out.rt_flags)
                                    : fq_token_scale<0, T>(vmax, vmin, sig_max, sig_min, out.rt_flags);
            const float inv = fq_uniform_f32(fq_fast_inv(scale));
            const bool magic = H16 || fq_magic_ok(vmax, vmin, inv);
            const bool clampq = fq_needs_clamp(vmax, vmin, inv);
            const FqH16Recip rc = H16 ? fq_h16_recip(scale) : FqH16Recip{0.0f, 0.0f};
            uint2 pk[MT];
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) {
                if constexpr (H16) {
                    const uint32_t(&hv)[8] = H[H16 ? mo : 0];
                    if (clampq) {
                        pk[mo].x = fq_quant8_h16<true>(hv[0], hv[1], hv[2], hv[3], rc);
                        pk[mo].y = fq_quant8_h16<true>(hv[4], hv[5], hv[6], hv[7], rc);
                    } else {
                        pk[mo].x = fq_quant8_h16<false>(hv[0], hv[1], hv[2], hv[3], rc);
                        pk[mo].y = fq_quant8_h16<false>(hv[4], hv[5], hv[6], hv[7], rc);
                    }
                } else if (TILES_ABL & 1) {
                    pk[mo] = uint2{__builtin_bit_cast(uint32_t, Y[mo][0]), __builtin_bit_cast(uint32_t, Y[mo][8])};
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
                    if (d0m)   // rare: an ambiguous digit somewhere in the wave -> the true division for this dword
                        pk[mo].x = fq_pack8(fq_qexact(yv[0], scale), fq_qexact(yv[1], scale), fq_qexact(yv[2], scale), fq_qexact(yv[3], scale),
                                            fq_qexact(yv[4], scale), fq_qexact(yv[5], scale), fq_qexact(yv[6], scale), fq_qexact(yv[7], scale));
                    if (d1m)
                        pk[mo].y = fq_pack8(fq_qexact(yv[8], scale), fq_qexact(yv[9], scale), fq_qexact(yv[10], scale), fq_qexact(yv[11], scale),
                                            fq_qexact(yv[12], scale), fq_qexact(yv[13], scale), fq_qexact(yv[14], scale), fq_qexact(yv[15], scale));
                }
            }
            // the DMA of the group's next token (requested at the start of phase B) is waited for HERE, in front of the stores:
            // the counter then only holds that request, and the stores are never waited for
            if (!waited) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            waited = true;
            if (!(TILES_ABL & 8)) {
                uint8_t* qtok = out.q[ci] + tok * ((int64_t)M * (N / 2)) + (h * NT * 16 + wq * 16) / 2;
#pragma unroll
                for (int mo = 0; mo < MT; ++mo)
                    if (nvalid && (mo * 32 + c) < M) {
                        uint8_t* dst = qtok + (mo * 32 + c) * (N / 2);
                        if (!ODDN) *reinterpret_cast<uint2*>(dst) = pk[mo];
                        else if (nval == 16) *reinterpret_cast<tl_uint2_a2*>(dst) = pk[mo];   // (row pitch N / 2 = 74 bytes: 2-byte aligned)
                        else {   // nval = 4, 8 or 12 columns: 2, 4 or 6 bytes
                            if (nval >= 4) *reinterpret_cast<uint16_t*>(dst) = (uint16_t)pk[mo].x;
                            if (nval >= 8) *reinterpret_cast<uint16_t*>(dst + 2) = (uint16_t)(pk[mo].x >> 16);
                            if (nval >= 12) *reinterpret_cast<uint16_t*>(dst + 4) = (uint16_t)pk[mo].y;
                        }
                    }
                // (FQ_RATIO_POST — deploy.nn.Quantizer(lac=False) behind a Hadamard rotation: no zero guard, an all-zero token stores scale 0
                //  as fq_rowquant_f16 does, deploy/nn/quantization.py:30)
                if (wq == 0 && lane == 0)
                    reinterpret_cast<T*>(out.scale[ci])[tok] = (H16 && (out.rt_flags & FQ_RATIO_POST) && vmax == 0.0f && vmin == 0.0f) ? (T)0.0f : (T)scale;
            }
        }
        k = knext;
    }
}

template <int MT, int NT, int N, int GROUPS, int LKS, bool RS, typename T, bool H16 = false>
int launch_tiles_t(const T* x, const uint4* ws, int64_t rows, int M, const FqQuantOut& out, int n_cu, hipStream_t stream) {
    typedef TilesGeom<MT, NT, N, GROUPS, LKS> G;
    static_assert(G::LDS <= 160 * 1024, "LDS budget");
    int64_t blocks = (rows + GROUPS - 1) / GROUPS;
    if (blocks > n_cu) blocks = n_cu;   // one persistent workgroup per CU
    if (blocks < 1) blocks = 1;
    const int64_t tpb = (rows + blocks - 1) / blocks;
    hipLaunchKernelGGL((fq_kron_tiles_kernel<MT, NT, N, GROUPS, LKS, RS, T, H16>), dim3((unsigned)blocks), dim3(G::THREADS), 0, stream, x, ws,
                       rows, tpb, M, out);
    return (int)hipGetLastError();
}
template <int MT, int NT, int N, int GROUPS, int LKS, bool RS>
int launch_tiles(bool is_bf16, const f16* x, const uint4* ws, int64_t rows, int M, const FqQuantOut& out, int n_cu, hipStream_t stream) {
    return is_bf16 ? launch_tiles_t<MT, NT, N, GROUPS, LKS, RS, bf16>((const bf16*)x, ws, rows, M, out, n_cu, stream)
                   : launch_tiles_t<MT, NT, N, GROUPS, LKS, RS, f16>(x, ws, rows, M, out, n_cu, stream);
}

}  // namespace

// Returns -1000 when the shape / output set is not one this kernel covers (the caller goes on to the workgroup-per-token kernel).
// ws: fragment workspace already filled by fq_kron_prepare_kernel (rfrag [NT][KS1][64], lfrag [2MT][MT][64]).
// flags & FQ_DT_BF16: bf16 activations and factors (the same geometry; bf16 MFMA and rounding points). On bf16 this kernel also takes
// N = 128 (112 x 128, 86 x 128: fq_kron_trio.hip is fp16-only) — three groups of four waves, as there.
int fq_launch_kron_tiles(int flags, const f16* x, const void* ws, const f16* diag, int64_t rows, int M, int N,
                         const FqQuantOut& out, int n_cu, hipStream_t stream) {
    const bool b = (flags & FQ_DT_BF16) != 0;
    flags &= ~FQ_DT_BF16;
    if (diag != nullptr || (out.rt_flags & FQ_GROUP128)) return -1000;
    const uint4* w = reinterpret_cast<const uint4*>(ws);
    const int lks = (M + 15) >> 4;
#ifndef TILES_N64
#define TILES_N64 1   // 0: the tall pairs stay with fq_kron_tall.hip (A/B builds)
#endif
    if (TILES_N64 && N == 64 && M > 64 && M <= 192 && !b && out.ws_group_stride == 0) {
        // (round 5) the TALL pairs — Hadamard rotations as a Kronecker launch — on two-wave token groups. Output sets: packed with the fp32
        // quantiser, or the fp16 Quantizer arithmetic (H16: packed and / or the rounded transform, the launches of ops.hadamard_quant /
        // hadamard_quantizer / hadamard at 11008, 8960, 5120 ...); the fp32 quantiser WITH the transform (a test-only set) stays with fq_kron_tall.hip
        const int ct = flags & FQ_CT_MASK, cq = ct & ~FQ_OUT_TRANSFORM;
        const bool yout = (ct & FQ_OUT_TRANSFORM) != 0;
        const bool yonly = yout && (cq & ~FQ_QUANT_F16) == 0 && out.n_clips == 0;
        const bool h16 = yonly || (cq == (FQ_OUT_PACKED | FQ_QUANT_F16) && (flags & FQ_ROUND_Y_F16));
        if (yout && (out.y == nullptr || !h16)) return -1000;
        if (!h16 && cq != FQ_OUT_PACKED) return -1000;
        // (measured, profiles/r05_tall_on_tiles.txt) the rotation ALONE (no clip set: 4 d bytes per token, 32-byte pieces of 128-byte rows from every
        // lane) is slower here than on the row-split kernel at six row tiles — 172 x 64: 180 against 171 us — and level or faster below
        if (yonly && M > 160) return -1000;
        FqQuantOut o2 = out;
        if (!yout) o2.y = nullptr;   // (the kernel writes the transform whenever out.y is given)
        const int MT = (M + 31) / 32;
#define FQ_T64(MT_, LKS_, G_)                                                                                               \
    if (MT == MT_ && lks == LKS_)                                                                                          \
        return h16 ? launch_tiles_t<MT_, 2, 64, G_, LKS_, false, f16, true>(x, w, rows, M, o2, n_cu, stream)               \
                   : launch_tiles_t<MT_, 2, 64, G_, LKS_, false, f16, false>(x, w, rows, M, o2, n_cu, stream);
        // four groups of two waves where 4 tokens + the L image fit 160 KB (a token: 2 LKS KB, the image: LKS x MT KB); 172 x 64 = 88 + 66 KB
#ifndef TILES_G64
#define TILES_G64 4   // (measurement knob) token groups of the 172 x 64 launch
#endif
        FQ_T64(3, 5, 4) FQ_T64(3, 6, 4) FQ_T64(4, 7, 4) FQ_T64(4, 8, 4) FQ_T64(5, 9, 4) FQ_T64(5, 10, 4) FQ_T64(6, 11, TILES_G64) FQ_T64(6, 12, 3)
#undef FQ_T64
        return -1000;
    }
    if ((flags & FQ_CT_MASK) != FQ_OUT_PACKED) return -1000;
    if (N == 112 && M > 64 && M <= 96) {   // 80 x 112: four groups of four waves
        return lks == 5 ? launch_tiles<3, 4, 112, TILES_G112, 5, false>(b, x, w, rows, M, out, n_cu, stream)
                        : launch_tiles<3, 4, 112, TILES_G112, 6, false>(b, x, w, rows, M, out, n_cu, stream);
    }
    if (N == 144 && M > 96 && M <= 128) {  // 128 x 144: two groups of five waves
        return lks == 7 ? launch_tiles<4, 5, 144, TILES_G144, 7, TILES_RS144>(b, x, w, rows, M, out, n_cu, stream)
                        : launch_tiles<4, 5, 144, TILES_G144, 8, TILES_RS144>(b, x, w, rows, M, out, n_cu, stream);
    }
    if (N == 192 && M > 128 && M <= 144) { // 144 x 192: two groups of six waves, R streamed; 160 KB hold nine K-steps of L
        return launch_tiles<5, 6, 192, 2, 9, true>(b, x, w, rows, M, out, n_cu, stream);
    }
    if (N == 148 && M > 96 && M <= 128 && !(M & 1)) {   // 128 x 148 (18944, Qwen2.5-7B ffn): two groups of five waves, rows of 296 bytes
        return lks == 7 ? launch_tiles<4, 5, 148, 2, 7, false>(b, x, w, rows, M, out, n_cu, stream)
                        : launch_tiles<4, 5, 148, 2, 8, false>(b, x, w, rows, M, out, n_cu, stream);
    }
    if (N == 176 && M > 160 && M <= 192) { // 168 x 176 (29568, Qwen2.5-72B ffn): ONE group of six waves (two tokens + the L image would need 184 KB);
                                           // the next token's DMA still runs under GEMM 2 and the quantiser
        return lks == 11 ? launch_tiles<6, 6, 176, 1, 11, false>(b, x, w, rows, M, out, n_cu, stream)
                         : launch_tiles<6, 6, 176, 1, 12, false>(b, x, w, rows, M, out, n_cu, stream);
    }
    if (N == 128 && M > 64 && M <= 96) {    // 86 x 128 (11008, Llama-2-7B ffn), fp16 and bf16: 123 us where fq_kron_trio.hip's MT = 3 build takes 145
        return lks == 5 ? launch_tiles<3, 4, 128, 3, 5, false>(b, x, w, rows, M, out, n_cu, stream)
                        : launch_tiles<3, 4, 128, 3, 6, false>(b, x, w, rows, M, out, n_cu, stream);
    }
#ifdef TILES_G128X4   // measurement: 112 x 128 on FOUR groups of four waves (R streamed; 28 KB per token without zero rows + 28 KB of L = 140 KB)
    if (N == 128 && M > 96 && M <= 112)
        return launch_tiles<4, 4, 128, 4, 7, true>(b, x, w, rows, M, out, n_cu, stream);
#endif
    if (b && N == 128 && M > 96 && M <= 128) {   // bf16 112 x 128: three groups of four waves (fp16: fq_kron_trio.hip, the same speed)
        return lks == 7 ? launch_tiles_t<4, 4, 128, 3, 7, false, bf16>((const bf16*)x, w, rows, M, out, n_cu, stream)
                        : launch_tiles_t<4, 4, 128, 3, 8, false, bf16>((const bf16*)x, w, rows, M, out, n_cu, stream);
    }
    return -1000;
}
