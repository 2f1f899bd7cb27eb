// fq_kron64.hip — fused Kronecker transform + per-token INT4 quantisation for d = 4096 (M = N = 64).
//
// Replaces, for the Llama-3-8B hidden size, deploy/kernels/kron_matmul.py:24-130 (Triton matmul_kernel,
// non-split branch) and the flat_utils.py:6-17 + quant_utils.py:71-119 op sequence.
//
// Design (gfx950, wave64):
//   * ONE WAVE OWNS ONE TOKEN (a 64x64 fp16 matrix X, 8 KB); a persistent 16-wave workgroup per CU walks
//     the tokens, 16 tokens in flight per CU.
//   * HBM -> LDS by LDS-DMA (global_load_lds_dwordx4), FULL 128-byte lines: each DMA instruction moves
//     8 whole rows of X (1 KB) — the streaming pattern that reaches ~6 TB/s on this chip. (Loading the
//     MFMA fragments straight from global touches 64 different half-lines per instruction: measured 2x
//     slower, address-processing bound.) The LDS image is lane-linear, so the bank-conflict swizzle is
//     applied to the per-lane SOURCE address (chunk ^= (row>>1)&7 inside a row: same cache line) and
//     mirrored on the ds_read_b128 side -> conflict-free fragment reads.
//   * GEMM 1:  U = X . R          (contraction over n, the contiguous axis of X)      32x32x16, K = n
//     GEMM 2:  Y^T = U^T . L      (contraction over m)                               32x32x16, K = m
//     The C/D fragment of GEMM 1 (lane holds one column n', 16 rows m) is, after fp16 conversion,
//     EXACTLY an A fragment of GEMM 2 (row n', 8 k-slots m per K=16 step) because the order of the
//     contraction index inside an MFMA is free as long as A and B agree. So the intermediate never
//     leaves registers and needs no cross-lane traffic. The fp16 rounding of U is the rounding
//     flat_utils.py:15 performs (torch.matmul output dtype) — path-A arithmetic.
//   * The permutations are absorbed into the one-time gather of the L and R B-operand fragments (LDS,
//     fragment order, 16 KB per workgroup, conflict-free b128 reads): columns of R are permuted so that,
//     after GEMM 2, lane (h, c) holds for output row m' = c the 32 CONSECUTIVE columns n' = 32h..32h+31
//     -> one 16-byte store of packed nibbles per output row.
//   * The token's single LDS buffer is refilled (next token) as soon as its fragments are in registers,
//     i.e. almost a full iteration ahead of use; X occupies VGPRs only during GEMM 1, which keeps the
//     lean variants under 128 VGPRs -> 4 waves/SIMD.
//
// Algorithmic traffic per token: 8192 B read + 2048 B packed + 2 B scale = 10242 B.
#include "fq_common.hpp"
#include "fq_gemm_common.hpp"

namespace {

constexpr int KM = 64, KN = 64, KD = KM * KN;
constexpr int FQ_K64_G128 = 0x10000;     // internal template bits (not ABI flags): the FQ_GROUP128 instantiation,
constexpr int FQ_K64_GROUPED = 0x20000;  // the grouped-launch instantiations (fq_kron_quant_grouped_f16)
constexpr int FQ_K64_MULTI = 0x40000;    // (round 4) the multi-job instantiation (fq_kron_quant_multi_f16): ONE launch over several
                                         // independent (activations, factor pair, outputs) jobs — a layer each, a rank's shard each
// One job of a multi-job launch, device memory, filled by the launcher (fq_capi.hip). Every job runs on `bpj` consecutive
// workgroups; tpb = tokens per workgroup of THIS job (ceil(rows / bpj)).
struct FqKron64Job {
    const void* x;       // [rows, 4096]
    const void* prep;    // the job's 16 KB fragment image (fq_kron_prepare_f16)
    void* q;             // [rows, 2048]
    void* scale;         // [rows]
    int64_t rows;
    int64_t tpb;
};
constexpr int FRAG_BYTES = 16 * 64 * 16;  // 16 fragments x 64 lanes x 16 B
constexpr int TOK_BYTES = KD * 2;         // 8192

// Packed-output stores (16 B per lane, 2 per token): plain write-back stores. Non-temporal, sc1 and write-through forms
// measured 34.3-35.4 us against 34.3 plain (round 2, DESIGN log): nothing to gain, the knob is gone.
__device__ __forceinline__ void store_q16(uint8_t* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
#ifndef FQ_K64_ABLATE
#define FQ_K64_ABLATE 0  // measurement builds only: bit 0 = no MFMA, bit 1 = no quantiser arithmetic, bit 2 = no DMA
#endif
template <typename T>
__device__ __forceinline__ f32x16 mfma32(typename FqVec<T>::x8 a, typename FqVec<T>::x8 b, f32x16 c) {
#if FQ_K64_ABLATE & 1
    c[0] += (float)a[0] + (float)b[0];
    return c;
#else
    return fq_mfma32<T>(a, b, c);
#endif
}

// physical column n' held at (n-tile nt, tile position pos) of GEMM-1's output / GEMM-2's output rows.
__device__ __forceinline__ int nperm(int nt, int pos) {
    return ((pos >> 2) & 1) * 32 + nt * 16 + (pos & 3) + 4 * (pos >> 3);
}

// LDS image of the B-operand fragments, built once per workgroup:
//   frag f in [0, 8):  f = nt*4 + s      R[n = 32h + 8s + j][n' = nperm(nt, c)]          (GEMM 1)
//   frag f in [8,16):  f = 8 + ks*2 + mo L[m = 32(ks>>1) + 16(ks&1) + 8(j>>2) + 4h + (j&3)][m' = 32mo + c]
// stored as [f][lane] 16-byte records.
//
// Workgroup size: 1024 threads (4 waves/SIMD, <= 128 VGPRs) for the lean output sets, 512 otherwise.
template <int FLAGS>
constexpr int kron64_threads() {
    constexpr int outs = ((FLAGS & FQ_OUT_PACKED) ? 1 : 0) + ((FLAGS & FQ_OUT_FAKEQUANT) ? 1 : 0) +
                         ((FLAGS & FQ_OUT_TRANSFORM) ? 1 : 0);
#ifndef FQ_K64_THREADS
#define FQ_K64_THREADS 1024
#endif
    // (round 3) the 16-bit-output sets (no packed output) run 8 waves: see STAGE in the kernel
    if (!(FLAGS & FQ_OUT_PACKED)) return 512;
    return (outs == 1 && !(FLAGS & (FQ_QUANT_F16 | FQ_K64_G128 | FQ_K64_GROUPED))) ? FQ_K64_THREADS : 512;
}

// TRACE (debug build of the same kernel, fq_debug_kron64_trace): lane 0 of every wave accumulates s_memtime
// deltas of the four phases of the token loop into trace[wave*4 .. +3].
#define FQ_TICK(var)                               \
    unsigned long long var = 0;                    \
    if (TRACE) {                                   \
        __builtin_amdgcn_sched_barrier(0);         \
        var = __builtin_amdgcn_s_memtime();        \
        __builtin_amdgcn_sched_barrier(0);         \
    }

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// LDS-DMA of one token into the wave's 8 KB buffer: instruction i moves rows 8i..8i+7 (1 KB contiguous in
// HBM); lane l supplies row 8i + (l>>3), 16-byte chunk (l&7) ^ ((row>>1)&7) and lands in slot 64 i + l.
// Inline asm on purpose: through the builtin hipcc treats the DMA as a store to the shared array and puts
// s_waitcnt vmcnt(0) in front of the next ds_read of ANY part of it (the fragment image), i.e. it waits for
// the prefetch right after issuing it. The asm form is invisible to its counters; completion is waited for by
// the COUNTED s_waitcnt at the top of the token loop. M0 (LDS base of the DMA) is saved/restored because the
// compiler owns it.
#define FQ_DMA_NT "nt"
__device__ __forceinline__ void dma_token(const void* __restrict__ x, int64_t tok, unsigned lds_base, int lane) {
#if FQ_K64_ABLATE & 4
    return;  // measurement build: no HBM reads (compute on whatever is in LDS)
#endif
    // (row>>1)&7 of row 8i + (lane>>3) is (4i + (lane>>4)) & 7 = 4(i&1) + (lane>>4): the chunk swizzle only depends
    // on the parity of i -> two per-lane offsets, everything else is immediates and scalars.
    const unsigned ce = ((lane & 7) ^ (lane >> 4)) << 4;
    const unsigned voff_e = (lane >> 3) * 128 + ce;         // even i
    const unsigned voff_o = (lane >> 3) * 128 + (ce ^ 64);  // odd i: chunk ^ 4
    const unsigned char* base = reinterpret_cast<const unsigned char*>(x) + tok * TOK_BYTES;  // wave-uniform
    // readfirstlane returns int: widen through unsigned or a low half >= 2^31 sign-extends into the high half
    const unsigned lo32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)base);
    const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((size_t)base >> 32));
    const unsigned long long sb0 = (unsigned long long)lo32 | ((unsigned long long)hi32 << 32);
    const unsigned long long sb1 = sb0 + 4096;
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"  // SGPR base / offsets may come straight from v_readfirstlane: 5 wait states before VMEM reads them
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3 " FQ_DMA_NT "\n\t"
        "global_load_lds_dwordx4 %2, %3 offset:1024 " FQ_DMA_NT "\n\t"
        "global_load_lds_dwordx4 %1, %3 offset:2048 " FQ_DMA_NT "\n\t"
        "global_load_lds_dwordx4 %2, %3 offset:3072 " FQ_DMA_NT "\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %4 " FQ_DMA_NT "\n\t"
        "global_load_lds_dwordx4 %2, %4 offset:1024 " FQ_DMA_NT "\n\t"
        "global_load_lds_dwordx4 %1, %4 offset:2048 " FQ_DMA_NT "\n\t"
        "global_load_lds_dwordx4 %2, %4 offset:3072 " FQ_DMA_NT "\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff_e), "v"(voff_o), "s"(sb0), "s"(sb1), "s"(lds_base), "s"(lds_base + 4096)
        : "memory");
}

// (the asm quantiser fq_quant8_two lives in fq_common.hpp: shared with the other Kronecker kernels)

// Quantise + pack one token's fragment: fills the 2 x 4 dwords this lane stores and returns the mask of dwords
// (bit 4*mo + w) in which some lane of the wave saw an ambiguous digit (to be redone with the true division).
template <bool CLAMP>
__device__ __forceinline__ unsigned quant_pack_token(const f32x16 (&Y)[2][2], const float (&inv)[2], uint32_t (&pw)[2][4]) {
    unsigned near = 0;
#pragma unroll
    for (int mo = 0; mo < 2; ++mo) {
        // (inv[0] and inv[1] are the same register unless the launch is FQ_GROUP128: one scale per pair of output rows)
        const float ilo = inv[mo] * 0.999999523162841796875f, ihi = inv[mo] * 1.000000476837158203125f;  // 1 -+ 2^-21
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const f32x16& t = Y[w >> 1][mo];
            const int b = (w & 1) * 8;
            unsigned long long differ;
            pw[mo][w] = fq_quant8<CLAMP>(t[b + 0], t[b + 1], t[b + 2], t[b + 3], t[b + 4], t[b + 5], t[b + 6],
                                          t[b + 7], inv[mo], ilo, ihi, differ);
            near |= differ ? (1u << (4 * mo + w)) : 0u;  // SALU only
        }
    }
    return near;
}

// Take the next token of the workgroup's range (LDS counter) and start its DMA into this wave's (already drained)
// buffer. Placed behind GEMM 1's first K-step: all eight X fragments were requested before the first MFMA, so the
// buffer is free, and the DMA has the whole iteration to land (pulling after the statistics instead measured
// 39.4 us against 35.7, tools/time_variants.py, round 1).
#define FQ_PULL_NEXT(LDS_DST)                                                \
    {                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   \
        int nxt = 0;                                                         \
        if (lane == 0) nxt = (int)atomicAdd(next_slot, 1u);                  \
        nxt = __builtin_amdgcn_readfirstlane(nxt);                           \
        if (nxt < blk_cnt) dma_token(x, blk_base + nxt, (LDS_DST), lane);    \
        next_pulled = nxt;                                                   \
        __builtin_amdgcn_sched_barrier(0);                                   \
    }

template <int FLAGS, bool TRACE = false, typename T = f16>
__global__ __launch_bounds__(kron64_threads<FLAGS>()) void fq_kron64_kernel(const T* __restrict__ x_,
                                                           const T* __restrict__ left,
                                                           const T* __restrict__ right,
                                                           const T* __restrict__ diag,
                                                           int64_t rows_, int64_t tpb_, const uint4* __restrict__ prep_,
                                                           FqQuantOut out_, unsigned long long* __restrict__ trace,
                                                           const FqKron64Job* __restrict__ jobs, int bpj) {
    typedef typename FqVec<T>::x8 X8;  // eight activation / matrix elements = one 16-byte MFMA operand
    // (round 4) FQ_K64_MULTI: the workgroup finds its job — scalar loads of six qwords — and runs it exactly as a launch of its own
    // would, on the job's `bpj` workgroups; everything below sees the job's pointers, row count and a job-relative workgroup index.
    constexpr bool MULTI = (FLAGS & FQ_K64_MULTI) != 0;
    const T* x = x_;
    const uint4* prep = prep_;
    int64_t rows = rows_, tpb = tpb_;
    FqQuantOut out = out_;
    unsigned wg = blockIdx.x;
    if constexpr (MULTI) {
        const unsigned job = blockIdx.x / (unsigned)bpj;
        wg = blockIdx.x - job * (unsigned)bpj;
        const FqKron64Job jb = jobs[job];
        x = static_cast<const T*>(jb.x);
        prep = static_cast<const uint4*>(jb.prep);
        out.q[0] = static_cast<uint8_t*>(jb.q);
        out.scale[0] = static_cast<f16*>(jb.scale);
        rows = jb.rows;
        tpb = jb.tpb;
    }
    constexpr int THREADS = kron64_threads<FLAGS>();
    constexpr int WAVES = THREADS / 64;
    constexpr bool G128 = (FLAGS & FQ_K64_G128) != 0;
    constexpr bool GROUPED = (FLAGS & FQ_K64_GROUPED) != 0;  // per-group clip pairs (own instantiations: the cursor costs
                                                             // the single-output kernels their 128-register fit)
    // output stores a single-clip packed token issues after its DMA (2 x 16 B + the scale): lets the top-of-loop
    // wait be a COUNTED vmcnt that does not also wait for the previous token's stores to reach memory.
    constexpr bool COUNTED_WAIT = (FLAGS & FQ_CT_MASK & ~FQ_IN_RMSNORM) == FQ_OUT_PACKED;
    constexpr int OUT_SET = FLAGS & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT | FQ_OUT_TRANSFORM);
    // (round 3) FULL-LINE stores for the 16-bit outputs (fake-quant / transform: 8 KB per token). Straight from the C/D fragment a
    // lane holds 64 bytes of one output row and a store instruction puts 16 bytes into each of 32 different 128-byte lines:
    // measured (tools/time_variants.py MODE=fq / y, same kernel with and without its stores) 35.6 us without them, 61-65 us with.
    // With STAGE the eight pieces go through the wave's own 8 KB token buffer (XOR-swizzled, conflict-free both ways) and leave
    // as eight 1 KB-contiguous non-temporal stores; the buffer is only free once the X fragments are in registers AND the next
    // token has not been requested, so these instantiations request it after the stores instead of behind GEMM 1's first K-step
    // (the other waves of the CU cover the wait), and they run EIGHT waves per CU, not sixteen: 16384 tokens, fake-quant output
    // 64.0 -> 50.3 us (0.52 -> 0.67 of 8 TB/s), transform only 61 -> 46.3 (0.72), transform + fake-quant 99 -> 71.7 (0.70).
    // Tried and dropped (A/B in one process): 16 waves with staging 63.3 (no gain over 63.5 unstaged: more waves,
    // more requests in flight, lower throughput), a second token buffer per wave with the early request 52.3 (vs 50.3), a
    // counted wait that leaves the stores in flight (no change), plain / sc1 / write-through stores (51.8 / 62.7 / 62.7).
    constexpr bool STAGE = !(FLAGS & FQ_OUT_PACKED) && OUT_SET != 0;

    unsigned long long tr_wait = 0, tr_g1 = 0, tr_g2 = 0, tr_epi = 0;
    const unsigned long long tr_start = TRACE ? __builtin_amdgcn_s_memtime() : 0;
    const unsigned long long tr_start_rt = TRACE ? __builtin_amdgcn_s_memrealtime() : 0;  // 100 MHz, chip-wide
    __shared__ __attribute__((aligned(16))) unsigned char smem[FRAG_BYTES + WAVES * TOK_BYTES + 16];
    unsigned* next_slot = reinterpret_cast<unsigned*>(smem + FRAG_BYTES + WAVES * TOK_BYTES);  // work counter
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int h = lane >> 5;
    const int c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t wave_id = (int64_t)blockIdx.x * WAVES + wave;   // (trace builds only)
    const int64_t n_waves = (int64_t)gridDim.x * WAVES;
    unsigned char* tokbuf = smem + FRAG_BYTES + wave * TOK_BYTES;  // wave-private, wave-uniform address
    const unsigned tok_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)tokbuf);

    // Tokens: workgroup b owns the contiguous range [blk_base, blk_base + blk_cnt) (tpb = tokens per workgroup, from
    // the host: no gridDim / division here, and with the first 16 kernarg dwords preloaded into SGPRs the prologue
    // below starts without a kernarg round trip); its waves PULL tokens from a counter in LDS. (A static
    // tok += n_waves split ties the kernel's duration to the slowest wave: the SIMD arbiter favours older waves,
    // measured 2.8x spread in per-wave loop time.) The first WAVES tokens are handed out statically.
    const int64_t blk_base = (int64_t)wg * tpb;
    const int blk_cnt = (int)(rows - blk_base < tpb ? (rows - blk_base < 0 ? 0 : rows - blk_base) : tpb);
    if (tid == 0) *next_slot = WAVES;
    int slot = wave;

    // ---- prologue: B-operand fragments of R and L (gathered straight from the 2 x 8 KB row-major matrices) and the
    // first token of every wave. The order is what the TRACE build measured as mattering:
    //  (1) every wave issues its 8 small L/R loads, then a bare s_barrier: all of them are in the CU's memory
    //      pipeline before the first DMA. (DMAs of 16 waves are 128 KB of misses per CU; a gather load queued
    //      behind them used to hold the workgroup at the fragment barrier for ~5 us.)
    //  (2) the first DMAs go out in SIMD-slot order (waves 4g..4g+3 sit on the four SIMDs): group 0 at once,
    //      group 1 after it has written its fragments, groups 2, 3 after the fragment barrier plus (g-1) *
    //      FQ_K64_STAGGER * 64 cycles. All 4096 waves asking at once is a 32 MB burst that comes back
    //      interleaved, i.e. every wave waits ~6 us and the SIMDs then run in lock step; ordered, the first
    //      quarter computes after ~2 us and the four waves of a SIMD stay out of phase.
    // The gather loads are inline asm because the compiler cannot see the DMAs in its vmcnt accounting: their
    // completion is waited for by hand (8 younger VMEM ops = the DMA, for group 0).
#ifndef FQ_K64_STAGGER
#define FQ_K64_STAGGER 20
#endif
    uint4* frag = reinterpret_cast<uint4*>(smem);
    constexpr int ITEMS = 16 * 64 / THREADS;  // fragment slots (f, lane') this thread fills: 1 or 2
    static_assert(ITEMS * THREADS == 16 * 64, "fragment image is 1024 x 16 B");
    unsigned gv[ITEMS][8];
    // (round 3) `prep` != nullptr: the caller passed the fragment image fq_kron_prepare_f16 wrote (FQ_WS_PREPARED, 16 KB, the
    // exact LDS image below): ONE coalesced 16-byte load per slot instead of eight 2-byte gathers and their address arithmetic
    // — the matrices are constants of a deployed layer. Same wait accounting (these loads are older than the DMA).
    u32x4 pv[ITEMS];
    const bool prepared = prep != nullptr;   // (kernel-uniform)
    if (prepared) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pv[it]) : "v"(prep + tid + it * THREADS) : "memory");
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        if (prepared) break;
        const int item = tid + it * THREADS;                              // (f, lane') with lane' fastest
        const int f = __builtin_amdgcn_readfirstlane(item >> 6);          // one fragment per wave and item
        const int ln = item & 63, fh = ln >> 5, fc = ln & 31;
#define FQ_LD(j, off) asm volatile("global_load_ushort %0, %1, off offset:" #off : "=v"(gv[it][j]) : "v"(src) : "memory");
        if (f < 8) {  // element j: row + j of R
            const int nt = f >> 2, sk = f & 3;
            const T* src = right + (fh * 32 + sk * 8) * KN + nperm(nt, fc);
            FQ_LD(0, 0) FQ_LD(1, 128) FQ_LD(2, 256) FQ_LD(3, 384) FQ_LD(4, 512) FQ_LD(5, 640) FQ_LD(6, 768) FQ_LD(7, 896)
        } else {      // element j: row + 8 (j>>2) + (j&3) of L
            const int ks = (f - 8) >> 1, mo = (f - 8) & 1;
            const T* src = left + ((ks >> 1) * 32 + 16 * (ks & 1) + 4 * fh) * KM + mo * 32 + fc;
            FQ_LD(0, 0) FQ_LD(1, 128) FQ_LD(2, 256) FQ_LD(3, 384) FQ_LD(4, 1024) FQ_LD(5, 1152) FQ_LD(6, 1280) FQ_LD(7, 1408)
        }
#undef FQ_LD
    }
    const bool have_first = slot < blk_cnt;
    const int grp = wave >> 2;
    unsigned long long tr_args = 0, tr_issue = 0;
    if (TRACE) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_args) : "s"(blk_cnt));  // kernargs are here
    __builtin_amdgcn_s_barrier();  // (1): no memory wait in front of it, only "all gather loads are issued"
    if (have_first && grp == 0) dma_token(x, blk_base + slot, tok_lds, lane);
    if (TRACE) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_issue));  // prologue loads issued (group 0)
    // the remaining kernel arguments, fetched while the gather is in flight instead of one miss at a time later
    asm volatile("" : : "s"(out.n_clips), "s"(out.rt_flags), "s"(out.sig_max[0]), "s"(out.sig_min[0]), "s"(out.q[0]),
                 "s"(out.scale[0]));
    // gather data has landed once at most the DMA's 8 ops are outstanding; ties the wait to the registers
#define FQ_GV(it) "+v"(gv[it][0]), "+v"(gv[it][1]), "+v"(gv[it][2]), "+v"(gv[it][3]), "+v"(gv[it][4]), \
                  "+v"(gv[it][5]), "+v"(gv[it][6]), "+v"(gv[it][7])
    // ONE asm statement for both cases (only group 0 has a DMA in flight): separate statements in the arms of
    // an if make the register allocator copy the still-in-flight registers ahead of the wait.
    const int dma_ops = ((FQ_K64_ABLATE & 4) || !have_first || grp != 0) ? 0 : 8;
#define FQ_WAIT_GATHER                                                                                      \
    "s_cmp_eq_u32 %[n], 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(8)\n\ts_branch 2f\n1:\n\ts_waitcnt vmcnt(0)\n2:"
    if (prepared) {
        if (ITEMS == 1) asm volatile(FQ_WAIT_GATHER : "+v"(pv[0]) : [n] "s"(dma_ops) : "scc");
        else asm volatile(FQ_WAIT_GATHER : "+v"(pv[0]), "+v"(pv[ITEMS - 1]) : [n] "s"(dma_ops) : "scc");
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) frag[tid + it * THREADS] = __builtin_bit_cast(uint4, pv[it]);
    } else {
        if (ITEMS == 1) asm volatile(FQ_WAIT_GATHER : FQ_GV(0) : [n] "s"(dma_ops) : "scc");
        else asm volatile(FQ_WAIT_GATHER : FQ_GV(0), FQ_GV(ITEMS - 1) : [n] "s"(dma_ops) : "scc");
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            uint4 v;
            v.x = gv[it][0] | (gv[it][1] << 16);  // global_load_ushort zero-extends
            v.y = gv[it][2] | (gv[it][3] << 16);
            v.z = gv[it][4] | (gv[it][5] << 16);
            v.w = gv[it][6] | (gv[it][7] << 16);
            frag[tid + it * THREADS] = v;
        }
    }
#undef FQ_WAIT_GATHER
#undef FQ_GV
    const unsigned long long tr_gather = TRACE ? __builtin_amdgcn_s_memtime() : 0;
    if (have_first && grp == 1) dma_token(x, blk_base + slot, tok_lds, lane);
    unsigned long long tr_prebar = 0;
    if (TRACE) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_prebar) : : "memory");
    __syncthreads();  // (the compiler knows of no VMEM in flight: this is lgkmcnt(0) + s_barrier, the DMAs stay out)
    const unsigned long long tr_barrier = TRACE ? __builtin_amdgcn_s_memtime() : 0;
    if (have_first && grp >= 2) {
#if FQ_K64_STAGGER
        for (int g = grp - 1; g > 0; --g) __builtin_amdgcn_s_sleep(FQ_K64_STAGGER);
#endif
        dma_token(x, blk_base + slot, tok_lds, lane);
    }

    // fragment-read address of this lane inside the token buffer: row (32 mt + c), chunk (4h + s) ^ ((c>>1)&7)
    const int sw = (c >> 1) & 7;
    const int lane_off = c * KN + h * 32;  // same element offset in HBM (used by diag and by the outputs)
    bool first = true;
    int next_pulled = 0;
    FqGroupCursor gcur;  // grouped launches: the clip pair follows the token's group

#ifndef FQ_K64_HOIST
#define FQ_K64_HOIST 0   // measurement: 1 = the 16 factor fragments live in registers (64 VGPRs; needs FQ_K64_THREADS=512: two waves per SIMD)
#endif
#if FQ_K64_HOIST
    X8 BF[16];
#pragma unroll
    for (int f = 0; f < 16; ++f) BF[f] = __builtin_bit_cast(X8, frag[f * 64 + lane]);
#define FQ_BFRAG(f) BF[f]
#else
#define FQ_BFRAG(f) __builtin_bit_cast(X8, myfrag[(f) * 64])
#endif
    while (slot < blk_cnt) {
        const int64_t tok = blk_base + slot;
        // Launder the lane offset every iteration: otherwise LICM hoists all 16 loop-invariant fragment reads
        // (64 VGPRs) out of the token loop and the register allocator spills them.
        int foff = lane;
        asm volatile("" : "+v"(foff));
        const uint4* myfrag = frag + foff;
        FQ_TICK(c0)
        // the DMA of this token was issued an iteration ago (or in the prologue); younger ops = its 3 stores
        if (COUNTED_WAIT && !first) {
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        first = false;
        FQ_TICK(c1)
        // STAGE: piece w (16 bytes = columns 32 h + 8 w .. + 8) of output row 32 mo + c -> the token buffer, chunk index XOR
        // (row & 7); then the whole 8 KB image leaves in row-linear order, 1 KB per store instruction
        auto stage_put = [&](int mo, int w, u32x4 v) {
            const int r = mo * 32 + c;
            reinterpret_cast<u32x4*>(tokbuf)[r * 8 + ((h * 4 + w) ^ (r & 7))] = v;
        };
        auto stage_flush = [&](void* dst_token) {
            const u32x4* tb = reinterpret_cast<const u32x4*>(tokbuf);
            u32x4* dp = reinterpret_cast<u32x4*>(dst_token) + lane;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const u32x4 v = tb[(8 * i + (lane >> 3)) * 8 + ((lane & 7) ^ (lane >> 3))];
#if !(FQ_K64_ABLATE & 8)
                __builtin_nontemporal_store(v, dp + i * 64);
#else
                asm volatile("" : : "v"(v));
#endif
            }
        };

        u32x4 X[2][4];
        {
            const u32x4* tb = reinterpret_cast<const u32x4*>(tokbuf);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int s = 0; s < 4; ++s) X[mt][s] = tb[(mt * 32 + c) * 8 + ((h * 4 + s) ^ sw)];
        }
        if (FLAGS & FQ_IN_RMSNORM) {
            // deploy.nn.RMSNorm in front of the transform (normalization.py:16-23): the wave's X fragments cover the
            // token exactly once, so sum(x^2) is 64 fmas per lane (four chains) + a wave all-reduce; the normalised
            // values are rounded to fp16 (the module returns fp16) before they enter GEMM 1.
            float ss[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const X8 v = __builtin_bit_cast(X8, X[mt][s]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) ss[j & 3] = __builtin_fmaf((float)v[j], (float)v[j], ss[j & 3]);
                }
            const float tot = fq_wave_sum((ss[0] + ss[1]) + (ss[2] + ss[3]));
            const float rinv = __builtin_amdgcn_rsqf(tot / (float)KD + out.rms_eps);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    X8 v = __builtin_bit_cast(X8, X[mt][s]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fq_mul_to<T>((float)v[j], rinv);
                    X[mt][s] = __builtin_bit_cast(u32x4, v);
                }
        }
        if (diag != nullptr) {  // x * diag_scale, rounded to fp16 (trans_utils.py:86-90)
            const uint4* dp = reinterpret_cast<const uint4*>(diag + lane_off);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    X8 dv = __builtin_bit_cast(X8, dp[mt * (32 * KN / 8) + s]);
                    X[mt][s] = __builtin_bit_cast(u32x4, __builtin_bit_cast(X8, X[mt][s]) * dv);
                }
        }

        // ---- GEMM 1: U[mt][nt] = X(mt,:) . R(:, nt); four independent accumulator chains ----
        // Matrix phases run at raised priority: an MFMA needs one issue slot per 32 cycles, so letting it win the
        // arbitration keeps the matrix pipe fed while the other waves' VALU work (quantiser) fills the gaps.
        __builtin_amdgcn_s_setprio(2);
        f32x16 U[2][2];
        U[0][0] = f32x16{0}; U[0][1] = f32x16{0}; U[1][0] = f32x16{0}; U[1][1] = f32x16{0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const X8 b0 = FQ_BFRAG(0 * 4 + s);
            const X8 b1 = FQ_BFRAG(1 * 4 + s);
            U[0][0] = mfma32<T>(__builtin_bit_cast(X8, X[0][s]), b0, U[0][0]);
            U[1][0] = mfma32<T>(__builtin_bit_cast(X8, X[1][s]), b0, U[1][0]);
            U[0][1] = mfma32<T>(__builtin_bit_cast(X8, X[0][s]), b1, U[0][1]);
            U[1][1] = mfma32<T>(__builtin_bit_cast(X8, X[1][s]), b1, U[1][1]);
            if (s == 0 && !STAGE) FQ_PULL_NEXT(tok_lds)  // (all eight X fragments were requested before the first MFMA)
        }

        // ---- fp16 rounding of U (flat_utils.py:15); C fragment -> A fragment of GEMM 2, no data movement ----
        X8 Uh[2][4];  // [nt][ks]
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) Uh[nt][ks][j] = (T)U[ks >> 1][nt][(ks & 1) * 8 + j];

        FQ_TICK(c2)
        // ---- GEMM 2: Y^T(nt, mo) = U(:, nt)^T . L(:, mo) ----
        f32x16 Y[2][2];  // [nt][mo]: Y^T[n' = 32h + 16nt + r][m' = 32mo + c]
        Y[0][0] = f32x16{0}; Y[0][1] = f32x16{0}; Y[1][0] = f32x16{0}; Y[1][1] = f32x16{0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const X8 b0 = FQ_BFRAG(8 + ks * 2 + 0);
            const X8 b1 = FQ_BFRAG(8 + ks * 2 + 1);
            Y[0][0] = mfma32<T>(Uh[0][ks], b0, Y[0][0]);
            Y[1][0] = mfma32<T>(Uh[1][ks], b0, Y[1][0]);
            Y[0][1] = mfma32<T>(Uh[0][ks], b1, Y[0][1]);
            Y[1][1] = mfma32<T>(Uh[1][ks], b1, Y[1][1]);
        }
        __builtin_amdgcn_s_setprio(0);

        if (out.rt_flags & FQ_ROUND_Y_F16) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mo = 0; mo < 2; ++mo)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Y[nt][mo][r] = (float)(T)Y[nt][mo][r];
        }

        if (FLAGS & FQ_OUT_TRANSFORM) {
#pragma unroll
            for (int mo = 0; mo < 2; ++mo) {
                uint4* yp = reinterpret_cast<uint4*>(out.y + tok * KD + (mo * 32 + c) * KN + h * 32);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        X8 v;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (T)Y[nt][mo][w * 8 + e];
                        if (STAGE) stage_put(mo, nt * 2 + w, __builtin_bit_cast(u32x4, v));
                        else yp[nt * 2 + w] = __builtin_bit_cast(uint4, v);
                    }
            }
            if (STAGE) stage_flush(reinterpret_cast<T*>(out.y) + tok * KD);
        }

        if (FLAGS & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT)) {
            // four independent partial chains each (v_max3/v_min3), then a wave all-reduce
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
            FQ_TICK(c3)

            for (int ci = 0; ci < out.n_clips; ++ci) {
                float sig_max = out.sig_max[ci], sig_min = out.sig_min[ci];
                if (GROUPED) fq_token_sigs(out, ci, tok, gcur, sig_max, sig_min);
                const float scale = fq_token_scale<FLAGS, T>(vmax, vmin, sig_max, sig_min, out.rt_flags);
                // value of element e (0..7) of dword w (0..3) of output row mo: Y^T[n' = 32h + 8w + e]
#define FQ_YV(mo, w, e) Y[(w) >> 1][mo][((w) & 1) * 8 + (e)]
                if (FLAGS & FQ_OUT_PACKED) {
                    // FQ_GROUP128 (own instantiation): one scale per 128 consecutive elements = output rows (2j, 2j+1),
                    // i.e. lanes c in {2j, 2j+1}, both column halves h, per output-row tile mo; [rows, 32] scales
                    float scl[2] = {scale, scale};
                    bool g_magic = true, g_clamp = false;
                    if (G128) {
#pragma unroll
                        for (int mo = 0; mo < 2; ++mo) {
                            float a = FqMaxOp()(Y[0][mo][0], Y[0][mo][1]), b = FqMinOp()(Y[0][mo][0], Y[0][mo][1]);
#pragma unroll
                            for (int r = 2; r < 32; r += 2) {
                                a = fq_max3(a, Y[r >> 4][mo][r & 15], Y[r >> 4][mo][(r & 15) + 1]);
                                b = fq_min3(b, Y[r >> 4][mo][r & 15], Y[r >> 4][mo][(r & 15) + 1]);
                            }
                            const float gmax = fq_group4_reduce(a, FqMaxOp()), gmin = fq_group4_reduce(b, FqMinOp());
                            scl[mo] = fq_token_scale<FLAGS, T>(gmax, gmin, sig_max, sig_min, out.rt_flags);
                            const float gi = fq_fast_inv(scl[mo]);
                            g_magic = g_magic && !__any(!fq_magic_ok(gmax, gmin, gi));
                            g_clamp = g_clamp || __any(fq_needs_clamp(gmax, gmin, gi));
                            if (h == 0 && !(c & 1)) reinterpret_cast<T*>(out.scale[ci])[tok * (KD / 128) + ((mo * 32 + c) >> 1)] = (T)scl[mo];
                        }
                    } else if (lane == 0) {
                        reinterpret_cast<T*>(out.scale[ci])[tok] = (T)scale;
                    }
                    uint32_t pw[2][4];
                    if (FLAGS & FQ_QUANT_F16) {
#pragma unroll
                        for (int mo = 0; mo < 2; ++mo)
#pragma unroll
                            for (int w = 0; w < 4; ++w) {
                                uint32_t d = 0;
#pragma unroll
                                for (int e = 0; e < 8; ++e)
                                    d |= (uint32_t)(fq_quant1<FLAGS, T>(FQ_YV(mo, w, e), scl[mo]) & 15) << (4 * e);
                                pw[mo][w] = d;
                            }
                    } else {
                        const float inv[2] = {fq_fast_inv(scl[0]), G128 ? fq_fast_inv(scl[1]) : fq_fast_inv(scl[0])};
                        // bit (4*mo + w) of near: some lane's dword has an ambiguous digit
                        unsigned near;
#if FQ_K64_ABLATE & 2
                        near = 0;
#pragma unroll
                        for (int mo = 0; mo < 2; ++mo)
#pragma unroll
                            for (int w = 0; w < 4; ++w)
                                pw[mo][w] = __builtin_bit_cast(unsigned, FQ_YV(mo, w, 0)) ^ __builtin_bit_cast(unsigned, FQ_YV(mo, w, 1)) ^
                                            __builtin_bit_cast(unsigned, FQ_YV(mo, w, 2)) ^ __builtin_bit_cast(unsigned, FQ_YV(mo, w, 3)) ^
                                            __builtin_bit_cast(unsigned, FQ_YV(mo, w, 4)) ^ __builtin_bit_cast(unsigned, FQ_YV(mo, w, 5)) ^
                                            __builtin_bit_cast(unsigned, FQ_YV(mo, w, 6)) ^ __builtin_bit_cast(unsigned, FQ_YV(mo, w, 7)) ^
                                            __builtin_bit_cast(unsigned, inv[mo]);
#else
                        if (G128 ? !g_magic : !fq_magic_ok(vmax, vmin, inv[0])) {
                            near = 0xffu;  // quotients too large for the magic-number rounding: true division throughout
                        } else if (G128 ? g_clamp : fq_needs_clamp(vmax, vmin, inv[0])) {
                            near = quant_pack_token<true>(Y, inv, pw);
                        } else {
                            near = quant_pack_token<false>(Y, inv, pw);
                        }
#endif
                        if (near) {  // rare (~3 % of tokens): redo the flagged dwords with the true division
#pragma unroll
                            for (int mo = 0; mo < 2; ++mo)
#pragma unroll
                                for (int w = 0; w < 4; ++w)
                                    if (near & (1u << (4 * mo + w)))
                                        pw[mo][w] = fq_pack8(
                                            fq_qexact(FQ_YV(mo, w, 0), scl[mo]), fq_qexact(FQ_YV(mo, w, 1), scl[mo]),
                                            fq_qexact(FQ_YV(mo, w, 2), scl[mo]), fq_qexact(FQ_YV(mo, w, 3), scl[mo]),
                                            fq_qexact(FQ_YV(mo, w, 4), scl[mo]), fq_qexact(FQ_YV(mo, w, 5), scl[mo]),
                                            fq_qexact(FQ_YV(mo, w, 6), scl[mo]), fq_qexact(FQ_YV(mo, w, 7), scl[mo]));
                        }
                    }
#pragma unroll
                    for (int mo = 0; mo < 2; ++mo)
                        store_q16(out.q[ci] + tok * (KD / 2) + (mo * 32 + c) * (KN / 2) + h * 16,
                                  u32x4{pw[mo][0], pw[mo][1], pw[mo][2], pw[mo][3]});
                }
                if (FLAGS & FQ_OUT_FAKEQUANT) {
                    X8 fv[2][4];
                    if (FLAGS & FQ_QUANT_F16) {
#pragma unroll
                        for (int mo = 0; mo < 2; ++mo)
#pragma unroll
                            for (int w = 0; w < 4; ++w)
#pragma unroll
                                for (int e = 0; e < 8; ++e)
                                    fv[mo][w][e] = fq_dequant1<FLAGS, T>(fq_quant1<FLAGS, T>(FQ_YV(mo, w, e), scale), scale);
                    } else {
                        const float inv = fq_fast_inv(scale);
                        const bool magic = fq_magic_ok(vmax, vmin, inv);
                        const bool clampq = fq_needs_clamp(vmax, vmin, inv);
                        // single-width asm quantiser (fq_fake8, fq_common.hpp), each 16-byte piece stored at once (keeping a
                        // token's eight pieces costs 32 VGPRs this kernel lacks), ONE exactness vote per token
                        float dmax = 0.0f;
#if FQ_K64_ABLATE & 8
#define FQ_FQ_STORE(mo, w, v) asm volatile("" : : "v"(v));
#else
#define FQ_FQ_STORE(mo, w, v)                                                                                              \
    if (STAGE) stage_put(mo, w, v);                                                                                        \
    else reinterpret_cast<u32x4*>(reinterpret_cast<T*>(out.fq[ci]) + tok * KD + ((mo) * 32 + c) * KN + h * 32)[w] = (v);
#endif
                        if (magic) {
#pragma unroll
                            for (int mo = 0; mo < 2; ++mo)
#pragma unroll
                                for (int w = 0; w < 4; ++w) {
                                    const u32x4 o = clampq
                                        ? fq_fake8<true, T>(FQ_YV(mo, w, 0), FQ_YV(mo, w, 1), FQ_YV(mo, w, 2), FQ_YV(mo, w, 3), FQ_YV(mo, w, 4),
                                                            FQ_YV(mo, w, 5), FQ_YV(mo, w, 6), FQ_YV(mo, w, 7), inv, scale, dmax)
                                        : fq_fake8<false, T>(FQ_YV(mo, w, 0), FQ_YV(mo, w, 1), FQ_YV(mo, w, 2), FQ_YV(mo, w, 3), FQ_YV(mo, w, 4),
                                                             FQ_YV(mo, w, 5), FQ_YV(mo, w, 6), FQ_YV(mo, w, 7), inv, scale, dmax);
                                    FQ_FQ_STORE(mo, w, o)
                                }
                        }
                        if (!magic || fq_wave_needs_exact(dmax)) {  // rare (~3 % of tokens): the whole token with the true division
#pragma unroll
                            for (int mo = 0; mo < 2; ++mo)
#pragma unroll
                                for (int w = 0; w < 4; ++w) {
                                    X8 o;
#pragma unroll
                                    for (int e = 0; e < 8; ++e) o[e] = fq_fake<T>(scale, fq_qexact(FQ_YV(mo, w, e), scale));
                                    FQ_FQ_STORE(mo, w, __builtin_bit_cast(u32x4, o))
                                }
                        }
#undef FQ_FQ_STORE
                        if (STAGE) stage_flush(reinterpret_cast<T*>(out.fq[ci]) + tok * KD);
                    }
                    if (FLAGS & FQ_QUANT_F16) {
#pragma unroll
                        for (int mo = 0; mo < 2; ++mo) {
                            uint4* fp =
                                reinterpret_cast<uint4*>(out.fq[ci] + tok * KD + (mo * 32 + c) * KN + h * 32);
#pragma unroll
                            for (int w = 0; w < 4; ++w) {
                                if (STAGE) stage_put(mo, w, __builtin_bit_cast(u32x4, fv[mo][w]));
                                else fp[w] = __builtin_bit_cast(uint4, fv[mo][w]);
                            }
                        }
                        if (STAGE) stage_flush(reinterpret_cast<T*>(out.fq[ci]) + tok * KD);
                    }
                }
#undef FQ_YV
            }
            FQ_TICK(c4)
            if (TRACE) {
                tr_wait += c1 - c0;
                tr_g1 += c2 - c1;
                tr_g2 += c3 - c2;
                tr_epi += c4 - c3;
            }
        }
        if (STAGE) FQ_PULL_NEXT(tok_lds)  // (the buffer served as the output stage: the next token is requested only now)
        slot = next_pulled;
    }
    if (TRACE && lane == 0) {
        trace[n_waves * 4 + wave_id * 10 + 0] = tr_start;                       // absolute stamps: start, end,
        trace[n_waves * 4 + wave_id * 10 + 1] = __builtin_amdgcn_s_memtime();   // fragment gather done, barrier passed,
        trace[n_waves * 4 + wave_id * 10 + 2] = tr_gather;                      // start / end on the 100 MHz clock
        trace[n_waves * 4 + wave_id * 10 + 3] = tr_barrier;
        trace[n_waves * 4 + wave_id * 10 + 4] = tr_start_rt;
        trace[n_waves * 4 + wave_id * 10 + 5] = __builtin_amdgcn_s_memrealtime();
        trace[n_waves * 4 + wave_id * 10 + 6] = tr_args;                        // kernargs loaded / prologue loads issued
        trace[n_waves * 4 + wave_id * 10 + 7] = tr_issue;
        trace[n_waves * 4 + wave_id * 10 + 8] = tr_prebar;                      // own fragment writes done, at the barrier
        trace[wave_id * 4 + 0] = tr_wait;
        trace[wave_id * 4 + 1] = tr_g1;
        trace[wave_id * 4 + 2] = tr_g2;
        trace[wave_id * 4 + 3] = tr_epi;
    }
}


// =====================================================================================================================
// (round 6) THE TRANSFORM AS THE GEMM'S PROLOGUE, decode regime: [RMSNorm +] 64 x 64 Kronecker transform + per-token INT4 quantisation +
// up to four Linear4bit projections of the quantised tokens (q / k / v, or up / gate: their own clip pair, weights, scales, bias, output)
// as ONE launch for M <= 16 tokens — deploy/nn/online_trans.py + quantization.py in front of deploy/nn/linear.py:40-54, the launch pair
// fq_[rmsnorm_]kron_quant_f16 -> fq_int4_skinny_linear_multi_f16 of the captured decode step. At one to sixteen tokens both launches sit on
// the latency floor of a small dispatch (5.0 + 5.6 us for q / k / v at one token, 5.0 + 11.8 for up / gate: profiles/r05_decode_layer.txt)
// whose parts — kernel arguments, fragment image, token, weights — are dependent round trips to memory. Here every persistent workgroup
//   1. requests the fragment image, the tokens and its first TWO feature tiles of weights at once,
//   2. transforms and quantises the M tokens itself (a wave per token, two rounds for M > 8: the code of fq_kron64_kernel) with the clip
//      pair of ITS problem while the weights are in flight — 192 .. 256 workgroups repeat the same 1 MFLOP per token, nobody waits for a
//      producer launch — and leaves the packed digits and the scales in LDS,
//   3. walks its feature tiles (32 features, K split over the 8 waves as in fq_gemm_i4_skinny_kernel: same weight image, same MFMAs,
//      same reduction through ds_add, same sym_dequant arithmetic) with the tile after next requested under the current one.
// The quantised activations are never written to memory. Bit-identical to the two launches by construction (tests/test_gpu_fused_decode.py).
// Every VMEM operation of the kernel is inline asm: the compiler's own s_waitcnt bookkeeping would otherwise turn each wait for a scale
// or a store into a wait for the prefetched weights (vmcnt is one in-order counter).
// =====================================================================================================================
constexpr int KL_WAVES = 8, KL_THREADS = KL_WAVES * 64, KL_MAXM = 16;
constexpr int KL_KB = KD / 64;                    // 64-k blocks of a weight row tile (K = 4096)
constexpr int KL_BPW = KL_KB / KL_WAVES;          // blobs per wave and tile: 8
struct K64LinProblems {
    const uint4* wimg[4];     // weight images (fq_int4_to_frag: blob (32-feature tile, 64-k block) = 64 lanes x 16 B)
    f16* y[4];                // [M, N[p]]
    const f16* scol[4];       // [N[p]] weight scales
    const f16* bias[4];       // [N[p]] or nullptr
    float sig_max[4], sig_min[4];
    int N[4];                 // N[p] % 32 == 0
    int wg0[5];               // first workgroup of problem p; [n] = the grid
    int n;
};
typedef int kl_i32x16 __attribute__((ext_vector_type(16)));

// Loads whose completion is waited for by hand. "+v": the destination is the register the variable ALREADY lives in — with "=v" the
// register allocator is free to load into a fresh register and copy it into the loop-carried one at the back edge, i.e. to read a register
// whose data has not landed (found in the ISA of the first build: v_mov_b64 of all eight weight fragments in front of the wait).
__device__ __forceinline__ void kl_load16(u32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off nt" : "+v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void kl_load2(unsigned& d, const void* p) { asm volatile("global_load_ushort %0, %1, off" : "+v"(d) : "v"(p) : "memory"); }
// ONE asm statement per wait, the count selected at run time inside it (sel 0..3): separate statements in the arms of an if make the
// register allocator copy the still-in-flight registers ahead of the wait (fq_kron64_kernel's prologue, FQ_WAIT_GATHER).
#define KL_WAIT4(sel, C0, C1, C2, C3, ...)                                                                               \
    asm volatile("s_cmp_eq_u32 %[n], 0\n\ts_cbranch_scc1 5f\n\ts_cmp_eq_u32 %[n], 1\n\ts_cbranch_scc1 6f\n\t"           \
                 "s_cmp_eq_u32 %[n], 2\n\ts_cbranch_scc1 7f\n\ts_waitcnt vmcnt(" #C3 ")\n\ts_branch 9f\n"                \
                 "5:\n\ts_waitcnt vmcnt(" #C0 ")\n\ts_branch 9f\n6:\n\ts_waitcnt vmcnt(" #C1 ")\n\ts_branch 9f\n"        \
                 "7:\n\ts_waitcnt vmcnt(" #C2 ")\n9:"                                                                   \
                 : __VA_ARGS__ : [n] "s"(__builtin_amdgcn_readfirstlane(sel)) : "scc", "memory")

// the transform of the token in `tokbuf` (this wave's 8 KB, landed): Y^T fragments + the token's extrema — the body of fq_kron64_kernel's loop
template <bool RMS>
__device__ __forceinline__ void kl_transform(const unsigned char* tokbuf, const uint4* frag, int lane, float rms_eps, int rt_flags,
                                             f32x16 (&Y)[2][2], float& vmax, float& vmin) {
    typedef f16x8 X8;
    const int h = lane >> 5, c = lane & 31, sw = (c >> 1) & 7;
    u32x4 X[2][4];
    {
        const u32x4* tb = reinterpret_cast<const u32x4*>(tokbuf);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int s = 0; s < 4; ++s) X[mt][s] = tb[(mt * 32 + c) * 8 + ((h * 4 + s) ^ sw)];
    }
    if (RMS) {   // deploy.nn.RMSNorm (normalization.py:16-23), as in fq_kron64_kernel
        float ss[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const X8 v = __builtin_bit_cast(X8, X[mt][s]);
#pragma unroll
                for (int j = 0; j < 8; ++j) ss[j & 3] = __builtin_fmaf((float)v[j], (float)v[j], ss[j & 3]);
            }
        const float tot = fq_wave_sum((ss[0] + ss[1]) + (ss[2] + ss[3]));
        const float rinv = __builtin_amdgcn_rsqf(tot / (float)KD + rms_eps);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                X8 v = __builtin_bit_cast(X8, X[mt][s]);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fq_mul_to<f16>((float)v[j], rinv);
                X[mt][s] = __builtin_bit_cast(u32x4, v);
            }
    }
    const uint4* myfrag = frag + lane;
    f32x16 U[2][2];
    U[0][0] = f32x16{0}; U[0][1] = f32x16{0}; U[1][0] = f32x16{0}; U[1][1] = f32x16{0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const X8 b0 = __builtin_bit_cast(X8, myfrag[(0 * 4 + s) * 64]);
        const X8 b1 = __builtin_bit_cast(X8, myfrag[(1 * 4 + s) * 64]);
        U[0][0] = fq_mfma32<f16>(__builtin_bit_cast(X8, X[0][s]), b0, U[0][0]);
        U[1][0] = fq_mfma32<f16>(__builtin_bit_cast(X8, X[1][s]), b0, U[1][0]);
        U[0][1] = fq_mfma32<f16>(__builtin_bit_cast(X8, X[0][s]), b1, U[0][1]);
        U[1][1] = fq_mfma32<f16>(__builtin_bit_cast(X8, X[1][s]), b1, U[1][1]);
    }
    X8 Uh[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) Uh[nt][ks][j] = (f16)U[ks >> 1][nt][(ks & 1) * 8 + j];
    Y[0][0] = f32x16{0}; Y[0][1] = f32x16{0}; Y[1][0] = f32x16{0}; Y[1][1] = f32x16{0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const X8 b0 = __builtin_bit_cast(X8, myfrag[(8 + ks * 2 + 0) * 64]);
        const X8 b1 = __builtin_bit_cast(X8, myfrag[(8 + ks * 2 + 1) * 64]);
        Y[0][0] = fq_mfma32<f16>(Uh[0][ks], b0, Y[0][0]);
        Y[1][0] = fq_mfma32<f16>(Uh[1][ks], b0, Y[1][0]);
        Y[0][1] = fq_mfma32<f16>(Uh[0][ks], b1, Y[0][1]);
        Y[1][1] = fq_mfma32<f16>(Uh[1][ks], b1, Y[1][1]);
    }
    if (rt_flags & FQ_ROUND_Y_F16) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mo = 0; mo < 2; ++mo)
#pragma unroll
                for (int r = 0; r < 16; ++r) Y[nt][mo][r] = (float)(f16)Y[nt][mo][r];
    }
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
    vmax = fq_wave_max(fq_max3(pmax[0], pmax[1], FqMaxOp()(pmax[2], pmax[3])));
    vmin = fq_wave_min(fq_min3(pmin[0], pmin[1], FqMinOp()(pmin[2], pmin[3])));
}

template <bool RMS>
__global__ __launch_bounds__(KL_THREADS) void fq_kron64_linear_kernel(const f16* __restrict__ x, const uint4* __restrict__ prep, int M,
                                                                      float rms_eps, int rt_flags, K64LinProblems pr) {
    using namespace fqgemm;
    extern __shared__ __attribute__((aligned(16))) unsigned char klsm[];
    // [fragment image 16 KB][M token buffers of 8 KB: the token, then its packed digits in the first 2 KB][tile sums 2 x 32 x 33 int][scales 16 float]
    uint4* frag = reinterpret_cast<uint4*>(klsm);
    unsigned char* toks = klsm + FRAG_BYTES;
    int (*tile)[32][33] = reinterpret_cast<int (*)[32][33]>(toks + (size_t)M * TOK_BYTES);
    float* sc_lds = reinterpret_cast<float*>(toks + (size_t)M * TOK_BYTES + 2 * 32 * 33 * 4);
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- which problem, which feature tiles (workgroup-uniform) ----
    int p = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (q < pr.n && (int)blockIdx.x >= pr.wg0[q]) p = q;
    const uint4* wimg = pr.wimg[0];
    f16* yout = pr.y[0];
    const f16* scol = pr.scol[0];
    const f16* bias = pr.bias[0];
    float sig_max = pr.sig_max[0], sig_min = pr.sig_min[0];
    int N = pr.N[0], wg_lo = pr.wg0[0], wg_hi = pr.wg0[1];
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (p == q) wimg = pr.wimg[q], yout = pr.y[q], scol = pr.scol[q], bias = pr.bias[q], sig_max = pr.sig_max[q], sig_min = pr.sig_min[q],
                    N = pr.N[q], wg_lo = pr.wg0[q], wg_hi = pr.wg0[q + 1];
    const int n_tiles = N >> 5, step = wg_hi - wg_lo;
    int t0 = (int)blockIdx.x - wg_lo;                      // tiles t0, t0 + step, ...
    const f16* bsrc = bias != nullptr ? bias : scol;       // (a dummy source keeps the number of loads per tile fixed: the waits below count them)

    // ---- 1. everything this workgroup needs from memory, requested at once ----
    u32x4 pv[2] = {};
    kl_load16(pv[0], prep + tid);
    kl_load16(pv[1], prep + tid + KL_THREADS);
    const unsigned tok_lds0 = (unsigned)(size_t)(lds_void*)toks;
    for (int m = wave; m < M; m += KL_WAVES)               // (wave-uniform) 8 LDS-DMA instructions per token, older than every weight request
        dma_token(x, m, __builtin_amdgcn_readfirstlane(tok_lds0 + (unsigned)m * TOK_BYTES), lane);
    // weights of two tiles in flight per wave: ring slot = tile parity; + the tile's column scale and bias for this thread's output element
    u32x4 W0[KL_BPW] = {}, W1[KL_BPW] = {};
    unsigned s0 = 0, b0 = 0, s1 = 0, b1 = 0;
    const int nl = tid & 31, mrow = tid >> 5;              // the output element of this thread in a tile: token mrow (clamped below), feature nl
#define KL_REQ(Wr, sr, br, t)                                                                                     \
    {                                                                                                             \
        const uint4* wp_ = wimg + ((size_t)(t) * KL_KB + wave) * 64 + lane;                                       \
        _Pragma("unroll") for (int j = 0; j < KL_BPW; ++j) kl_load16(Wr[j], wp_ + (size_t)j * KL_WAVES * 64);     \
        kl_load2(sr, scol + (t) * 32 + nl);                                                                       \
        kl_load2(br, bsrc + (t) * 32 + nl);                                                                       \
    }
    const bool have0 = t0 < n_tiles, have1 = t0 + step < n_tiles;
    if (have0) KL_REQ(W0, s0, b0, t0)
    if (have1) KL_REQ(W1, s1, b1, t0 + step)
    // the image and the tokens are older than the weights: wait for them only (10 loads per requested tile stay in flight)
    {
        const int sel = have1 ? 2 : have0 ? 1 : 0;
        KL_WAIT4(sel, 0, 10, 20, 20, "+v"(pv[0]), "+v"(pv[1]));
    }
    frag[tid] = __builtin_bit_cast(uint4, pv[0]);
    frag[tid + KL_THREADS] = __builtin_bit_cast(uint4, pv[1]);
    for (int i = tid; i < 2 * 32 * 33; i += KL_THREADS) (&tile[0][0][0])[i] = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // the image is in LDS (the tokens: every wave waited for its own DMAs)

    // ---- 2. transform + quantise: wave w takes tokens w, w + 8 ----
#ifndef FQ_KL_ABL
#define FQ_KL_ABL 0   // measurement builds (wrong results): 1 = no transform (the digits are whatever the token's first 2 KB hold)
#endif
    for (int m = wave; m < M; m += KL_WAVES) {
        unsigned char* tokbuf = toks + (size_t)m * TOK_BYTES;
        if (FQ_KL_ABL & 1) {
            if (lane == 0) sc_lds[m] = 1.0f;
            continue;
        }
        f32x16 Y[2][2];
        float vmax, vmin;
        kl_transform<RMS>(tokbuf, frag, lane, rms_eps, rt_flags, Y, vmax, vmin);
        const float scale = fq_token_scale<FQ_OUT_PACKED, f16>(vmax, vmin, sig_max, sig_min, rt_flags);
        const float inv[2] = {fq_fast_inv(scale), fq_fast_inv(scale)};
        uint32_t pw[2][4];
        unsigned near;
        if (!fq_magic_ok(vmax, vmin, inv[0])) near = 0xffu;
        else if (fq_needs_clamp(vmax, vmin, inv[0])) near = quant_pack_token<true>(Y, inv, pw);
        else near = quant_pack_token<false>(Y, inv, pw);
#define FQ_YV(mo, w, e) Y[(w) >> 1][mo][((w) & 1) * 8 + (e)]
        if (near) {
#pragma unroll
            for (int mo = 0; mo < 2; ++mo)
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    if (near & (1u << (4 * mo + w)))
                        pw[mo][w] = fq_pack8(fq_qexact(FQ_YV(mo, w, 0), scale), fq_qexact(FQ_YV(mo, w, 1), scale), fq_qexact(FQ_YV(mo, w, 2), scale),
                                             fq_qexact(FQ_YV(mo, w, 3), scale), fq_qexact(FQ_YV(mo, w, 4), scale), fq_qexact(FQ_YV(mo, w, 5), scale),
                                             fq_qexact(FQ_YV(mo, w, 6), scale), fq_qexact(FQ_YV(mo, w, 7), scale));
        }
#undef FQ_YV
        // the packed row of the token, natural order (what fq_kron64_kernel stores): output row 32 mo + c, bytes 16 h .. of its 32
#pragma unroll
        for (int mo = 0; mo < 2; ++mo)
            *reinterpret_cast<u32x4*>(tokbuf + (mo * 32 + c) * (KN / 2) + h * 16) = u32x4{pw[mo][0], pw[mo][1], pw[mo][2], pw[mo][3]};
        if (lane == 0) sc_lds[m] = (float)(f16)scale;      // (the fp16 scale the packed launch stores and the GEMM reads)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- 3. the feature tiles ----
    // B fragments of this wave's 8 k-blocks: token min(c, M - 1), packed bytes 32 kb + 16 h .. (loop-invariant: registers)
    u32x4 BX[KL_BPW];
    {
        const unsigned char* xr = toks + (size_t)(c < M ? c : M - 1) * TOK_BYTES + 16 * h;
#pragma unroll
        for (int j = 0; j < KL_BPW; ++j) BX[j] = *reinterpret_cast<const u32x4*>(xr + (wave + j * KL_WAVES) * 32);
    }
    const int mo_ = mrow < M ? mrow : M - 1;               // rows >= M hold copies of token M - 1: the same value to the same address
    const f16 srow = (f16)sc_lds[mo_];
    int par = 0;
    auto consume = [&](u32x4 (&Wr)[KL_BPW]) {
        kl_i32x16 acc = kl_i32x16{0};
#pragma unroll
        for (int j = 0; j < KL_BPW; ++j) {
            const i32x4_t a0 = unpack16(make_uint2(Wr[j][0], Wr[j][1])), a1 = unpack16(make_uint2(Wr[j][2], Wr[j][3]));
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, unpack16(make_uint2(BX[j][0], BX[j][1])), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, unpack16(make_uint2(BX[j][2], BX[j][3])), acc, 0, 0, 0);
        }
        return acc;
    };
    auto finish = [&](const kl_i32x16& acc, unsigned sr, unsigned br, int t) {
        // lane (h, c): token c, features 16 h + r; the eight waves' partial sums meet in LDS
#pragma unroll
        for (int r = 0; r < 16; ++r) atomicAdd(&tile[par][c][16 * h + r], acc[r]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int v = tile[par][mrow][nl] >> 8;            // products carry 256 (unpack16)
        f16 yv = dequant1(v, srow, __builtin_bit_cast(f16, (unsigned short)sr));
        if (bias != nullptr) yv = yv + __builtin_bit_cast(f16, (unsigned short)br);
        const unsigned yb = (unsigned)__builtin_bit_cast(unsigned short, yv);
        f16* dst = yout + (size_t)mo_ * N + t * 32 + nl;
        asm volatile("global_store_short %0, %1, off" : : "v"(dst), "v"(yb) : "memory");
        tile[par][mrow][nl] = 0;                           // (this buffer is next used two tiles on: a barrier lies in between)
        par ^= 1;
    };
    for (int t = t0; t < n_tiles; t += 2 * step) {
        // Issue order of an iteration: [wait slot 0] MFMAs, request slot 0 again (10 loads), store of tile t (1), [wait slot 1] MFMAs, request
        // slot 1 again (10), store of tile t + step (1). Operations YOUNGER than the loads a wait is for:
        //   slot 0, first iteration: slot 1's request (10 | 0);          later: store, slot 1's request, store (12 | 2)
        //   slot 1, first iteration: slot 0's new request, store (11 | 1); later: store, request, store (12 | 2)
        const bool nx = t + step < n_tiles, first = t == t0;
#define KL_W0 "+v"(W0[0]), "+v"(W0[1]), "+v"(W0[2]), "+v"(W0[3]), "+v"(W0[4]), "+v"(W0[5]), "+v"(W0[6]), "+v"(W0[7]), "+v"(s0), "+v"(b0)
        {
            const int sel = (first ? 0 : 2) + (nx ? 1 : 0);    // first: 0 | 10, later: 2 | 12
            KL_WAIT4(sel, 0, 10, 2, 12, KL_W0);
        }
#undef KL_W0
        const kl_i32x16 a0 = consume(W0);
        unsigned sk0, bk0;   // (real moves: a C++ copy lets the allocator keep the OLD value in place and load the new one elsewhere)
        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(sk0), "=&v"(bk0) : "v"(s0), "v"(b0));
        if (t + 2 * step < n_tiles) KL_REQ(W0, s0, b0, t + 2 * step)
        finish(a0, sk0, bk0, t);
        if (!nx) break;
        // tile t + step (slot 1): younger = slot 0's 10 loads (when requested) + the store of tile t
        const bool nx2 = t + 2 * step < n_tiles;
#define KL_W1 "+v"(W1[0]), "+v"(W1[1]), "+v"(W1[2]), "+v"(W1[3]), "+v"(W1[4]), "+v"(W1[5]), "+v"(W1[6]), "+v"(W1[7]), "+v"(s1), "+v"(b1)
        {
            const int sel = (first ? 0 : 2) + (nx2 ? 1 : 0);   // first: 1 | 11, later: 2 | 12
            KL_WAIT4(sel, 1, 11, 2, 12, KL_W1);
        }
#undef KL_W1
        const kl_i32x16 a1 = consume(W1);
        unsigned sk1, bk1;
        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(sk1), "=&v"(bk1) : "v"(s1), "v"(b1));
        if (t + 3 * step < n_tiles) KL_REQ(W1, s1, b1, t + 3 * step)
        finish(a1, sk1, bk1, t + step);
    }
#undef KL_REQ
}

}  // namespace

// The fused decode launch (fq_kron64_linear_multi_f16): -1000 when the shape is not covered. 
int fq_launch_kron64_linear(const f16* x, const void* prep, int64_t M, float rms_eps, bool rms, int rt_flags, int n, const void* const* wimg,
                            const f16* const* scol, const f16* const* bias, const float* sig_max, const float* sig_min, const int* N,
                            f16* const* y, int n_cu, hipStream_t stream) {
    if (n < 1 || n > 4 || M < 1 || M > KL_MAXM || prep == nullptr) return -1000;
    K64LinProblems pr = {};
    pr.n = n;
    int tiles[4] = {0, 0, 0, 0}, total = 0;
    for (int p = 0; p < n; ++p) {
        if (N[p] < 32 || (N[p] & 31)) return -1000;
        pr.wimg[p] = reinterpret_cast<const uint4*>(wimg[p]);
        pr.y[p] = y[p];
        pr.scol[p] = scol[p];
        pr.bias[p] = bias ? bias[p] : nullptr;
        pr.sig_max[p] = sig_max[p];
        pr.sig_min[p] = sig_min[p];
        pr.N[p] = N[p];
        tiles[p] = N[p] >> 5;
        total += tiles[p];
    }
    // workgroups: one per CU (or one per tile when there are fewer tiles), dealt to the problems in proportion to their tiles
#ifndef FQ_KL_WG_PER_CU
#define FQ_KL_WG_PER_CU 1   // (measurement knob) persistent workgroups per CU
#endif
    int grid = n_cu * FQ_KL_WG_PER_CU;
    if (grid > total) grid = total;
    // a problem's share of the grid in proportion to its tiles, then trimmed so that its workgroups walk the SAME number of tiles where the
    // counts allow (448 tiles on a share of 128 workgroups are 3.5 each — half of them stream a fourth tile alone, with too few requests in
    // flight for the memory system: 13.0 us; 112 workgroups x 4 tiles: measured below)
    int wgs[4] = {0, 0, 0, 0};
    for (int p = 0; p < n; ++p) {
        int share = (int)((int64_t)grid * tiles[p] / total);
        if (share < 1) share = 1;
        if (share > tiles[p]) share = tiles[p];
        const int per = (tiles[p] + share - 1) / share;
        wgs[p] = (tiles[p] + per - 1) / per;
    }
    pr.wg0[0] = 0;
    for (int p = 0; p < n; ++p) pr.wg0[p + 1] = pr.wg0[p] + wgs[p];
    for (int p = n + 1; p < 5; ++p) pr.wg0[p] = pr.wg0[n];
    const int lds = FRAG_BYTES + (int)M * TOK_BYTES + 2 * 32 * 33 * 4 + KL_MAXM * 4;
    if (rms) {
        FQ_RAISE_LDS_CAP((fq_kron64_linear_kernel<true>), 160 * 1024);
        hipLaunchKernelGGL((fq_kron64_linear_kernel<true>), dim3((unsigned)pr.wg0[n]), dim3(KL_THREADS), lds, stream, x,
                           reinterpret_cast<const uint4*>(prep), (int)M, rms_eps, rt_flags, pr);
    } else {
        FQ_RAISE_LDS_CAP((fq_kron64_linear_kernel<false>), 160 * 1024);
        hipLaunchKernelGGL((fq_kron64_linear_kernel<false>), dim3((unsigned)pr.wg0[n]), dim3(KL_THREADS), lds, stream, x,
                           reinterpret_cast<const uint4*>(prep), (int)M, rms_eps, rt_flags, pr);
    }
    return (int)hipGetLastError();
}

// Host-side launcher used by the C ABI (fq_capi.hip). Returns hipError_t as int.
// The 16 KB fragment image of (left, right) for `prep`: slot (f, lane') as the kernel's prologue gathers it.
__global__ void fq_kron64_prepare_kernel(const unsigned short* __restrict__ left, const unsigned short* __restrict__ right,
                                         uint4* __restrict__ image) {
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= 16 * 64) return;
    const int f = item >> 6, ln = item & 63, fh = ln >> 5, fc = ln & 31;
    unsigned e[8];
    if (f < 8) {
        const int nt = f >> 2, sk = f & 3;
        const unsigned short* src = right + (fh * 32 + sk * 8) * KN + nperm(nt, fc);
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = src[j * KN];
    } else {
        const int ks = (f - 8) >> 1, mo = (f - 8) & 1;
        const unsigned short* src = left + ((ks >> 1) * 32 + 16 * (ks & 1) + 4 * fh) * KM + mo * 32 + fc;
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = src[(8 * (j >> 2) + (j & 3)) * KM];
    }
    image[item] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
}

template <int FLAGS, typename T>
static int launch_kron64(const T* x, const T* left, const T* right, const T* diag, int64_t rows,
                         const FqQuantOut& out, int n_cu, hipStream_t stream, const void* prep = nullptr) {
    constexpr int THREADS = kron64_threads<FLAGS>();
    // few rows (decode): four tokens per workgroup, so that only the first SIMD-slot group of waves has work and no wave
    // sits out the start-up stagger (up to 3 x 2560 cycles for the last group: most of an 11 us launch at 16 rows)
    int64_t blocks = (rows + 3) / 4;
    if (blocks > n_cu) blocks = n_cu;  // one persistent workgroup per CU (LDS: 16 KB + 8 KB per wave)
    if (blocks < 1) blocks = 1;
    const int64_t tpb = (rows + blocks - 1) / blocks;
    hipLaunchKernelGGL((fq_kron64_kernel<FLAGS, false, T>), dim3((unsigned)blocks), dim3(THREADS), 0, stream, x, left,
                       right, diag, rows, tpb, reinterpret_cast<const uint4*>(prep), out, (unsigned long long*)nullptr,
                       (const FqKron64Job*)nullptr, 0);
    return (int)hipGetLastError();
}

// The multi-job launch (fq_kron_quant_multi_{f16,bf16}): packed output, one clip pair for all jobs, prepared images. `jobs` is
// DEVICE memory, [n_jobs], already filled (x, prep, q, scale, rows, tpb = ceil(rows / bpj)).
template <typename T>
static int launch_kron64_multi(const FqKron64Job* jobs, int n_jobs, int bpj, const FqQuantOut& out, hipStream_t stream) {
    constexpr int FL = FQ_OUT_PACKED | FQ_K64_MULTI;
    constexpr int THREADS = kron64_threads<FL>();
    hipLaunchKernelGGL((fq_kron64_kernel<FL, false, T>), dim3((unsigned)(n_jobs * bpj)), dim3(THREADS), 0, stream, (const T*)nullptr,
                       (const T*)nullptr, (const T*)nullptr, (const T*)nullptr, (int64_t)0, (int64_t)0, (const uint4*)jobs /* != nullptr: prepared */,
                       out, (unsigned long long*)nullptr, jobs, bpj);
    return (int)hipGetLastError();
}
int fq_launch_kron64_multi(int bf16_dtype, const void* jobs, int n_jobs, int bpj, const FqQuantOut& out, hipStream_t stream) {
    return bf16_dtype ? launch_kron64_multi<bf16>((const FqKron64Job*)jobs, n_jobs, bpj, out, stream)
                      : launch_kron64_multi<f16>((const FqKron64Job*)jobs, n_jobs, bpj, out, stream);
}

template <typename T>
static int launch_kron64_any(int flags, const T* x, const T* left, const T* right, const T* diag,
                             int64_t rows, const FqQuantOut& out, int n_cu, hipStream_t stream, const void* prep) {
    // Compile-time specialisations: output set x fp16-quant arithmetic. Everything else is run-time.
#define FQ_CASE(F)                                                                     \
    case (F):                                                                          \
        return launch_kron64<(F), T>(x, left, right, diag, rows, out, n_cu, stream, prep);      \
    case (F) | FQ_QUANT_F16:                                                           \
        return launch_kron64<(F) | FQ_QUANT_F16, T>(x, left, right, diag, rows, out, n_cu, stream, prep);
    if (out.rt_flags & FQ_GROUP128) {  // per-128-element scales: packed output, fp32 quantiser arithmetic only
        if ((flags & FQ_CT_MASK) != FQ_OUT_PACKED) return -1000;
        if (out.group_offsets != nullptr)
            return launch_kron64<FQ_OUT_PACKED | FQ_K64_G128 | FQ_K64_GROUPED, T>(x, left, right, diag, rows, out, n_cu, stream, prep);
        return launch_kron64<FQ_OUT_PACKED | FQ_K64_G128, T>(x, left, right, diag, rows, out, n_cu, stream, prep);
    }
    if (out.group_offsets != nullptr) {  // grouped launch: packed or fake-quant output (+ the transform)
        switch (flags & FQ_CT_MASK) {
            case FQ_OUT_PACKED:
                return launch_kron64<FQ_OUT_PACKED | FQ_K64_GROUPED, T>(x, left, right, diag, rows, out, n_cu, stream, prep);
            case FQ_OUT_FAKEQUANT:
                return launch_kron64<FQ_OUT_FAKEQUANT | FQ_K64_GROUPED, T>(x, left, right, diag, rows, out, n_cu, stream, prep);
            case FQ_OUT_PACKED | FQ_OUT_TRANSFORM:
                return launch_kron64<FQ_OUT_PACKED | FQ_OUT_TRANSFORM | FQ_K64_GROUPED, T>(x, left, right, diag, rows, out, n_cu, stream, prep);
            case FQ_OUT_FAKEQUANT | FQ_OUT_TRANSFORM:
                return launch_kron64<FQ_OUT_FAKEQUANT | FQ_OUT_TRANSFORM | FQ_K64_GROUPED, T>(x, left, right, diag, rows, out, n_cu, stream, prep);
            // the all-low-precision quantiser (clip parameters of the activation's dtype: the DeepSeek flow under
            // torch.set_default_dtype(bfloat16), main_dpskv3.py:395)
            case FQ_OUT_FAKEQUANT | FQ_QUANT_F16:
                return launch_kron64<FQ_OUT_FAKEQUANT | FQ_QUANT_F16 | FQ_K64_GROUPED, T>(x, left, right, diag, rows, out, n_cu, stream, prep);
            case FQ_OUT_FAKEQUANT | FQ_OUT_TRANSFORM | FQ_QUANT_F16:
                return launch_kron64<FQ_OUT_FAKEQUANT | FQ_OUT_TRANSFORM | FQ_QUANT_F16 | FQ_K64_GROUPED, T>(x, left, right, diag, rows, out, n_cu, stream, prep);
            default:
                return -1000;
        }
    }
    switch (flags & FQ_CT_MASK) {
        FQ_CASE(FQ_OUT_PACKED)
        FQ_CASE(FQ_OUT_FAKEQUANT)
        FQ_CASE(FQ_OUT_PACKED | FQ_OUT_FAKEQUANT)
        FQ_CASE(FQ_OUT_TRANSFORM | FQ_OUT_PACKED)
        FQ_CASE(FQ_OUT_TRANSFORM | FQ_OUT_FAKEQUANT)
        FQ_CASE(FQ_OUT_TRANSFORM | FQ_OUT_PACKED | FQ_OUT_FAKEQUANT)
        // (the RMSNorm-fused launches serve deploy.nn.RMSNorm, an fp16-only module in the reference)
        case FQ_OUT_PACKED | FQ_IN_RMSNORM:
            if constexpr (FqVec<T>::is_f16) return launch_kron64<FQ_OUT_PACKED | FQ_IN_RMSNORM, T>(x, left, right, diag, rows, out, n_cu, stream, prep);
            return -1000;
        case FQ_OUT_TRANSFORM | FQ_IN_RMSNORM:
            if constexpr (FqVec<T>::is_f16) return launch_kron64<FQ_OUT_TRANSFORM | FQ_IN_RMSNORM, T>(x, left, right, diag, rows, out, n_cu, stream, prep);
            return -1000;
        case FQ_OUT_TRANSFORM | FQ_OUT_PACKED | FQ_IN_RMSNORM:
            if constexpr (FqVec<T>::is_f16) return launch_kron64<FQ_OUT_TRANSFORM | FQ_OUT_PACKED | FQ_IN_RMSNORM, T>(x, left, right, diag, rows, out, n_cu, stream, prep);
            return -1000;
        case FQ_OUT_TRANSFORM:
        case FQ_OUT_TRANSFORM | FQ_QUANT_F16:
            return launch_kron64<FQ_OUT_TRANSFORM, T>(x, left, right, diag, rows, out, n_cu, stream, prep);
        default:
            return -1000;
    }
#undef FQ_CASE
}

int fq_launch_kron64(int flags, const f16* x, const f16* left, const f16* right, const f16* diag,
                     int64_t rows, const FqQuantOut& out, int n_cu, hipStream_t stream, const void* prep) {
    if (flags & FQ_DT_BF16)
        return launch_kron64_any<bf16>(flags & ~FQ_DT_BF16, (const bf16*)x, (const bf16*)left, (const bf16*)right, (const bf16*)diag,
                                       rows, out, n_cu, stream, prep);
    return launch_kron64_any<f16>(flags, x, left, right, diag, rows, out, n_cu, stream, prep);
}

int fq_launch_kron64_prepare(const void* left, const void* right, void* image, hipStream_t stream) {
    hipLaunchKernelGGL(fq_kron64_prepare_kernel, dim3(4), dim3(256), 0, stream, (const unsigned short*)left,
                       (const unsigned short*)right, reinterpret_cast<uint4*>(image));
    return (int)hipGetLastError();
}

// Debug: the packed kernel with per-phase s_memtime accounting (see FQ_TICK). trace: [n_waves, 4] u64,
// n_waves = 16 * min(ceil(rows/16), n_cu).
int fq_launch_kron64_trace(const f16* x, const f16* left, const f16* right, int64_t rows, const FqQuantOut& out,
                           unsigned long long* trace, int n_cu, hipStream_t stream) {
    int64_t blocks = (rows + 15) / 16;
    if (blocks > n_cu) blocks = n_cu;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((fq_kron64_kernel<FQ_OUT_PACKED, true, f16>), dim3((unsigned)blocks), dim3(1024), 0, stream, x, left,
                       right, (const f16*)nullptr, rows, (rows + blocks - 1) / blocks, (const uint4*)nullptr, out, trace, (const FqKron64Job*)nullptr, 0);
    return (int)hipGetLastError();
}
