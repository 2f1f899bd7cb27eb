// fq_kron_wave.hip — fused Kronecker transform + per-token INT4 quantisation, ONE WAVE PER TOKEN, for factor pairs
// whose token fits a wave's registers: M <= 64, N in {64, 80, 112, 128} (<= 4 column tiles):
// d = 8192 (64x128, Llama-2-70B hidden), 7168 (64x112, DeepSeek-V3 hidden), 5120 (64x80, Qwen2.5-14B/32B hidden),
// 3584 (56x64), 2048 (32x64).
// Packed INT4 + fp16 scale output (the deploy.nn.OnlineTrans contract: deploy/kernels/kron_matmul.py:192-266,
// functional/online_trans.py:113-122); everything else goes to fq_kron_generic.hip.
//
// Why: the workgroup-per-token kernel needs three workgroup barriers per token and spends ~60 % of its wave
// cycles waiting at them (rocprofv3: SQ_WAIT_ANY / SQ_WAVE_CYCLES). Here a wave owns its token end to end, as in
// fq_kron64.hip, and nothing is synchronised after the prologue:
//   * token: HBM -> LDS by DMA (global_load_lds_dwordx4, full 1 KB lines, source-side XOR swizzle so the A-fragment
//     ds_read_b128 are conflict-free), into a wave-private buffer; the next token's DMA is issued as soon as
//     GEMM 1 has read the current one, and is waited for with a COUNTED vmcnt at the top of the loop;
//   * R and L fragments: fragment-ordered image in LDS, shared by the workgroup, copied once from the workspace
//     that fq_kron_prepare_kernel fills (same layout as the generic kernel's);
//   * GEMM 1 with the K-step outermost: each A fragment is read once and used for all column tiles
//     (U: NT x MT accumulators), then fp16 rounding (flat_utils.py:15) turns the C fragments into GEMM 2's A
//     fragments; GEMM 2 with (ks, mo) outermost: each L fragment is read once and used for all NT tiles;
//   * statistics, scale, magic-number quantiser and nibble packing on the accumulators in registers; a lane ends
//     up with NT*16 consecutive n' of an output row = NT*8 contiguous bytes -> 16-byte stores.
// LDS reads per 64x128 token: 16 (A) + 32 (R) + 8 (L) ds_read_b128 against 96 MFMAs.
// bf16 activations (round 3, second session): the kernel is a template on the element type like the other path-A kernels (bf16 MFMA,
// U / the optional Y rounding / the norm / the scale in bf16); fq_launch_kron_wave_bf16 serves the packed-only bf16 launches
// that used to run the workgroup-per-token kernel (64 x 128: 117 us against 77).
#include "fq_common.hpp"
#include "fq_dma.hpp"

namespace {

#ifndef WAVE_ABL
#define WAVE_ABL 0   // measurement builds (tools/variants.sh): 1 no quantiser arithmetic, 2 no GEMM 1 MFMAs, 4 no GEMM 2 MFMAs, 16 no DMA after the first token
#endif

// (round 5, measured and removed: 64 x 128 with EIGHT resident tokens — the L fragments in registers and static token slots free the
//  8 KB + 16 bytes between 7 and 8 tokens, two waves on every SIMD instead of 2 / 2 / 2 / 1 — 76.3 against 75.5 us, packed3 112.9 against
//  110.9: the launch runs at the 1400 W package cap, 1.75 GHz, and is bound by joules per token, not by waves — profiles/r05_energy_census.txt)
template <int MT, int NT, int KS1, int W>
struct WaveGeom {
    static constexpr int N = KS1 * 16;
    static constexpr int CPR = KS1 * 2;                     // 16-byte chunks per token row
    static constexpr int TOKBUF = MT * 32 * CPR * 16;       // bytes, rows padded to the tile
    static constexpr int RFR = NT * KS1 * 64;               // uint4
    static constexpr int LFR = 2 * MT * MT * 64;            // uint4
    static constexpr int LDS = (RFR + LFR) * 16 + W * TOKBUF + 16;
    static constexpr int V1 = N - NT * 16;                  // valid n' in the h = 1 half of a lane's run
    static constexpr int NPIECE = (NT * 2 + 3) / 4;         // 16-byte pieces of a lane's packed run (the last may be 8 bytes)
    // VMEM store instructions per output row group: piece p is a 16-byte store for the halves with >= 32 p + 32 valid n',
    // an 8-byte store for those with exactly 32 p + 16 (h = 0 runs hold NT * 16 valid n', h = 1 runs V1)
    static constexpr int row_stores() {
        int n = 0;
        for (int p = 0; p < NPIECE; ++p) {
            const int a = NT * 16 - 32 * p, b = V1 - 32 * p;
            n += (a >= 32 || b >= 32) ? 1 : 0;
            n += ((a >= 16 && a < 32) || (b >= 16 && b < 32)) ? 1 : 0;
        }
        return n;
    }
    static constexpr int STORES = MT * row_stores() + 1;    // VMEM stores per token and clip (+ the scale)
};

// The single-width asm quantiser (fq_quant8_two, fq_common.hpp) over one output row group (mo) of a lane: NT * 2 packed
// dwords; returns the mask of dwords in which some lane of the wave saw an ambiguous digit (redone with the true division).
template <bool CLAMP, int NT, int MT>
__device__ __forceinline__ unsigned quant_row(const f32x16 (&Y)[NT][MT], int mo, float inv, uint32_t (&pw)[NT * 2]) {
    const float ilo = fq_inv_lo(inv), ihi = fq_inv_hi(inv);
    unsigned near = 0;
#pragma unroll
    for (int k = 0; k < NT * 2; ++k) {
        const f32x16& t = Y[k >> 1][mo];
        const int b = (k & 1) * 8;
        unsigned long long differ;
        pw[k] = fq_quant8<CLAMP>(t[b + 0], t[b + 1], t[b + 2], t[b + 3], t[b + 4], t[b + 5], t[b + 6], t[b + 7], inv, ilo, ihi,
                                     differ);
        near |= differ ? (1u << k) : 0u;
    }
    return near;
}

// RMS (round 3): deploy.nn.RMSNorm in front of the transform in the same launch (deploy/nn/normalization.py:16-23,
// modeling_llama.py:351-357 apply it in front of EVERY pair; round 2 fused it for 64 x 64 only). The wave owns the whole token
// in its LDS buffer: one linear pass over it for the sum of squares (M * N / 512 conflict-free ds_read_b128 per lane, fp32 fmas
// in four chains, wave all-reduce, v_rsq_f32), then every A fragment is scaled — fp32 product rounded to fp32 and to fp16, the
// module's `(x.float() * rsqrt(...)).to(fp16)` — on its way into GEMM 1. No extra HBM traffic; the separate normalisation launch
// moves 4 bytes per element.
// OS (round 4): the output set — FQ_OUT_PACKED (the deploy contract, everything above), or FQ_OUT_FAKEQUANT (FlatQuantizedLinear.
// _eval_forward: scale * q in the activation dtype, flat_linear.py:75-80 — the reference's fake-quant eval flow, the only one it has for
// DeepSeek-V3) or FQ_OUT_TRANSFORM (kronecker_matmul alone). A lane holds NT * 16 consecutive n' of an output row = NT * 32 contiguous
// bytes (a whole 128-byte line at N = 128): stored as 16-byte pieces straight from the fragments; fq_fake8 with one exactness vote per
// token, as fq_kron64.hip. fp32 quantiser arithmetic, per-token scales; everything else stays with the workgroup-per-token kernel
// (which ran these outputs until now: 64 x 128 fake-quant 121 us, 32 x 64 45 us = 0.37 of the roofline).
template <int MT, int NT, int KS1, int W, bool RMS = false, typename T = f16, int OS = FQ_OUT_PACKED>
__global__ __launch_bounds__(W * 64) void fq_kron_wave_kernel(const T* __restrict__ x, const uint4* __restrict__ ws,
                                                            int64_t rows, int64_t tpb, int M, FqQuantOut out) {
    typedef WaveGeom<MT, NT, KS1, W> G;
    typedef typename FqVec<T>::x8 X8;
    static_assert(OS == FQ_OUT_PACKED || OS == FQ_OUT_FAKEQUANT || OS == FQ_OUT_TRANSFORM, "one output set per instantiation");
    constexpr int N = G::N, CPR = G::CPR;
    static_assert(NT <= 4 && MT <= 2, "a token must fit one wave's accumulators");
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS];
    uint4* rfr = reinterpret_cast<uint4*>(smem);
    uint4* lfr = rfr + G::RFR;
    unsigned char* tok0 = smem + (G::RFR + G::LFR) * 16;
    unsigned* next_slot = reinterpret_cast<unsigned*>(tok0 + W * G::TOKBUF);

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* tokbuf = tok0 + wave * G::TOKBUF;
    const unsigned tok_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)tokbuf);
    const int64_t tok_bytes = (int64_t)M * N * 2;
    const int n_dma = (M * CPR) >> 6;  // the launcher guarantees M * CPR % 64 == 0

    const int64_t blk_base = (int64_t)blockIdx.x * tpb;
    const int blk_cnt = (int)(rows - blk_base < tpb ? (rows - blk_base < 0 ? 0 : rows - blk_base) : tpb);
    if (tid == 0) *next_slot = W;
    int slot = wave;

    // ---- once per workgroup: fragment image (coalesced copy), zero rows below the token, first DMA ----
    for (int i = tid; i < G::RFR + G::LFR; i += W * 64) rfr[i] = ws[i];
    for (int i = M * CPR + lane; i < MT * 32 * CPR; i += 64) reinterpret_cast<uint4*>(tokbuf)[i] = make_uint4(0, 0, 0, 0);
    unsigned voff[4];
    dma_offsets<CPR>(lane, voff);
    if (slot < blk_cnt) dma_token<CPR>(reinterpret_cast<const f16*>(x), blk_base + slot, tok_bytes, n_dma, tok_lds, voff);   // (16-bit elements either way)
    __syncthreads();  // (no VMEM the compiler knows of is in flight: lgkmcnt(0) + s_barrier)

    // A-fragment byte offsets of this lane: row (32 mt + c), chunk (2 s + h) ^ swz(row)
    const int sw = swz<CPR>(c);  // rows 32 mt + c share their low bits with c
    bool first = true;
    int next_pulled = 0;
    FqGroupCursor gcur;  // grouped launches: the clip pair follows the token's group

    while (slot < blk_cnt) {
        const int64_t tok = blk_base + slot;
        int foff = lane;
        asm volatile("" : "+v"(foff));  // keep the fragment reads inside the loop (LICM would hoist and spill them)
        const uint4* myr = rfr + foff;
        const uint4* myl = lfr + foff;
        // this token's DMA: issued in the prologue or an iteration ago; younger VMEM ops = that iteration's stores
        // (round 5, measured and removed: counted waits for the multi-clip launches too — vmcnt(n_clips x STORES) — 64 x 128 with three
        //  clip sets 112.5 against 113.3 us, C4 step 54.57 against 54.75 ms: the store acknowledgements are already hidden; profiles/r05_nclip_wait.txt)
        if (OS == FQ_OUT_PACKED && !first && out.n_clips == 1) {
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(G::STORES) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        first = false;

        float rinv = 1.0f;
        if (RMS) {
            const uint4* tb = reinterpret_cast<const uint4*>(tokbuf);
            float ss[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            constexpr int NCH = MT * 32 * CPR / 64;    // 16-byte chunks per lane (rows beyond M are zero: they add nothing)
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const X8 v = __builtin_bit_cast(X8, tb[i * 64 + lane]);   // (the swizzle permutes chunks inside a row: the sum does not care)
#pragma unroll
                for (int j = 0; j < 8; ++j) ss[j & 3] = __builtin_fmaf((float)v[j], (float)v[j], ss[j & 3]);
            }
            const float tot = fq_wave_sum((ss[0] + ss[1]) + (ss[2] + ss[3]));
            rinv = __builtin_amdgcn_rsqf(tot / (float)(M * N) + out.rms_eps);
        }
        auto norm8 = [&](X8 v) -> X8 {
            if (RMS) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fq_mul_to<T>((float)v[j], rinv);
            }
            return v;
        };

        // ---- GEMM 1, K-step outermost: U[nt][mt] += X(mt, s) . R(s, nt) ----
        f32x16 U[NT][MT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) U[nt][mt] = f32x16{0};
        {
            const uint4* tb = reinterpret_cast<const uint4*>(tokbuf);
            X8 A[2][MT], B[2][NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) A[0][mt] = norm8(__builtin_bit_cast(X8, tb[(mt * 32 + c) * CPR + (h ^ sw)]));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) B[0][nt] = __builtin_bit_cast(X8, myr[(nt * KS1) * 64]);
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                if (s + 1 < KS1) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        A[(s + 1) & 1][mt] =
                            norm8(__builtin_bit_cast(X8, tb[(mt * 32 + c) * CPR + (((s + 1) * 2 + h) ^ sw)]));
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        B[(s + 1) & 1][nt] = __builtin_bit_cast(X8, myr[(nt * KS1 + s + 1) * 64]);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        if (!(WAVE_ABL & 2)) U[nt][mt] = fq_mfma32<T>(A[s & 1][mt], B[s & 1][nt], U[nt][mt]);
                        else if (WAVE_ABL & 32) asm volatile("" : : "v"(A[s & 1][mt]), "v"(B[s & 1][nt]));   // (round 6: the operand reads stay)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the token buffer has been read: pull the next token and start its DMA
        // (16-bit outputs: the buffer serves as their output stage first — the request goes out behind the stores, below)
        auto pull_next = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            int nxt = 0;
            if (lane == 0) nxt = (int)atomicAdd(next_slot, 1u);
            nxt = __builtin_amdgcn_readfirstlane(nxt);
            if (nxt < blk_cnt) dma_token<CPR>(reinterpret_cast<const f16*>(x), blk_base + nxt, tok_bytes, n_dma, tok_lds, voff);
            next_pulled = nxt;
        };
        if constexpr (OS == FQ_OUT_PACKED) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            int nxt = 0;
            if (lane == 0) nxt = (int)atomicAdd(next_slot, 1u);
            nxt = __builtin_amdgcn_readfirstlane(nxt);
            if (nxt < blk_cnt && !(WAVE_ABL & 16)) dma_token<CPR>(reinterpret_cast<const f16*>(x), blk_base + nxt, tok_bytes, n_dma, tok_lds, voff);
            next_pulled = nxt;
        }

        // ---- fp16 rounding of U (flat_utils.py:15): C fragments -> A fragments of GEMM 2, no data movement ----
        X8 Uh[NT][2 * MT];  // [nt][ks]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int ks = 0; ks < 2 * MT; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) Uh[nt][ks][j] = (T)U[nt][ks >> 1][(ks & 1) * 8 + j];

        // ---- GEMM 2, (ks, mo) outermost: Y[nt][mo] += U(:, nt)^T(ks) . L(ks, mo) ----
        f32x16 Y[NT][MT];  // Y^T of tile (nt, mo): rows n' = h*NT*16 + nt*16 + r, col m' = 32 mo + c
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) Y[nt][mo] = f32x16{0};
        {
            X8 B[2];
            B[0] = __builtin_bit_cast(X8, myl[0]);
#pragma unroll
            for (int i = 0; i < 2 * MT * MT; ++i) {  // i = ks * MT + mo
                if (i + 1 < 2 * MT * MT) B[(i + 1) & 1] = __builtin_bit_cast(X8, myl[(i + 1) * 64]);
                const int ks = i / MT, mo = i % MT;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (!(WAVE_ABL & 4)) Y[nt][mo] = fq_mfma32<T>(Uh[nt][ks], B[i & 1], Y[nt][mo]);
                    else if (WAVE_ABL & 32) asm volatile("" : : "v"(B[i & 1]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (out.rt_flags & FQ_ROUND_Y_F16) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Y[nt][mo][r] = (float)(T)Y[nt][mo][r];
        }

        // ---- per-token extrema over the valid entries ----
        // (one independent max3 / min3 chain per tile: a single running pair is a 128-deep dependent chain, which a
        //  CU holding only ~2 waves per SIMD cannot hide)
        float vmax = -INFINITY, vmin = INFINITY;
        {
            float pmax[NT][MT], pmin[NT][MT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mo = 0; mo < MT; ++mo) {
                    const f32x16& t = Y[nt][mo];
                    float a = FqMaxOp()(t[0], t[1]), b = FqMinOp()(t[0], t[1]);
#pragma unroll
                    for (int r = 2; r < 16; r += 2) {
                        a = fq_max3(a, t[r], t[r + 1]);
                        b = fq_min3(b, t[r], t[r + 1]);
                    }
                    const bool ok = ((N == NT * 32) || (h * NT * 16 + nt * 16) < N) && (mo * 32 + c) < M;
                    pmax[nt][mo] = ok ? a : -INFINITY;
                    pmin[nt][mo] = ok ? b : INFINITY;
                }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mo = 0; mo < MT; ++mo) {
                    vmax = fmaxf(vmax, pmax[nt][mo]);
                    vmin = fminf(vmin, pmin[nt][mo]);
                }
        }
        vmax = fq_wave_max(vmax);
        vmin = fq_wave_min(vmin);

        if constexpr (OS != FQ_OUT_PACKED) {
            // 16-bit outputs: piece (nt, b) of row m' = 32 mo + c holds n' = h*NT*16 + nt*16 + 8 b .. + 7. Straight from the fragments a
            // store instruction puts 16 bytes into each of 64 different lines (measured: 64 x 128 fake-quant 161 us that way, 121 on the
            // workgroup-per-token kernel). The pieces go through the wave's own token buffer instead — in the layout the DMA left the
            // token in (same chunk rotation: conflict-free writes) — and leave as 1 KB-contiguous stores with the DMA's own per-lane
            // offsets (stage_flush): the inverse of the staging copy.
            auto piece_ok = [&](int nt, int mo) { return ((N == NT * 32) || (h * NT * 16 + nt * 16) < N) && (mo * 32 + c) < M; };
            uint4* stg = reinterpret_cast<uint4*>(tokbuf);
            auto stage_put = [&](int nt, int mo, int b, uint4 v) {
                const int row = mo * 32 + c;
                stg[row * CPR + ((h * NT * 2 + nt * 2 + b) ^ sw)] = v;
            };
            auto stage_flush = [&](T* dst) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                unsigned char* g = reinterpret_cast<unsigned char*>(dst);
                for (int i = 0; i < n_dma; ++i) {
                    const uint4 v = stg[i * 64 + lane];
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(g + (int64_t)i * 1024 + voff[i & 3]));
                }
            };
            if constexpr (OS == FQ_OUT_TRANSFORM) {
#pragma unroll
                for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            X8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = (T)Y[nt][mo][b * 8 + e];
                            if (piece_ok(nt, mo)) stage_put(nt, mo, b, __builtin_bit_cast(uint4, o));
                        }
                stage_flush(reinterpret_cast<T*>(out.y) + tok * ((int64_t)M * N));
            } else {
                for (int ci = 0; ci < out.n_clips; ++ci) {
                    float sig_max, sig_min;
                    fq_token_sigs(out, ci, tok, gcur, sig_max, sig_min);
                    const float scale = fq_token_scale<0, T>(vmax, vmin, sig_max, sig_min, out.rt_flags);
                    const float inv = fq_fast_inv(scale);
                    const bool magic = fq_magic_ok(vmax, vmin, inv), clampq = fq_needs_clamp(vmax, vmin, inv);
                    float dmax = 0.0f;
                    if (magic) {
#pragma unroll
                        for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                                for (int b = 0; b < 2; ++b) {
                                    const f32x16& t = Y[nt][mo];
                                    const int k = b * 8;
                                    const u32x4 o = clampq ? fq_fake8<true, T>(t[k], t[k + 1], t[k + 2], t[k + 3], t[k + 4], t[k + 5], t[k + 6], t[k + 7], inv, scale, dmax)
                                                           : fq_fake8<false, T>(t[k], t[k + 1], t[k + 2], t[k + 3], t[k + 4], t[k + 5], t[k + 6], t[k + 7], inv, scale, dmax);
                                    if (piece_ok(nt, mo)) stage_put(nt, mo, b, __builtin_bit_cast(uint4, o));
                                }
                    }
                    if (!magic || fq_wave_needs_exact(dmax)) {   // rare: the whole token again with the true division
#pragma unroll
                        for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                                for (int b = 0; b < 2; ++b) {
                                    X8 o;
#pragma unroll
                                    for (int e = 0; e < 8; ++e) o[e] = fq_fake<T>(scale, fq_qexact(Y[nt][mo][b * 8 + e], scale));
                                    if (piece_ok(nt, mo)) stage_put(nt, mo, b, __builtin_bit_cast(uint4, o));
                                }
                    }
                    stage_flush(reinterpret_cast<T*>(out.fq[ci]) + tok * ((int64_t)M * N));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the next clip set rewrites the stage)
                }
            }
            pull_next();   // the buffer is free again: the next token's DMA (the other waves of the CU cover its latency)
            slot = next_pulled;
            continue;
        }

        for (int ci = 0; ci < out.n_clips; ++ci) {
            float sig_max, sig_min;
            fq_token_sigs(out, ci, tok, gcur, sig_max, sig_min);
            const bool g128 = (NT == 2 && N == 64) && (out.rt_flags & FQ_GROUP128);  // (the launcher admits it for N = 64 only)
            float scale = 0.0f, inv = 0.0f;
            bool magic = true, clampq = false;
            if (!g128) {
                scale = fq_token_scale<0>(vmax, vmin, sig_max, sig_min, out.rt_flags);
                inv = fq_fast_inv(scale);
                magic = fq_magic_ok(vmax, vmin, inv);
                clampq = fq_needs_clamp(vmax, vmin, inv);
                if (lane == 0) reinterpret_cast<T*>(out.scale[ci])[tok] = (T)scale;
            }
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) {
                if (g128) {
                    // one scale per 128 consecutive elements = output rows (2j, 2j+1): this lane's row 32 mo + c and its
                    // neighbour's, both column halves (ActivationQuantizer(groupsize=128) reshapes to (-1, 128))
                    float a = -INFINITY, b = INFINITY;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const f32x16& t = Y[nt][mo];
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            a = fq_max3(a, t[r], t[r + 1]);
                            b = fq_min3(b, t[r], t[r + 1]);
                        }
                    }
                    const float gmax = fq_group4_reduce(a, FqMaxOp()), gmin = fq_group4_reduce(b, FqMinOp());
                    scale = fq_token_scale<0>(gmax, gmin, sig_max, sig_min, out.rt_flags);
                    inv = fq_fast_inv(scale);
                    magic = !__any(!fq_magic_ok(gmax, gmin, inv));    // wave-uniform route: the slowest any lane needs
                    clampq = __any(fq_needs_clamp(gmax, gmin, inv)) != 0;
                    if (h == 0 && !(c & 1) && (mo * 32 + c) < M)
                        reinterpret_cast<T*>(out.scale[ci])[tok * (int64_t)(M * N / 128) + ((mo * 32 + c) >> 1)] = (T)scale;
                }
                uint32_t pw[NT * 2];  // dword nt*2 + w: elements n' = h*NT*16 + nt*16 + 8w .. +8 of row 32 mo + c
                unsigned near = (1u << (NT * 2)) - 1;  // !magic: quotients too large for the magic-number rounding
                if (WAVE_ABL & 1) {
                    near = 0;
#pragma unroll
                    for (int k = 0; k < NT * 2; ++k) pw[k] = __builtin_bit_cast(uint32_t, Y[k >> 1][mo][(k & 1) * 8]);
                } else if (magic) {
                    if (clampq) near = quant_row<true, NT, MT>(Y, mo, inv, pw);
                    else near = quant_row<false, NT, MT>(Y, mo, inv, pw);
                }
                if (near) {  // rare: redo the flagged dwords with the true division
#pragma unroll
                    for (int k = 0; k < NT * 2; ++k) {
                        if (near & (1u << k)) {
                            const f32x16& t = Y[k >> 1][mo];
                            const int b = (k & 1) * 8;
                            pw[k] = fq_pack8p(f32x2{fq_qexact(t[b + 0], scale), fq_qexact(t[b + 1], scale)},
                                              f32x2{fq_qexact(t[b + 2], scale), fq_qexact(t[b + 3], scale)},
                                              f32x2{fq_qexact(t[b + 4], scale), fq_qexact(t[b + 5], scale)},
                                              f32x2{fq_qexact(t[b + 6], scale), fq_qexact(t[b + 7], scale)});
                        }
                    }
                }
                // the lane's run of row m' = 32 mo + c starts at n' = h*NT*16: NT*8 bytes, valid up to N
                if ((mo * 32 + c) < M) {
                    uint8_t* qrow = out.q[ci] + tok * ((int64_t)M * N / 2) + (int64_t)(mo * 32 + c) * (N / 2) + h * (NT * 8);
                    const int nvalid = h ? G::V1 : NT * 16;  // valid elements of the run
#pragma unroll
                    for (int p = 0; p < G::NPIECE; ++p) {
                        constexpr int LAST = NT * 2 - 1;  // (indices beyond the run only appear in branches that are never taken)
                        if (nvalid >= p * 32 + 32)
                            *reinterpret_cast<uint4*>(qrow + p * 16) =
                                make_uint4(pw[4 * p], pw[4 * p + 1 < LAST ? 4 * p + 1 : LAST], pw[4 * p + 2 < LAST ? 4 * p + 2 : LAST],
                                           pw[4 * p + 3 < LAST ? 4 * p + 3 : LAST]);
                        else if (nvalid >= p * 32 + 16)
                            *reinterpret_cast<uint2*>(qrow + p * 16) = make_uint2(pw[4 * p], pw[4 * p + 1]);
                    }
                }
            }
        }
        slot = next_pulled;
    }
}

template <int MT, int NT, int KS1, int W, bool RMS = false, typename T = f16, int OS = FQ_OUT_PACKED>
int launch_wave(const T* x, const uint4* ws, int64_t rows, int M, const FqQuantOut& out, int n_cu, hipStream_t stream) {
    typedef WaveGeom<MT, NT, KS1, W> G;
    static_assert(G::LDS <= 160 * 1024, "LDS budget");
    int64_t blocks = (rows + W - 1) / W;
    if (blocks > n_cu) blocks = n_cu;  // one persistent workgroup per CU
    if (blocks < 1) blocks = 1;
    const int64_t tpb = (rows + blocks - 1) / blocks;
    hipLaunchKernelGGL((fq_kron_wave_kernel<MT, NT, KS1, W, RMS, T, OS>), dim3((unsigned)blocks), dim3(W * 64), 0, stream, x, ws, rows,
                       tpb, M, out);
    return (int)hipGetLastError();
}


// (round 5, built, measured and removed: this kernel on v_mfma_f32_16x16x32_{f16,bf16} — under the power limit the 16x16x32 shape sustains
//  2.07 PFLOP/s where the 32x32x16 shape sustains 1.78 (half the accumulator traffic per FLOP: tools/mfma_energy.hip, profiles/r05_mfma_energy.txt),
//  and every dense kernel here is bound by joules. A full port — 16-row / 16-column tiles, K-steps of 32, the R image's columns permuted so that a lane
//  group ends up with N/4 consecutive n' of its output row, L's image re-paired so that two C tiles of GEMM 1 ARE the A operand of GEMM 2, both built in
//  the prologue from the same workspace, 1 KB-contiguous packed stores — passed 140 parity cases on its first run (fp16 bit-equal to the
//  workgroup-per-token kernel; bf16 differs from it in accumulation order, as a different instruction may) and measured 64 x 128 packed 78.4 us against
//  79-81, bf16 74.3 against 74.9, with three clip sets 124.4 against 117.0, C4's RMSNorm launches 139 / 115 against 129 / 108 (244 VGPRs): twice as many
//  MFMA instructions for the same FLOPs cost issue slots (~14 cycles each, docs/DESIGN_LOG.md section 9) that eat the energy the shape saves.
//  profiles/r05_wave16.txt; the lane maps and image re-indexings: docs/experiments/fq_kron_wave16_kernel.hip.txt)

}  // namespace

// Returns -1000 when the shape / output set is not one this kernel covers (the caller falls back to the generic one).
// ws: fragment workspace already filled by fq_kron_prepare_kernel (rfrag [NT][KS1][64], lfrag [2MT][MT][64]).
template <typename T>
static int launch_kron_wave_t(int flags, const T* x, const void* ws, const void* diag, int64_t rows, int M, int N,
                              const FqQuantOut& out, int n_cu, hipStream_t stream) {
    const bool rms = (flags & FQ_IN_RMSNORM) != 0;
    const int ct = flags & FQ_CT_MASK & ~FQ_IN_RMSNORM;
#ifndef WAVE_OS_MAXKS
#define WAVE_OS_MAXKS 5   // 16-bit outputs on this kernel up to N = 16 x this = 80 (measured, profiles/r04_wave_fakequant.txt: the wider pairs — 128 elements per lane, two waves per SIMD — are faster on the workgroup-per-token kernel: 64 x 128 fake-quant 145 vs 121 us)
#endif
    const bool os_ok = N / 16 <= WAVE_OS_MAXKS;
    const bool fq_only = os_ok && ct == FQ_OUT_FAKEQUANT && !rms && !(out.rt_flags & FQ_GROUP128);      // (round 4) 16-bit outputs, fp32 quantiser
    const bool y_only = os_ok && (ct == FQ_OUT_TRANSFORM || ct == (FQ_OUT_TRANSFORM | FQ_QUANT_F16)) && !rms && !(out.rt_flags & FQ_GROUP128) && out.n_clips == 0;
    if ((ct != FQ_OUT_PACKED && !fq_only && !y_only) || diag != nullptr) return -1000;
    if ((fq_only && !out.fq[0]) || (y_only && !out.y)) return -1000;
    if (M < 1 || M > 64 || (N & 15) || ((M * (N / 8)) & 63)) return -1000;
    if ((out.rt_flags & FQ_GROUP128) && (N != 64 || (M & 1))) return -1000;  // groups = pairs of 64-element rows
    const int MT = (M + 31) / 32, KS1 = N / 16;
    const uint4* w = reinterpret_cast<const uint4*>(ws);
    if (rms && !FqVec<T>::is_f16) return -1000;   // (deploy.nn.RMSNorm is an fp16 module: no bf16 instantiation with the norm)
#define FQ_W(MT_, NT_, KS1_, W_)                                                                          \
    if (MT == MT_ && KS1 == KS1_) {                                                                      \
        if constexpr (FqVec<T>::is_f16) {                                                                \
            if (rms) return launch_wave<MT_, NT_, KS1_, W_, true, T>(x, w, rows, M, out, n_cu, stream);  \
        }                                                                                                \
        if (fq_only) return launch_wave<MT_, NT_, KS1_, W_, false, T, FQ_OUT_FAKEQUANT>(x, w, rows, M, out, n_cu, stream); \
        if (y_only) return launch_wave<MT_, NT_, KS1_, W_, false, T, FQ_OUT_TRANSFORM>(x, w, rows, M, out, n_cu, stream);   \
        return launch_wave<MT_, NT_, KS1_, W_, false, T>(x, w, rows, M, out, n_cu, stream);              \
    }
    FQ_W(2, 4, 8, 7)    // 64x128
    FQ_W(2, 4, 7, 8)    // 64x112
    FQ_W(2, 3, 5, 12)   // 64x80 (5120 = Qwen2.5-14B/32B hidden, in the reference's benchmark list: kernel_benchmark.py:234-246)
    FQ_W(2, 2, 4, 16)   // 56x64, (64x64 stays with fq_kron64.hip)
    FQ_W(1, 2, 4, 16)   // 32x64
#undef FQ_W
    return -1000;
}

int fq_launch_kron_wave(int flags, const f16* x, const void* ws, const f16* diag, int64_t rows, int M, int N,
                        const FqQuantOut& out, int n_cu, hipStream_t stream) {
    return launch_kron_wave_t<f16>(flags, x, ws, diag, rows, M, N, out, n_cu, stream);
}

// the bf16 instantiations (called from fq_launch_kron_generic_bf16, fq_kron_generic2.hip; 64 x 64 stays with fq_kron64.hip)
int fq_launch_kron_wave_bf16(int flags, const void* x, const void* ws, const void* diag, int64_t rows, int M, int N,
                             const FqQuantOut& out, int n_cu, hipStream_t stream) {
    if (M == 64 && N == 64) return -1000;
    if (out.rt_flags & FQ_GROUP128) return -1000;   // (bf16 group-128 launches: the workgroup kernel's group epilogue)
    return launch_kron_wave_t<bf16>(flags, (const bf16*)x, ws, diag, rows, M, N, out, n_cu, stream);
}
