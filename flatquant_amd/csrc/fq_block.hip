// fq_block.hip — single-matrix transform over the last axis of [R, C] blocks, fused with per-block INT4
// quantisation: the o_proj input transform across heads.
//
// Replaces deploy/kernels/block_matmul.py:29-104 (Triton matmul_quant_kernel: b [B, hd, H] @ c [H, H], the
// quantised block TRANSPOSED before packing, :86-101) and the path-A op llama_utils.py:275-277.
//
//   Y[t] = x[t] ([R, C] row-major, R = head_dim, C = num_heads) . P ([C, C]); statistics over all R*C values;
//   packed / fp16 outputs in the reference's transposed order [C][R] per token (transpose_out = 1).
//
// One wave per token. MFMA 32x32x16: A = 32 rows of x (k = C, contiguous: 16-byte fragment loads straight
// from HBM), B = P fragments (LDS, fragment order, built once per workgroup). The A rows a lane supplies are
// PERMUTED (free: each lane picks the row it loads) so that in the C/D fragment lane (h, c') ends up with the
// 64 consecutive rows r = 64h .. 64h+63 of output column c' — i.e. one contiguous run of the transposed
// output, stored with 16-byte stores. Supported: R in {32, 64, 96, 128}, C in {32, 64}.
//
// C = 32: a row is 64 bytes and the two k-halves of the 32 lanes cover it, so the fragment loads are 2 KB contiguous
// per instruction. C = 64 (Llama-2-70B: 64 heads): the same loads would take 32 bytes out of each of 32 different
// 128-byte lines per instruction (4x the address work, measured 150 us against 45 us for the same bytes per
// element) -> DMA variant: the token goes HBM -> LDS in whole 1 KB pieces (fq_dma.hpp, source-side swizzle) into a
// wave-private buffer, the A fragments are conflict-free ds_read_b128, and the next token's DMA is issued as soon as
// the GEMM has read the current one.
#include "fq_common.hpp"
#include "fq_dma.hpp"

namespace {

__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// physical row supplied by A-fragment lane index i of row tile rt (RT tiles): see header comment
__device__ __forceinline__ int rmap(int RT, int rt, int i) {
    return ((i >> 2) & 1) * (RT * 16) + rt * 16 + (i & 3) + 4 * (i >> 3);
}

// NAT = false: outputs in the reference kernel's transposed order [C][R] (block_matmul.py:86-101): X rows are the MFMA A
// operand (row order permuted by rmap), P the B operand, lane (h, c) ends with column c' = 32 ct + c and a run of rows.
// NAT = true: natural order [R][C] = inp.reshape(-1, C) @ P, the contract of {SVD,Inv}SingleTransMatrix.forward
// (trans_utils.py:21-25): the operands swap roles — P^T rows (permuted the same way) are the A operand, X^T the B
// operand, whose fragment is the very same 16-byte piece of row (32 rt + c) — and lane (h, c) ends with row 32 rt + c and
// a run of CT*16 consecutive columns. Either way a lane stores contiguous runs; the epilogue below is written once in
// terms of (outer tile o = its major index / 32, inner tiles i along its run).
// PK: packed-only launches (the deploy contract), compiled separately: the output-set branches and their register copies drop
// out, the quantiser is the single-width asm one of fq_common.hpp (fq_quant8_two), the extrema run as independent max3 / min3
// chains, and — for C = 32, where the token comes straight from HBM — the next token's fragments are prefetched into
// registers behind the GEMM so that they land during the epilogue (the loop used to load and multiply in the same breath).
template <int RT, int CT, bool DMA, bool NAT, bool PK>
__global__ __launch_bounds__(256, (PK && CT == 1) ? (RT == 4 ? 3 : 4) : (PK ? 2 : 1)) void fq_block_kernel(const f16* __restrict__ x, const f16* __restrict__ P,
                                                       int64_t rows, FqQuantOut out, int flags_rt) {
    const int flags = PK ? (FQ_OUT_PACKED | (flags_rt & ~(FQ_OUT_FAKEQUANT | FQ_OUT_TRANSFORM | FQ_QUANT_F16))) : flags_rt;
    constexpr int R = RT * 32, C = CT * 32, KS = C / 16, D = R * C, CPR = C / 8;
    constexpr int OC = NAT ? RT : CT, IC = NAT ? CT : RT, LEN = IC * 32;  // outer / inner tile counts, length of a major row
    static_assert(!DMA || CPR == 8, "the DMA variant is the C = 64 one");
    __shared__ __attribute__((aligned(16))) uint4 pfrag[KS * CT * 64];  // [(s*CT + ct)][lane]
    __shared__ __attribute__((aligned(16))) unsigned char tokmem[DMA ? 4 * D * 2 : 16];  // wave-private token buffers
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    for (int item = tid; item < KS * CT * 64; item += 256) {
        const int f = item >> 6, ln = item & 63, fh = ln >> 5, fc = ln & 31;
        const int s = f / CT, ct = f - s * CT;
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = P[(s * 16 + fh * 8 + j) * C + (NAT ? rmap(CT, ct, fc) : ct * 32 + fc)];
        pfrag[item] = __builtin_bit_cast(uint4, v);
    }
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (tid >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    unsigned char* tokbuf = tokmem + (DMA ? (tid >> 6) * D * 2 : 0);
    const unsigned tok_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)tokbuf);
    unsigned voff[4] = {0, 0, 0, 0};
    if (DMA) {
        dma_offsets<8>(lane, voff);
        if (wave_id < rows) dma_token<8>(x, wave_id, (int64_t)D * 2, D * 2 / 1024, tok_lds, voff);
    }
    __syncthreads();  // (the compiler knows of no VMEM in flight here: lgkmcnt(0) + s_barrier)
    // packed-only single-clip launches: the stores a token issues after its successor's DMA are a fixed number, so the
    // wait for that DMA can leave them in flight
    constexpr int PACKED_STORES = OC * ((IC + 1) / 2) + 1;
    const bool counted = DMA && (flags & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT | FQ_OUT_TRANSFORM)) == FQ_OUT_PACKED &&
                         out.n_clips == 1;
    bool first = true;
    constexpr bool PFR = PK && !DMA;      // register prefetch of the next token's A fragments
    u32x4 PF[PFR ? RT : 1][PFR ? KS : 1];
    // (inline asm + a hand-placed counted wait: the fragments of the NEXT token are in flight across the epilogue, whose stores
    //  are issued after them and may stay in flight at the wait)
#define FQ_BLOCK_PF(t)                                                                                              \
    {                                                                                                               \
        _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                                         \
            const int row_ = NAT ? rt * 32 + c : rmap(RT, rt, c);                                                   \
            const u32x4* xp_ = reinterpret_cast<const u32x4*>(x + (t) * D + (int64_t)row_ * C + h * 8);             \
            _Pragma("unroll") for (int s = 0; s < KS; ++s)                                                          \
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(PF[PFR ? rt : 0][PFR ? s : 0]) : "v"(xp_), "n"(s * 32) : "memory"); \
        }                                                                                                           \
    }
    if (PFR && wave_id < rows) FQ_BLOCK_PF(wave_id)

    for (int64_t tok = wave_id; tok < rows; tok += n_waves) {
        if (PFR) {
            if (!first && out.n_clips == 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PACKED_STORES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            first = false;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(PF[PFR ? rt : 0][PFR ? s : 0]));  // arrived with the wait above
        }
        if (DMA) {
            if (counted && !first) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PACKED_STORES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            first = false;
        }
        int foff = lane;
        asm volatile("" : "+v"(foff));  // keep the fragment reads inside the loop (see fq_kron64.hip)
        const uint4* myp = pfrag + foff;

        f32x16 Y[RT][CT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int row = NAT ? rt * 32 + c : rmap(RT, rt, c);
            const uint4* xp = reinterpret_cast<const uint4*>(x + tok * D + (int64_t)row * C + h * 8);
            const uint4* lp = reinterpret_cast<const uint4*>(tokbuf) + row * CPR;  // chunk (2 s + h) ^ swz(row)
            const int sw = DMA ? swz<8>(row) : 0;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) Y[rt][ct] = f32x16{0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                f16x8 a;  // k = 16 s + 8 h .. +8
                if (DMA) a = __builtin_bit_cast(f16x8, lp[(s * 2 + h) ^ sw]);
                else if (PFR) a = __builtin_bit_cast(f16x8, PF[PFR ? rt : 0][PFR ? s : 0]);
                else a = __builtin_bit_cast(f16x8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(xp) + s * 2));
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const f16x8 pf = __builtin_bit_cast(f16x8, myp[(s * CT + ct) * 64]);
                    Y[rt][ct] = NAT ? mfma32(pf, a, Y[rt][ct]) : mfma32(a, pf, Y[rt][ct]);
                }
            }
        }
        if (PFR && tok + n_waves < rows) {  // the fragments have been consumed: the next token's land during the epilogue
            __builtin_amdgcn_sched_barrier(0);
            FQ_BLOCK_PF(tok + n_waves)
        }
        if (DMA) {  // the buffer has been read: fetch this wave's next token while the current one is quantised
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (tok + n_waves < rows) dma_token<8>(x, tok + n_waves, (int64_t)D * 2, D * 2 / 1024, tok_lds, voff);
        }
        // lane (h, c) now holds, for column c' = 32 ct + c, rows r = RT*16*h + 16 rt + reg   (NAT: for row 32 rt + c,
        // columns c' = CT*16*h + 16 ct + reg): major index 32 o + c, run of IC*16 elements starting at IC*16*h
#define FQ_TILE(o, i) (NAT ? Y[o][i] : Y[i][o])

        if (flags & FQ_ROUND_Y_F16) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Y[rt][ct][r] = (float)(f16)Y[rt][ct][r];
        }
        if (flags & FQ_OUT_TRANSFORM) {
#pragma unroll
            for (int o = 0; o < OC; ++o) {
                uint4* yp = reinterpret_cast<uint4*>(out.y + tok * D + (int64_t)(o * 32 + c) * LEN + h * (IC * 16));
#pragma unroll
                for (int i = 0; i < IC; ++i)
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        f16x8 v;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (f16)FQ_TILE(o, i)[w * 8 + e];
                        yp[i * 2 + w] = __builtin_bit_cast(uint4, v);
                    }
            }
        }
        if (PK) {
            float pmx[RT * CT], pmn[RT * CT];  // one independent max3 / min3 chain per tile
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const f32x16& t = Y[rt][ct];
                    float a = FqMaxOp()(t[0], t[1]), b = FqMinOp()(t[0], t[1]);
#pragma unroll
                    for (int r = 2; r < 16; r += 2) {
                        a = fq_max3(a, t[r], t[r + 1]);
                        b = fq_min3(b, t[r], t[r + 1]);
                    }
                    pmx[rt * CT + ct] = a;
                    pmn[rt * CT + ct] = b;
                }
            float vmax = pmx[0], vmin = pmn[0];
#pragma unroll
            for (int k = 1; k < RT * CT; ++k) {
                vmax = fmaxf(vmax, pmx[k]);
                vmin = fminf(vmin, pmn[k]);
            }
            vmax = fq_wave_max(vmax);
            vmin = fq_wave_min(vmin);
            for (int ci = 0; ci < out.n_clips; ++ci) {
                const float scale = fq_token_scale<0>(vmax, vmin, out.sig_max[ci], out.sig_min[ci], flags);
                const float inv = fq_fast_inv(scale);
                const bool magic = fq_magic_ok(vmax, vmin, inv), clampq = fq_needs_clamp(vmax, vmin, inv);
                const float ilo = fq_inv_lo(inv), ihi = fq_inv_hi(inv);
                if (lane == 0) out.scale[ci][tok] = (f16)scale;
#pragma unroll
                for (int o = 0; o < OC; ++o) {
                    uint8_t* qrow = out.q[ci] + tok * (D / 2) + (int64_t)(o * 32 + c) * (LEN / 2) + h * (IC * 8);
                    uint2 pk[IC];
#pragma unroll
                    for (int i = 0; i < IC; ++i) {
                        const f32x16& t = FQ_TILE(o, i);
                        unsigned long long d0 = ~0ull, d1 = ~0ull;
                        pk[i] = uint2{0u, 0u};
                        if (magic) {
                            if (clampq) {
                                pk[i].x = fq_quant8<true>(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], inv, ilo, ihi, d0);
                                pk[i].y = fq_quant8<true>(t[8], t[9], t[10], t[11], t[12], t[13], t[14], t[15], inv, ilo, ihi, d1);
                            } else {
                                pk[i].x = fq_quant8<false>(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], inv, ilo, ihi, d0);
                                pk[i].y = fq_quant8<false>(t[8], t[9], t[10], t[11], t[12], t[13], t[14], t[15], inv, ilo, ihi, d1);
                            }
                        }
                        if (d0)  // rare: an ambiguous digit somewhere in the wave -> the true division for this dword
                            pk[i].x = fq_pack8(fq_qexact(t[0], scale), fq_qexact(t[1], scale), fq_qexact(t[2], scale), fq_qexact(t[3], scale),
                                               fq_qexact(t[4], scale), fq_qexact(t[5], scale), fq_qexact(t[6], scale), fq_qexact(t[7], scale));
                        if (d1)
                            pk[i].y = fq_pack8(fq_qexact(t[8], scale), fq_qexact(t[9], scale), fq_qexact(t[10], scale), fq_qexact(t[11], scale),
                                               fq_qexact(t[12], scale), fq_qexact(t[13], scale), fq_qexact(t[14], scale), fq_qexact(t[15], scale));
                        if (i & 1)
                            *reinterpret_cast<uint4*>(qrow + (i - 1) * 8) = make_uint4(pk[i - 1].x, pk[i - 1].y, pk[i].x, pk[i].y);
                        else if (i == IC - 1)
                            *reinterpret_cast<uint2*>(qrow + i * 8) = pk[i];
                    }
                }
            }
            continue;
        }
        if (!(flags & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT))) continue;

        float vmax = Y[0][0][0], vmin = Y[0][0][0];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    vmax = fmaxf(fmaxf(vmax, Y[rt][ct][r]), Y[rt][ct][r + 1]);
                    vmin = fminf(fminf(vmin, Y[rt][ct][r]), Y[rt][ct][r + 1]);
                }
        vmax = fq_wave_max(vmax);
        vmin = fq_wave_min(vmin);

        for (int ci = 0; ci < out.n_clips; ++ci) {
            float scale;
            if (flags & FQ_QUANT_F16) scale = fq_token_scale<FQ_QUANT_F16>(vmax, vmin, out.sig_max[ci], out.sig_min[ci], flags);
            else scale = fq_token_scale<0>(vmax, vmin, out.sig_max[ci], out.sig_min[ci], flags);
            const float inv = fq_fast_inv(scale);
            if ((flags & FQ_OUT_PACKED) && lane == 0) out.scale[ci][tok] = (f16)scale;
#pragma unroll
            for (int o = 0; o < OC; ++o) {
                uint8_t* qrow = (flags & FQ_OUT_PACKED)
                                    ? out.q[ci] + tok * (D / 2) + (int64_t)(o * 32 + c) * (LEN / 2) + h * (IC * 8)
                                    : nullptr;
                f16* frow = (flags & FQ_OUT_FAKEQUANT)
                                ? out.fq[ci] + tok * D + (int64_t)(o * 32 + c) * LEN + h * (IC * 16)
                                : nullptr;
                uint2 pk[IC];
#pragma unroll
                for (int i = 0; i < IC; ++i) {
                    float qv[16];
                    if (flags & FQ_QUANT_F16) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) qv[r] = (float)fq_quant1<FQ_QUANT_F16>(FQ_TILE(o, i)[r], scale);
                    } else {
                        float dmax = 0.0f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) qv[r] = fq_qfast(FQ_TILE(o, i)[r], inv, dmax);
                        if (fq_wave_needs_exact(dmax)) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) qv[r] = fq_qexact(FQ_TILE(o, i)[r], scale);
                        }
                    }
                    if (flags & FQ_OUT_PACKED) {  // 8 bytes per row tile; two tiles share one 16-byte store
                        pk[i].x = fq_pack8(qv[0], qv[1], qv[2], qv[3], qv[4], qv[5], qv[6], qv[7]);
                        pk[i].y = fq_pack8(qv[8], qv[9], qv[10], qv[11], qv[12], qv[13], qv[14], qv[15]);
                        if (i & 1)
                            *reinterpret_cast<uint4*>(qrow + (i - 1) * 8) = make_uint4(pk[i - 1].x, pk[i - 1].y, pk[i].x, pk[i].y);
                        else if (i == IC - 1)
                            *reinterpret_cast<uint2*>(qrow + i * 8) = pk[i];
                    }
                    if (flags & FQ_OUT_FAKEQUANT) {
                        f16x8 v0, v1;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if (flags & FQ_QUANT_F16) {
                                v0[e] = fq_dequant1<FQ_QUANT_F16>((int)qv[e], scale);
                                v1[e] = fq_dequant1<FQ_QUANT_F16>((int)qv[8 + e], scale);
                            } else {
                                v0[e] = fq_fake_f16(scale, qv[e]);
                                v1[e] = fq_fake_f16(scale, qv[8 + e]);
                            }
                        }
                        uint4* fp = reinterpret_cast<uint4*>(frow + i * 16);
                        fp[0] = __builtin_bit_cast(uint4, v0);
                        fp[1] = __builtin_bit_cast(uint4, v1);
                    }
                }
            }
        }
    }
}

#undef FQ_TILE
#undef FQ_BLOCK_PF

// ---------------------------------------------------------------------------------------------------------------------
// Any head count, either element type (round 3). The reference kernel masks arbitrary sizes
// (block_matmul.py:56-66: `% M`, `% N`, masked loads / stores) and its callers meet num_attention_heads = 28 (Qwen2.5-7B),
// 40 (Llama-2-13B, Qwen2.5-14B / 32B), 12 / 14 / 16 (the small Qwen2.5), and the path-A op is dtype-generic
// ({SVD,Inv}SingleTransMatrix.forward, trans_utils.py:21-25; bf16 on Llama-3 / Qwen: model_utils.py:20). This kernel takes
// every even C <= 64 with R in {32, 64, 96, 128} for T in {f16, bf16}: the K dimension and the output columns are padded to
// whole 32-column tiles with ZEROS in the P fragments, the A fragments are loaded with the widest access the row length
// allows and zero beyond the row (a row of 28 heads is 56 bytes: dword pieces), padding columns are excluded from the
// extrema and the stores, and a lane's run of outputs is written with the widest store its address allows (packed rows of
// 14 bytes exist). One wave per token like the tuned kernel above, which keeps C in {32, 64} on fp16.
// ---------------------------------------------------------------------------------------------------------------------
// first `nbytes` (<= 16) bytes of v to dst, any alignment
__device__ __forceinline__ void blk_store_bytes(unsigned char* dst, u32x4 v, int nbytes) {
    if (nbytes <= 0) return;
    const unsigned a = (unsigned)(size_t)dst;
    if (nbytes == 16 && !(a & 15)) {
        *reinterpret_cast<u32x4*>(dst) = v;
    } else if (!((a | nbytes) & 3)) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * k < nbytes) reinterpret_cast<uint32_t*>(dst)[k] = v[k];
    } else if (!((a | nbytes) & 1)) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (2 * k < nbytes) reinterpret_cast<unsigned short*>(dst)[k] = (unsigned short)(v[k >> 1] >> (16 * (k & 1)));
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < nbytes) dst[k] = (unsigned char)(v[k >> 2] >> (8 * (k & 3)));
    }
}

template <int RT, int CT, bool NAT, typename T>
__global__ __launch_bounds__(256) void fq_block_any_kernel(const T* __restrict__ x, const T* __restrict__ P, int64_t rows, int Cv,
                                                           FqQuantOut out, int flags) {
    typedef typename FqVec<T>::x8 X8;
    constexpr int R = RT * 32, KS = CT * 2;
    constexpr int OC = NAT ? RT : CT, IC = NAT ? CT : RT;   // outer / inner tile counts of a lane's output runs
    const int D = R * Cv, LEN = NAT ? Cv : R;               // elements per token, length of a major row of the output
    __shared__ __attribute__((aligned(16))) uint4 pfrag[KS * CT * 64];  // [(s*CT + ct)][lane], zero outside [Cv, Cv]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    for (int item = tid; item < KS * CT * 64; item += 256) {
        const int f = item >> 6, ln = item & 63, fh = ln >> 5, fc = ln & 31;
        const int s = f / CT, ct = f - s * CT;
        const int col = NAT ? rmap(CT, ct, fc) : ct * 32 + fc;
        X8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = s * 16 + fh * 8 + j;
            v[j] = (k < Cv && col < Cv) ? P[k * Cv + col] : (T)0.0f;
        }
        pfrag[item] = __builtin_bit_cast(uint4, v);
    }
    __syncthreads();
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (tid >> 6), n_waves = (int64_t)gridDim.x * 4;
    const bool wide = !(Cv & 7);   // rows are whole 16-byte chunks
    // valid leading elements of tile (o, i) of this lane's run: transposed order masks whole columns c' = 32 o + c, natural
    // order cuts the run of columns IC*16*h + 16 i + reg at Cv
#define FQ_NVAL(o, i) (NAT ? min(16, max(0, Cv - (IC * 16 * h + 16 * (i)))) : (((o) * 32 + c) < Cv ? 16 : 0))
#define FQ_TILE(o, i) (NAT ? Y[o][i] : Y[i][o])
    for (int64_t tok = wave_id; tok < rows; tok += n_waves) {
        int foff = lane;
        asm volatile("" : "+v"(foff));  // keep the fragment reads inside the loop (see fq_kron64.hip)
        const uint4* myp = pfrag + foff;
        f32x16 Y[RT][CT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int row = NAT ? rt * 32 + c : rmap(RT, rt, c);
            const T* xr = x + tok * D + (int64_t)row * Cv;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) Y[rt][ct] = f32x16{0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (16 * s >= Cv) continue;              // (wave-uniform) a K-step of nothing but padding
                const int k0 = 16 * s + 8 * h, nv = Cv - k0;   // this lane's 8 k-slots, valid ones (even; may be <= 0)
                u32x4 w = {0u, 0u, 0u, 0u};
                if (wide) {
                    if (nv >= 8) w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(xr + k0));
                } else {
                    const uint32_t* xp = reinterpret_cast<const uint32_t*>(xr + k0);   // Cv even: rows are dword-aligned
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (2 * k < nv) w[k] = xp[k];
                }
                const X8 a = __builtin_bit_cast(X8, w);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const X8 pf = __builtin_bit_cast(X8, myp[(s * CT + ct) * 64]);
                    Y[rt][ct] = NAT ? fq_mfma32<T>(pf, a, Y[rt][ct]) : fq_mfma32<T>(a, pf, Y[rt][ct]);
                }
            }
        }
        if (flags & FQ_ROUND_Y_F16) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Y[rt][ct][r] = (float)(T)Y[rt][ct][r];
        }
        if (flags & FQ_OUT_TRANSFORM) {
#pragma unroll
            for (int o = 0; o < OC; ++o) {
                unsigned char* yp = reinterpret_cast<unsigned char*>(reinterpret_cast<T*>(out.y) + tok * D + (int64_t)(o * 32 + c) * LEN + h * (IC * 16));
#pragma unroll
                for (int i = 0; i < IC; ++i) {
                    const int nval = FQ_NVAL(o, i);
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        X8 v;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (T)FQ_TILE(o, i)[w * 8 + e];
                        blk_store_bytes(yp + (i * 16 + w * 8) * 2, __builtin_bit_cast(u32x4, v), 2 * min(8, nval - 8 * w));
                    }
                }
            }
        }
        if (!(flags & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT))) continue;

        float vmax = -INFINITY, vmin = INFINITY;
#pragma unroll
        for (int o = 0; o < OC; ++o)
#pragma unroll
            for (int i = 0; i < IC; ++i) {
                const int nval = FQ_NVAL(o, i);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (r < nval) {
                        vmax = fmaxf(vmax, FQ_TILE(o, i)[r]);
                        vmin = fminf(vmin, FQ_TILE(o, i)[r]);
                    }
            }
        vmax = fq_wave_max(vmax);
        vmin = fq_wave_min(vmin);

        for (int ci = 0; ci < out.n_clips; ++ci) {
            float scale;
            if (flags & FQ_QUANT_F16) scale = fq_token_scale<FQ_QUANT_F16, T>(vmax, vmin, out.sig_max[ci], out.sig_min[ci], flags);
            else scale = fq_token_scale<0, T>(vmax, vmin, out.sig_max[ci], out.sig_min[ci], flags);
            const float inv = fq_fast_inv(scale);
            const bool magic = !(flags & FQ_QUANT_F16) && fq_magic_ok(vmax, vmin, inv);
            if ((flags & FQ_OUT_PACKED) && lane == 0) reinterpret_cast<T*>(out.scale[ci])[tok] = (T)scale;
#pragma unroll
            for (int o = 0; o < OC; ++o) {
                unsigned char* qrow = (flags & FQ_OUT_PACKED)
                                          ? out.q[ci] + tok * (D / 2) + (int64_t)(o * 32 + c) * (LEN / 2) + h * (IC * 8)
                                          : nullptr;
                unsigned char* frow = (flags & FQ_OUT_FAKEQUANT)
                                          ? reinterpret_cast<unsigned char*>(reinterpret_cast<T*>(out.fq[ci]) + tok * D + (int64_t)(o * 32 + c) * LEN + h * (IC * 16))
                                          : nullptr;
#pragma unroll
                for (int i = 0; i < IC; ++i) {
                    const int nval = FQ_NVAL(o, i);
                    float qv[16];
                    if (flags & FQ_QUANT_F16) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) qv[r] = (float)fq_quant1<FQ_QUANT_F16, T>(FQ_TILE(o, i)[r], scale);
                    } else {
                        float dmax = 1.0f;  // !magic: quotients too large for the fast rounding -> the true division
                        if (magic) {
                            dmax = 0.0f;
#pragma unroll
                            for (int r = 0; r < 16; ++r) qv[r] = fq_qfast(r < nval ? FQ_TILE(o, i)[r] : 0.0f, inv, dmax);
                        }
                        if (fq_wave_needs_exact(dmax)) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) qv[r] = fq_qexact(FQ_TILE(o, i)[r], scale);
                        }
                    }
                    if (flags & FQ_OUT_PACKED) {
                        const u32x4 pk = {fq_pack8(qv[0], qv[1], qv[2], qv[3], qv[4], qv[5], qv[6], qv[7]),
                                          fq_pack8(qv[8], qv[9], qv[10], qv[11], qv[12], qv[13], qv[14], qv[15]), 0u, 0u};
                        blk_store_bytes(qrow + i * 8, pk, nval >> 1);
                    }
                    if (flags & FQ_OUT_FAKEQUANT) {
                        X8 v0, v1;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if (flags & FQ_QUANT_F16) {
                                v0[e] = fq_dequant1<FQ_QUANT_F16, T>((int)qv[e], scale);
                                v1[e] = fq_dequant1<FQ_QUANT_F16, T>((int)qv[8 + e], scale);
                            } else {
                                v0[e] = fq_fake<T>(scale, qv[e]);
                                v1[e] = fq_fake<T>(scale, qv[8 + e]);
                            }
                        }
                        blk_store_bytes(frow + i * 32, __builtin_bit_cast(u32x4, v0), 2 * min(8, nval));
                        blk_store_bytes(frow + i * 32 + 16, __builtin_bit_cast(u32x4, v1), 2 * min(8, nval - 8));
                    }
                }
            }
        }
    }
#undef FQ_NVAL
#undef FQ_TILE
}

template <int RT, int CT, bool NAT, typename T>
int launch_block_any(int flags, const T* x, const T* P, int64_t rows, int C, const FqQuantOut& out, int n_cu, hipStream_t stream) {
    int64_t blocks = (rows + 3) / 4;
    if (blocks > (int64_t)n_cu * 2) blocks = (int64_t)n_cu * 2;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((fq_block_any_kernel<RT, CT, NAT, T>), dim3((unsigned)blocks), dim3(256), 0, stream, x, P, rows, C, out, flags);
    return (int)hipGetLastError();
}

template <typename T>
int launch_block_any_t(int flags, const T* x, const T* P, int64_t rows, int R, int C, int transpose_out, const FqQuantOut& out,
                       int n_cu, hipStream_t stream) {
    const int RT = R / 32, CT = (C + 31) / 32;
#define FQ_BA(RT_, CT_)                                                                                          \
    if (RT == RT_ && CT == CT_)                                                                                  \
        return transpose_out ? launch_block_any<RT_, CT_, false, T>(flags, x, P, rows, C, out, n_cu, stream)     \
                             : launch_block_any<RT_, CT_, true, T>(flags, x, P, rows, C, out, n_cu, stream);
    FQ_BA(1, 1) FQ_BA(2, 1) FQ_BA(3, 1) FQ_BA(4, 1) FQ_BA(1, 2) FQ_BA(2, 2) FQ_BA(3, 2) FQ_BA(4, 2)
#undef FQ_BA
    return -1000;
}

template <int RT, int CT, bool NAT>
int launch_block(int flags, const f16* x, const f16* P, int64_t rows, const FqQuantOut& out, int n_cu,
                 hipStream_t stream) {
    constexpr bool DMA = CT == 2;  // C = 64: whole-line DMA staging (see the header)
    int64_t blocks = (rows + 3) / 4;
    const bool pk = (flags & FQ_CT_MASK) == FQ_OUT_PACKED;
    const int64_t cap = (int64_t)n_cu * ((pk && CT == 1) ? (RT == 4 ? 3 : 4) : 2);   // workgroups per CU the registers allow
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (pk)
        hipLaunchKernelGGL((fq_block_kernel<RT, CT, DMA, NAT, true>), dim3((unsigned)blocks), dim3(256), 0, stream, x, P, rows, out, flags);
    else
        hipLaunchKernelGGL((fq_block_kernel<RT, CT, DMA, NAT, false>), dim3((unsigned)blocks), dim3(256), 0, stream, x, P, rows, out, flags);
    return (int)hipGetLastError();
}

}  // namespace

int fq_launch_block(int flags, const f16* x, const f16* P, int64_t rows, int R, int C, int transpose_out,
                    const FqQuantOut& out, int n_cu, hipStream_t stream) {
    if ((R & 31) || R < 32 || R > 128 || (C & 1) || C < 2 || C > 64) return -1000;
    if (flags & FQ_DT_BF16)
        return launch_block_any_t<bf16>(flags & ~FQ_DT_BF16, (const bf16*)x, (const bf16*)P, rows, R, C, transpose_out, out, n_cu, stream);
    if (C != 32 && C != 64) return launch_block_any_t<f16>(flags, x, P, rows, R, C, transpose_out, out, n_cu, stream);
    const int RT = R / 32, CT = C / 32;
#define FQ_B(RT_, CT_)                                                                                       \
    if (RT == RT_ && CT == CT_)                                                                              \
        return transpose_out ? launch_block<RT_, CT_, false>(flags, x, P, rows, out, n_cu, stream)           \
                             : launch_block<RT_, CT_, true>(flags, x, P, rows, out, n_cu, stream);
    FQ_B(1, 1) FQ_B(2, 1) FQ_B(3, 1) FQ_B(4, 1) FQ_B(1, 2) FQ_B(2, 2) FQ_B(3, 2) FQ_B(4, 2)
#undef FQ_B
    return -1000;
}
