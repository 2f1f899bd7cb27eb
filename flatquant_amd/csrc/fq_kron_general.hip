// fq_kron_general.hip — the MFMA Kronecker transform + per-token INT4 quantisation for EVERY factor pair the specialised
// kernels (fq_kron64 / fq_kron_wave / fq_kron_trio / fq_kron_fast) do not take: any M <= 256, any even N <= 256 with
// M * N <= 32768, every output set, diag, grouped launches, post-scale. Qwen2.5's ffn widths are the reason it exists:
// 18944 = 128 x 148, 27648 = 144 x 192, 29568 = 168 x 176 (function_utils.py:11-21 picks the factor pair closest to
// sqrt(d), whatever its divisibility); the reference runs them through the masked Triton kernels
// (deploy/kernels/kron_matmul.py:29-110, block sizes rounded up to a power of two and masked).
//
// Same mathematics, rounding points and fragment chaining as every other Kronecker kernel here (U = X.R rounded to fp16,
// Y^T = U^T.L, the C fragment of GEMM 1 is the A fragment of GEMM 2), on the same workspace image
// (fq_kron_prepare_kernel pads R to 16 K-rows / 32 columns and L to 32 x 32 tiles with zeros). What is general:
//   * run-time N, K-steps and tile counts; M in 32-row tiles up to 8 (template), one n'-tile per wave, 4 or 8 waves;
//   * the token is staged with the widest access its row length allows (16 / 8 / 4 bytes for N % 8 / % 4 / % 2 == 0) into an
//     LDS image whose rows are padded to whole K-steps with zeros that are never overwritten;
//   * a lane's run of 16 output columns may be cut by N (N % 16 != 0): statistics and stores take the valid part only;
//   * output stages are dense in LDS and copied out with the widest store the token size allows; a run is put into the stage
//     with the widest LDS store its address allows (rows of 74 bytes exist).
// R and L fragments come from the workspace through L2 (as in the first-generation kernel this file replaces): this is the
// generality path, not a tuned one — 16384 tokens of 128 x 148 take well under a millisecond instead of the 24 ms of the
// plain-FMA kernel it also replaces.
#include "fq_common.hpp"

namespace {


struct GenGeom {
    int M, N;
    int KS1;    // K-steps of GEMM 1 = ceil(N / 16)
    int NT;     // 32-column n'-tiles = ceil(N / 32)
    int pitch;  // LDS row pitch of the staged token, in 16-byte chunks (odd, >= 2 KS1)
};

// 16 fp16 values of a lane's run -> dense fp16 stage in LDS, `nvalid` of them, at any 2-byte-aligned address
template <typename T>
__device__ __forceinline__ void gen_put_run16(T* p, const typename FqVec<T>::x8& v0, const typename FqVec<T>::x8& v1, int nvalid) {
    const unsigned a = (unsigned)(size_t)p;
    if (nvalid == 16 && !(a & 15)) {
        reinterpret_cast<uint4*>(p)[0] = __builtin_bit_cast(uint4, v0);
        reinterpret_cast<uint4*>(p)[1] = __builtin_bit_cast(uint4, v1);
    } else if (nvalid == 16 && !(a & 7)) {
        const uint4 w0 = __builtin_bit_cast(uint4, v0), w1 = __builtin_bit_cast(uint4, v1);
        reinterpret_cast<uint2*>(p)[0] = make_uint2(w0.x, w0.y);
        reinterpret_cast<uint2*>(p)[1] = make_uint2(w0.z, w0.w);
        reinterpret_cast<uint2*>(p)[2] = make_uint2(w1.x, w1.y);
        reinterpret_cast<uint2*>(p)[3] = make_uint2(w1.z, w1.w);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (e < nvalid) p[e] = v0[e];
            if (8 + e < nvalid) p[8 + e] = v1[e];
        }
    }
}
// 8 bytes of packed nibbles -> dense packed stage in LDS, `nb` of them, at any address
__device__ __forceinline__ void gen_put_run8(unsigned char* p, uint2 pk, int nb) {
    const unsigned a = (unsigned)(size_t)p;
    if (nb == 8 && !(a & 7)) {
        *reinterpret_cast<uint2*>(p) = pk;
    } else if (nb == 8 && !(a & 3)) {
        reinterpret_cast<uint32_t*>(p)[0] = pk.x;
        reinterpret_cast<uint32_t*>(p)[1] = pk.y;
    } else {
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (b < nb) p[b] = (unsigned char)(((b < 4 ? pk.x : pk.y) >> (8 * (b & 3))) & 0xFF);
    }
}
// dense LDS stage -> global, `nbytes` per token (dst = token start: aligned like nbytes, the base being an allocation)
__device__ __forceinline__ void gen_copy_out(unsigned char* dst, const unsigned char* src, int nbytes, int tid, int nthreads) {
    if (!(nbytes & 15)) {
        for (int q = tid; q < (nbytes >> 4); q += nthreads) reinterpret_cast<uint4*>(dst)[q] = reinterpret_cast<const uint4*>(src)[q];
    } else if (!(nbytes & 3)) {
        for (int q = tid; q < (nbytes >> 2); q += nthreads) reinterpret_cast<uint32_t*>(dst)[q] = reinterpret_cast<const uint32_t*>(src)[q];
    } else {
        for (int q = tid; q < nbytes; q += nthreads) dst[q] = src[q];
    }
}

template <int MT, int WAVES, bool LLDS, typename T = f16>
__global__ __launch_bounds__(WAVES * 64, 2) void fq_kron_general_kernel(const T* __restrict__ x, const uint4* __restrict__ ws,
                                                                   const T* __restrict__ diag, int64_t rows, GenGeom g,
                                                                   FqQuantOut out, int flags) {
    typedef typename FqVec<T>::x8 X8;
    typedef typename FqVec<T>::x4 X4;
    typedef typename FqVec<T>::x2 X2;
    constexpr int THREADS = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = g.M, N = g.N, KS1 = g.KS1, NT = g.NT, pitch = g.pitch;
    uint4* xs = reinterpret_cast<uint4*>(smem);                          // [MT*32][pitch] 16-byte chunks
    const int xs_chunks = MT * 32 * pitch;
    unsigned char* obuf = smem + (size_t)xs_chunks * 16;                 // packed output stage: M*N/2 bytes
    float* red = reinterpret_cast<float*>(obuf + ((M * N / 2 + 15) & ~15));  // [2][WAVES]
    uint4* lds_l = reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(red) + 2 * WAVES * sizeof(float) + 32 - ((2 * WAVES * sizeof(float)) & 15));
    const uint4* rfrag = ws;
    const uint4* lfrag_g = ws + (size_t)NT * KS1 * 64;

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave;                                   // this wave's n'-tile (the launcher guarantees NT <= WAVES)
    const int64_t d = (int64_t)M * N;
    const int n0 = h * NT * 16 + nt * 16;                  // this lane's run of 16 output columns starts here
    const int nvalid = nt < NT ? (N - n0 >= 16 ? 16 : (N - n0 > 0 ? N - n0 : 0)) : 0;
    FqGroupCursor gcur;

    for (int i = tid; i < xs_chunks; i += THREADS) xs[i] = make_uint4(0, 0, 0, 0);  // the padding stays zero
    if (LLDS) {  // the L fragment image (2 MT^2 KB) fits next to the token: every wave reads all of it once per sweep
        for (int i = tid; i < 2 * MT * MT * 64; i += THREADS) lds_l[i] = lfrag_g[i];
    }

    // Register prefetch of the next token (MT <= 4: the registers allow it; 16- and 8-byte staging): up to 8 x 16 bytes per
    // thread are requested right after the current token has been staged and written to LDS at the top of the next
    // iteration — the token's HBM latency then overlaps the two GEMMs and the output sweeps instead of preceding them.
    constexpr int NPF = MT <= 4 ? 8 : 1;
    u32x4 PF[NPF];
    const int pieces = !(N & 7) ? M * (N >> 3) : M * (N >> 2);            // 16-byte or 8-byte pieces per token
    const int per_thr = (pieces + THREADS - 1) / THREADS;
    const bool pf16 = MT <= 4 && !(N & 7) && per_thr <= NPF;
    const bool pf8 = MT <= 4 && (N & 7) && !(N & 3) && per_thr <= 2 * NPF;
#define FQ_GEN_PF(t)                                                                                                  \
    {                                                                                                                 \
    int tl_ = tid;                                                                                                    \
    asm volatile("" : "+v"(tl_)); /* per-piece addresses are loop-invariant: hoisted out of the token loop they are spilled */ \
    if (pf16) {                                                                                                       \
        const u32x4* xp_ = reinterpret_cast<const u32x4*>(x + (t) * d);                                               \
        _Pragma("unroll") for (int k = 0; k < NPF; ++k) {                                                             \
            int q_ = tl_ + k * THREADS;                                                                               \
            q_ = q_ < pieces ? q_ : pieces - 1;                                                                       \
            asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(PF[k]) : "v"(xp_ + q_) : "memory");               \
        }                                                                                                             \
    } else if (pf8) {                                                                                                 \
        typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));                                                  \
        const u32x2_* xp_ = reinterpret_cast<const u32x2_*>(x + (t) * d);                                             \
        _Pragma("unroll") for (int k = 0; k < NPF; ++k) {                                                             \
            int q0_ = tl_ + (2 * k) * THREADS, q1_ = tl_ + (2 * k + 1) * THREADS;                                     \
            q0_ = q0_ < pieces ? q0_ : pieces - 1;                                                                    \
            q1_ = q1_ < pieces ? q1_ : pieces - 1;                                                                    \
            u32x2_ lo_, hi_;                                                                                          \
            asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(lo_) : "v"(xp_ + q0_) : "memory");               \
            asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(hi_) : "v"(xp_ + q1_) : "memory");               \
            PF[k] = u32x4{lo_[0], lo_[1], hi_[0], hi_[1]};                                                            \
        }                                                                                                             \
    }                                                                                                                 \
    }
    if (blockIdx.x < rows) FQ_GEN_PF((int64_t)blockIdx.x)

    for (int64_t tok = blockIdx.x; tok < rows; tok += gridDim.x) {
        __syncthreads();  // everyone is done with xs / obuf of the previous token
        if (pf16 || pf8) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < NPF; ++k) asm volatile("" : "+v"(PF[k]));  // arrived with the wait above
            int tl = tid;
            asm volatile("" : "+v"(tl));
            if (pf16) {
                const int cpr = N >> 3;
                const uint4* dp = reinterpret_cast<const uint4*>(diag);
#pragma unroll
                for (int k = 0; k < NPF; ++k) {
                    const int q = tl + k * THREADS;
                    if (q < pieces) {
                        uint4 v = __builtin_bit_cast(uint4, PF[k]);
                        if (diag != nullptr) v = __builtin_bit_cast(uint4, __builtin_bit_cast(X8, v) * __builtin_bit_cast(X8, dp[q]));
                        const int row = q / cpr, ch = q - row * cpr;
                        xs[row * pitch + ch] = v;
                    }
                }
            } else {
                const int cpr = N >> 2;
                const uint2* dp = reinterpret_cast<const uint2*>(diag);
                uint2* xs2 = reinterpret_cast<uint2*>(xs);
#pragma unroll
                for (int k = 0; k < 2 * NPF; ++k) {
                    const int q = tl + k * THREADS;
                    if (q < pieces) {
                        uint2 v = (k & 1) ? make_uint2(PF[k >> 1][2], PF[k >> 1][3]) : make_uint2(PF[k >> 1][0], PF[k >> 1][1]);
                        if (diag != nullptr) v = __builtin_bit_cast(uint2, __builtin_bit_cast(X4, v) * __builtin_bit_cast(X4, dp[q]));
                        const int row = q / cpr, ch = q - row * cpr;
                        xs2[row * pitch * 2 + ch] = v;
                    }
                }
            }
        } else if (!(N & 7)) {   // rows are whole 16-byte chunks
            const int cpr = N >> 3;
            const uint4* xp = reinterpret_cast<const uint4*>(x + tok * d);
            const uint4* dp = reinterpret_cast<const uint4*>(diag);
            for (int q = tid; q < M * cpr; q += THREADS) {
                uint4 v = xp[q];
                if (diag != nullptr) v = __builtin_bit_cast(uint4, __builtin_bit_cast(X8, v) * __builtin_bit_cast(X8, dp[q]));
                const int row = q / cpr, ch = q - row * cpr;
                xs[row * pitch + ch] = v;
            }
        } else if (!(N & 3)) {  // 8-byte pieces (N = 148: 37 per row)
            const int cpr = N >> 2;
            const uint2* xp = reinterpret_cast<const uint2*>(x + tok * d);
            const uint2* dp = reinterpret_cast<const uint2*>(diag);
            uint2* xs2 = reinterpret_cast<uint2*>(xs);
            for (int q = tid; q < M * cpr; q += THREADS) {
                uint2 v = xp[q];
                if (diag != nullptr) v = __builtin_bit_cast(uint2, __builtin_bit_cast(X4, v) * __builtin_bit_cast(X4, dp[q]));
                const int row = q / cpr, ch = q - row * cpr;
                xs2[row * pitch * 2 + ch] = v;
            }
        } else {  // 4-byte pieces (N even)
            const int cpr = N >> 1;
            const uint32_t* xp = reinterpret_cast<const uint32_t*>(x + tok * d);
            const uint32_t* dp = reinterpret_cast<const uint32_t*>(diag);
            uint32_t* xs1 = reinterpret_cast<uint32_t*>(xs);
            for (int q = tid; q < M * cpr; q += THREADS) {
                uint32_t v = xp[q];
                if (diag != nullptr) v = __builtin_bit_cast(uint32_t, __builtin_bit_cast(X2, v) * __builtin_bit_cast(X2, dp[q]));
                const int row = q / cpr, ch = q - row * cpr;
                xs1[row * pitch * 4 + ch] = v;
            }
        }
        __syncthreads();
        if (tok + gridDim.x < rows) FQ_GEN_PF(tok + gridDim.x)

        // ---- GEMM 1 for this wave's n'-tile over all row tiles, rounded to fp16: the A fragments of GEMM 2 ----
        X8 Uh[MT][2];
        if (nt < NT) {  // (wave-uniform: an MFMA takes operands from all 64 lanes, also those whose run lies beyond N)
            f32x16 U[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) U[mt] = f32x16{0};
            const uint4* rf = rfrag + (size_t)nt * KS1 * 64 + lane;
            // R fragments come from L2 (the image is up to 128 KB): four K-steps are fetched while the previous four are
            // multiplied, so the L2 latency is paid once per token and not once per K-step (CH K-steps per batch)
            constexpr int CH = MT >= 7 ? 1 : 4;   // (M > 192: the accumulators leave no room for more)
            X8 bc[CH], bn[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) bc[j] = __builtin_bit_cast(X8, rf[(j < KS1 ? j : KS1 - 1) * 64]);
            for (int s0 = 0; s0 < KS1; s0 += CH) {
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int sn = s0 + CH + j;
                    bn[j] = __builtin_bit_cast(X8, rf[(sn < KS1 ? sn : KS1 - 1) * 64]);
                }
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    if (s0 + j < KS1) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const X8 a = __builtin_bit_cast(X8, xs[(mt * 32 + c) * pitch + (s0 + j) * 2 + h]);
                            U[mt] = fq_mfma32<T>(a, bc[j], U[mt]);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < CH; ++j) bc[j] = bn[j];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int j = 0; j < 8; ++j) Uh[mt][p][j] = (T)U[mt][p * 8 + j];
        }
        // GEMM 2 for ONE output row tile: Y^T of tile (nt, mo), rows n' = n0 + r, col m' = 32 mo + c, with the post-scale and
        // the fp16 rounding applied. The kernel never holds more than one of them: the extrema take one sweep over the row
        // tiles, every output set another (GEMM 2 is recomputed — MT^2 MFMAs — instead of keeping 16 MT accumulators and
        // everything derived from them live through the output code: 8 waves fit 256 VGPRs for every MT).
        const int ks_n = (M + 15) >> 4;  // rows of L beyond M are zero
        const float ps = out.post_scale;
        auto row_tile = [&](int mo) -> f32x16 {
            const uint4* lf = (LLDS ? static_cast<const uint4*>(lds_l) : lfrag_g) + (size_t)mo * 64 + lane;
            // the tile's L fragments are fetched in batches of MT before their MFMAs: a fragment read issued right in front of
            // its MFMA put a full LDS / L2 latency in front of each of the 2 MT dependent MFMAs (same summation order as before)
            f32x16 Y = f32x16{0};
#pragma unroll
            for (int k0 = 0; k0 < 2 * MT; k0 += MT) {
                X8 Bf[MT];
#pragma unroll
                for (int j = 0; j < MT; ++j) Bf[j] = __builtin_bit_cast(X8, lf[(size_t)(k0 + j) * MT * 64]);
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    if (k0 + j < ks_n) Y = fq_mfma32<T>(Uh[(k0 + j) >> 1][(k0 + j) & 1], Bf[j], Y);
            }
            if (ps != 0.0f) {  // (fq_kron_quant_ex_f16)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = Y[r] * ps;
                    asm volatile("" : "+v"(p));  // an fp32 VALUE (no fusion with a later rounding)
                    Y[r] = p;
                }
            }
            if (flags & FQ_ROUND_Y_F16) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Y[r] = (float)(T)Y[r];
            }
            return Y;
        };

        // ---- per-token extrema over the VALID entries (padding rows / columns are excluded) ----
        float vmax = -INFINITY, vmin = INFINITY;
        if (nt < NT) {
#pragma unroll 1
            for (int mo = 0; mo < MT; ++mo) {
                const f32x16 Y = row_tile(mo);
                if ((mo * 32 + c) < M) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (r < nvalid) {
                            vmax = fmaxf(vmax, Y[r]);
                            vmin = fminf(vmin, Y[r]);
                        }
                    }
                }
            }
        }
        vmax = fq_wave_max(vmax);
        vmin = fq_wave_min(vmin);
        if (lane == 0) {
            red[wave] = vmax;
            red[WAVES + wave] = vmin;
        }
        __syncthreads();  // also: every wave has finished reading xs -> it may be reused as an output stage
        vmax = red[0];
        vmin = red[WAVES];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            vmax = fmaxf(vmax, red[w]);
            vmin = fminf(vmin, red[WAVES + w]);
        }

        // ---- fp16 outputs (transform / fake-quant) are staged dense [M][N] in xs, then streamed out ----
        T* stage = reinterpret_cast<T*>(smem);
        if (flags & FQ_OUT_TRANSFORM) {
            if (nt < NT) {
#pragma unroll 1
                for (int mo = 0; mo < MT; ++mo) {
                    const f32x16 Y = row_tile(mo);
                    if (nvalid > 0 && (mo * 32 + c) < M) {
                        X8 v0, v1;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            v0[e] = (T)Y[e];
                            v1[e] = (T)Y[8 + e];
                        }
                        gen_put_run16<T>(stage + (mo * 32 + c) * N + n0, v0, v1, nvalid);
                    }
                }
            }
            __syncthreads();
            gen_copy_out(reinterpret_cast<unsigned char*>(out.y + tok * d), smem, (int)(d * 2), tid, THREADS);
            __syncthreads();
        }

        for (int ci = 0; ci < out.n_clips; ++ci) {
            if (!(flags & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT))) break;
            float scale, sig_max, sig_min;
            fq_token_sigs(out, ci, tok, gcur, sig_max, sig_min);
            if (flags & FQ_QUANT_F16) scale = fq_token_scale<FQ_QUANT_F16, T>(vmax, vmin, sig_max, sig_min, flags);
            else scale = fq_token_scale<0, T>(vmax, vmin, sig_max, sig_min, flags);
            const float inv = fq_fast_inv(scale);
            const bool magic = !(flags & FQ_QUANT_F16) && fq_magic_ok(vmax, vmin, inv);
            if (nt < NT) {
#pragma unroll 1
                for (int mo = 0; mo < MT; ++mo) {
                    const f32x16 Y = row_tile(mo);
                    const bool ok = nvalid > 0 && (mo * 32 + c) < M;
                    float qv[16];
                    if (flags & FQ_QUANT_F16) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) qv[r] = (float)fq_quant1<FQ_QUANT_F16, T>(Y[r], scale);
                    } else {
                        float dmax = 1.0f;  // !magic: quotients too large for the fast rounding -> the true division
                        if (magic) {
                            dmax = 0.0f;
#pragma unroll
                            for (int r = 0; r < 16; ++r) qv[r] = fq_qfast(ok ? Y[r] : 0.0f, inv, dmax);
                        }
                        if (fq_wave_needs_exact(dmax)) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) qv[r] = fq_qexact(Y[r], scale);
                        }
                    }
                    if (ok && (flags & FQ_OUT_PACKED)) {
                        uint2 pk;
                        pk.x = fq_pack8(qv[0], qv[1], qv[2], qv[3], qv[4], qv[5], qv[6], qv[7]);
                        pk.y = fq_pack8(qv[8], qv[9], qv[10], qv[11], qv[12], qv[13], qv[14], qv[15]);
                        gen_put_run8(obuf + (mo * 32 + c) * (N >> 1) + (n0 >> 1), pk, nvalid >> 1);
                    }
                    if (ok && (flags & FQ_OUT_FAKEQUANT)) {
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
                        gen_put_run16<T>(stage + (mo * 32 + c) * N + n0, v0, v1, nvalid);
                    }
                }
            }
            __syncthreads();
            if (flags & FQ_OUT_PACKED) {
                if (tid == 0) reinterpret_cast<T*>(out.scale[ci])[tok] = (T)scale;
                gen_copy_out(out.q[ci] + tok * (d >> 1), obuf, (int)(d >> 1), tid, THREADS);
            }
            if (flags & FQ_OUT_FAKEQUANT)
                gen_copy_out(reinterpret_cast<unsigned char*>(out.fq[ci] + tok * d), smem, (int)(d * 2), tid, THREADS);
            __syncthreads();
        }

        // the fp16 output stage lives in xs: restore the zero padding the next token relies on
        if (flags & (FQ_OUT_TRANSFORM | FQ_OUT_FAKEQUANT)) {
            for (int i = tid; i < xs_chunks; i += THREADS) xs[i] = make_uint4(0, 0, 0, 0);
        }
    }
}

#undef FQ_GEN_PF

template <int MT, int WAVES, bool LLDS, typename T>
int launch_general_l(int flags, const T* x, const uint4* ws, const T* diag, int64_t rows, const GenGeom& g,
                   const FqQuantOut& out, int n_cu, hipStream_t stream) {
    const size_t xs_bytes = (size_t)MT * 32 * g.pitch * 16;
    const size_t ob = ((size_t)g.M * g.N / 2 + 15) & ~(size_t)15;
    const size_t lds = xs_bytes + ob + 2 * WAVES * sizeof(float) + 48 + (LLDS ? (size_t)2 * MT * MT * 1024 : 0);
    if (lds > 160 * 1024) return -1000;
    auto kern = fq_kron_general_kernel<MT, WAVES, LLDS, T>;
    FQ_RAISE_LDS_CAP(kern, 160 * 1024);
    int per_cu = (int)((160 * 1024) / lds);  // workgroups that fit a CU's LDS: they overlap each other's synchronous token load
    const int cap = WAVES == 4 ? 4 : 2;
    if (per_cu > cap) per_cu = cap;
    if (per_cu < 1) per_cu = 1;
    int64_t blocks = (int64_t)n_cu * per_cu;
    if (blocks > rows) blocks = rows;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WAVES * 64), lds, stream, x, ws, diag, rows, g, out, flags);
    return (int)hipGetLastError();
}

template <int MT, int WAVES, typename T>
int launch_general(int flags, const T* x, const uint4* ws, const T* diag, int64_t rows, const GenGeom& g,
                   const FqQuantOut& out, int n_cu, hipStream_t stream) {
    const int rc = launch_general_l<MT, WAVES, true, T>(flags, x, ws, diag, rows, g, out, n_cu, stream);  // L image in LDS if it fits
    return rc != -1000 ? rc : launch_general_l<MT, WAVES, false, T>(flags, x, ws, diag, rows, g, out, n_cu, stream);
}

}  // namespace

// Every pair with M <= 256, even N <= 256 and M * N <= 32768. -1000 outside that range.
// ws: fragment workspace already filled by fq_kron_prepare_kernel (fq_kron_generic.hip).
int fq_launch_kron_general(int flags, const f16* x, const void* ws, const f16* diag, int64_t rows, int M, int N,
                           const FqQuantOut& out, int n_cu, hipStream_t stream) {
    if (M < 1 || N < 2 || (N & 1) || M > 256 || N > 256 || (int64_t)M * N > 32768) return -1000;
    if (out.rt_flags & FQ_GROUP128) return -1000;
    GenGeom g;
    g.M = M;
    g.N = N;
    g.KS1 = (N + 15) / 16;
    g.NT = (N + 31) / 32;
    g.pitch = (g.KS1 * 2) | 1;
    const int MT = (M + 31) / 32;
    const uint4* w = reinterpret_cast<const uint4*>(ws);
    if (flags & FQ_DT_BF16) {
        flags &= ~FQ_DT_BF16;
        const bf16* xb = (const bf16*)x;
        const bf16* db = (const bf16*)diag;
#define FQ_GNB(MT_)                                                                                              \
    if (MT == MT_)                                                                                               \
        return g.NT <= 4 ? launch_general<MT_, 4, bf16>(flags, xb, w, db, rows, g, out, n_cu, stream)            \
                         : launch_general<MT_, 8, bf16>(flags, xb, w, db, rows, g, out, n_cu, stream);
        FQ_GNB(1) FQ_GNB(2) FQ_GNB(3) FQ_GNB(4) FQ_GNB(5) FQ_GNB(6) FQ_GNB(7) FQ_GNB(8)
#undef FQ_GNB
        return -1000;
    }
#define FQ_GN(MT_)                                                                                        \
    if (MT == MT_)                                                                                        \
        return g.NT <= 4 ? launch_general<MT_, 4, f16>(flags, x, w, diag, rows, g, out, n_cu, stream)     \
                         : launch_general<MT_, 8, f16>(flags, x, w, diag, rows, g, out, n_cu, stream);
    FQ_GN(1) FQ_GN(2) FQ_GN(3) FQ_GN(4) FQ_GN(5) FQ_GN(6) FQ_GN(7) FQ_GN(8)
#undef FQ_GN
    return -1000;
}
