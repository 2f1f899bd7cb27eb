// fq_kron_tall.hip — fused Kronecker transform + per-token INT4 quantisation for TALL tokens: 64 < M <= 192, N = 64
// (packed output). The pairs behind it are online Hadamard rotations run as ONE Kronecker launch in front of the Quantizer
// (hadamard_utils.py:132-141 + deploy/nn/quantization.py:13-36; flatquant_amd/ops.py::_hadamard_as_kron):
//   11008 = 172 x 64 (Llama-2-7B ffn: had172 (x) H64), 8960 = 140 x 64, 9984 = 156 x 64, 6912 = 108 x 64, 7168 = 112 x 64 ...
// and any calibrated pair of that shape through fq_kron_quant_f16.
//
// Why its own kernel. With N = 64 a token has only two 32-column n'-tiles: the kernels that split a token's work by n'-tile
// (fq_kron_generic.hip, fq_kron_trio.hip) keep two waves busy, and the general kernel took 836 us per 16384 tokens of
// 172 x 64 (round 3). Here the token is split by ROW tile instead — a workgroup of MT = ceil(M / 32) waves per token:
//   * GEMM 1 (U = X.R): wave w multiplies ITS 32 rows of the token by R. Nobody else needs those rows: the A fragments come
//     straight from HBM into registers (one token ahead), the token is never staged in LDS; R (8 fragments) lives in registers;
//   * GEMM 2 (Y^T = U^T.L) contracts over ALL rows: every wave rounds its slice of U to fp16 and publishes it in LDS in
//     fragment order (4 KB per wave, double-buffered), one s_barrier, then wave w computes the output columns m' of ITS row
//     tile from everybody's slices; the 2 MT fragments of L it needs never change: registers;
//   * extrema: wave reduction, MT partials through LDS, second s_barrier; quantiser in registers; a lane's two n'-tiles of a
//     row are neighbours: one 16-byte store, and the 64 lanes of a store cover 1 KB of contiguous output.
// Two or three such workgroups per CU overlap each other's barriers. Same mathematics, rounding points, fragment chaining and
// workspace image (fq_kron_prepare_kernel) as the other Kronecker kernels.
// (round 4) this kernel keeps the two-sided quantiser of round 3: it runs at 2.26-2.29 GHz, VALU-issue-bound rather than power-bound,
// and the low-half form's v_min3_u16 issues at half rate (tools/microbench/vrate.hip) — 164 -> 178 us with it (profiles/r04_quant_lo_ab.txt)
#define FQ_QUANT_LO 0
#include "fq_common.hpp"

namespace {

#ifndef TALL_ASM_LOADS
#define TALL_ASM_LOADS 1
#endif
#ifndef TALL_ABL
#define TALL_ABL 0   // measurement builds: 1 = the first token's rows are reused (no loads after the first)
#endif
constexpr int TALL_N = 64, TALL_NT = 2, TALL_KS1 = 4;

// Workgroup barrier for LDS traffic only. __syncthreads() also drains vmcnt: the next token's rows — requested one token ahead
// on purpose — would be waited for at every barrier (measured: 5 us per token and workgroup, the HBM latency, whatever M).
#define FQ_TALL_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int MT, bool H16, bool YOUT>
__global__ __launch_bounds__(MT * 64) void fq_kron_tall_kernel(const f16* __restrict__ x, const uint4* __restrict__ ws,
                                                               int64_t rows, int M, FqQuantOut out) {
    constexpr int N = TALL_N, NT = TALL_NT, KS1 = TALL_KS1;
    // Uh image of one token: [MT waves][NT][2 halves][64 lanes] uint4, double-buffered; then the extrema [2][2][8] floats
    __shared__ __attribute__((aligned(16))) uint4 uimg[2][MT * NT * 2 * 64];
    __shared__ float red[2][2][8];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // this wave's row tile (GEMM 1) and output column tile (GEMM 2)
    const int64_t d = (int64_t)M * N;
    const int ks_n = (M + 15) >> 4;   // K-steps of GEMM 2 that hold rows of the token (rows of L beyond M are zero)

    // ---- constants of the launch: R (all of it) and this wave's column of L, as MFMA B fragments in registers ----
    f16x8 RF[NT][KS1], LF[2 * MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < KS1; ++s) RF[nt][s] = __builtin_bit_cast(f16x8, ws[(nt * KS1 + s) * 64 + lane]);
    {
        const uint4* lsrc = ws + NT * KS1 * 64;
#pragma unroll
        for (int ks = 0; ks < 2 * MT; ++ks) LF[ks] = __builtin_bit_cast(f16x8, lsrc[(ks * MT + w) * 64 + lane]);
    }
    // (round 5) The fragments are PINNED here: an asm statement that names a register makes the compiler wait for its load in front of
    // the statement. Without it the scheduler sank the fragment loads below the prologue's vmcnt(0) and waited for them lazily, at
    // their first uses INSIDE the token loop — s_waitcnt vmcnt(19) ... vmcnt(0) in front of GEMM 2's MFMAs, executed every iteration:
    // no-ops for the fragments after the first token, but the counter is shared with the untracked row loads issued just above them,
    // so every token waited for the NEXT token's rows in the middle of GEMM 2:
    // the builds of rounds 3 and 4 did that. Measured (round 5, profiles/r05_tall_prefetch.txt): 172 x 64 167 -> 162.5 us, 140 x 64 166 -> 155,
    // 96 x 64 83 -> 77 — and NOT more: a build that never loads after the first token runs in the same 163 us, and two deeper-prefetch
    // builds (a counted wait behind the stores; two register sets with vmcnt(4)) measured 165 / 170 us and were removed. What bounds this
    // kernel is its two workgroup barriers per token over 6 waves on 4 SIMDs, not the loads.
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < KS1; ++s) asm volatile("" : "+v"(RF[nt][s]));
#pragma unroll
    for (int ks = 0; ks < 2 * MT; ++ks) asm volatile("" : "+v"(LF[ks]));
    // this lane's row of the token: row 32 w + c, its K-half h of every K-step: 16 bytes at chunk 2 s + h of a 128-byte row
    const bool row_ok = (w * 32 + c) < M;
#if TALL_ASM_LOADS
    // (round 4) The next token's rows are requested by loads the compiler does not track, and waited for by an explicit vmcnt(0) IN
    // FRONT OF the token's stores. With compiler-tracked loads the wait sat at the top of the loop, behind the stores of the previous
    // token (vmcnt counts loads and stores in one queue, and the stores are conditional, so the compiler has to wait for 0): every
    // token then waited for its predecessor's stores to be acknowledged — 1-2 us of a 6.6 us token period at 2.4 GHz and 1095 W.
    // Rows below the token read row M - 1 instead of zeros (no select behind the load, which would copy a register the load has not
    // written yet): a padding row of U holds finite copies and meets zero rows of L in GEMM 2.
    const int row_ld = row_ok ? w * 32 + c : M - 1;
    const int64_t lane_off = (int64_t)row_ld * N + h * 8;   // in elements
    auto fetch = [&](int64_t tok, f16x8 (&A)[KS1]) {
        const f16* p = x + tok * d + lane_off;
        asm volatile("global_load_dwordx4 %0, %4, off\n\t"
                     "global_load_dwordx4 %1, %4, off offset:32\n\t"
                     "global_load_dwordx4 %2, %4, off offset:64\n\t"
                     "global_load_dwordx4 %3, %4, off offset:96"
                     : "=&v"(A[0]), "=&v"(A[1]), "=&v"(A[2]), "=&v"(A[3]) : "v"(p) : "memory");
    };
    static_assert(KS1 == 4, "fetch issues four loads");
// (the builtin, not inline asm: the compiler's own wait insertion sees it — it then knows that the fragment loads of the prologue
//  are complete as well, and places no vmcnt wait of its own inside the loop; 0x0F70 = vmcnt(0), expcnt and lgkmcnt unconstrained)
#define FQ_TALL_WAIT_ROWS() { asm volatile("" ::: "memory"); __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" ::: "memory"); }
#else
    const int64_t lane_off = (int64_t)(w * 32 + c) * N + h * 8;   // in elements
    auto fetch = [&](int64_t tok, f16x8 (&A)[KS1]) {
        const f16* p = x + tok * d + lane_off;
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            if (row_ok) A[s] = *reinterpret_cast<const f16x8*>(p + s * 16);
            else A[s] = f16x8{0};
        }
    };
#define FQ_TALL_WAIT_ROWS()
#endif
    const float ps = out.post_scale != 0.0f ? out.post_scale : 1.0f;
    FqGroupCursor gcur;

    // ---- one token: `A` holds its rows (landed); `nxt` is the token whose rows are requested into `A` once GEMM 1 has read it ----
    auto token = [&](const int64_t tok, const int it, f16x8(&A)[KS1], const int64_t nxt) __attribute__((always_inline)) {
        const int buf = it & 1;
        // ================= GEMM 1: this wave's 32 rows against R, both n'-tiles =================
        f32x16 U[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) U[nt] = f32x16{0};
#pragma unroll
        for (int s = 0; s < KS1; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) U[nt] = fq_mfma32<f16>(A[s], RF[nt][s], U[nt]);
        {
            // the next rows (nxt == tok: the last token re-reads itself — no branch round the untracked loads): in flight under
            // everything below. (round 3, with compiler-tracked loads: two tokens ahead measured no faster, 140 x 64 206 vs 202 us —
            // every wait was a vmcnt(0) then, whatever the distance.)
#if TALL_ASM_LOADS
            fetch(nxt, A);
#else
            if (nxt != tok) fetch(nxt, A);
#endif
        }
        // U rounded to fp16: the C fragment of GEMM 1 is the A fragment of GEMM 2 (K-steps 2 w and 2 w + 1) — published
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                f16x8 uh;
#pragma unroll
                for (int j = 0; j < 8; ++j) uh[j] = (f16)U[nt][p * 8 + j];
                uimg[buf][((w * NT + nt) * 2 + p) * 64 + lane] = __builtin_bit_cast(uint4, uh);
            }
        FQ_TALL_LDS_BARRIER();
        // ================= GEMM 2: output columns m' = 32 w + c from every wave's slice of U =================
        f32x16 Y[NT];   // Y^T of tile (nt, mo = w): rows n' = h*32 + nt*16 + r, col m' = 32 w + c
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Y[nt] = f32x16{0};
        {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const uint4* ub = uimg[buf] + ln;
            f16x8 UA[2][NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) UA[0][nt] = __builtin_bit_cast(f16x8, ub[(nt * 2) * 64]);
#pragma unroll
            for (int ks = 0; ks < 2 * MT; ++ks) {
                if (ks + 1 < 2 * MT) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        UA[(ks + 1) & 1][nt] = __builtin_bit_cast(f16x8, ub[((((ks + 1) >> 1) * NT + nt) * 2 + ((ks + 1) & 1)) * 64]);
                }
                if (ks < 2 * MT - 2 || ks < ks_n) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) Y[nt] = fq_mfma32<f16>(UA[ks & 1][nt], LF[ks], Y[nt]);
                }
            }
        }
        // ================= extrema of this wave's slice (rows of the output beyond M are padding) =================
        uint32_t H[H16 ? NT : 1][8];   // H16: the fp16 pairs the deploy Quantizer sees
        float vmax = -INFINITY, vmin = INFINITY;
        if (H16) {
            f16x2 pmax = {(f16)-INFINITY, (f16)-INFINITY}, pmin = {(f16)INFINITY, (f16)INFINITY};
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f16x2 pr = fq_mul_to_f16x2(Y[nt][2 * j], Y[nt][2 * j + 1], f32x2{ps, ps});
                    H[H16 ? nt : 0][j] = __builtin_bit_cast(uint32_t, pr);
                    pmax = fq_pk_max(pmax, pr);
                    pmin = fq_pk_min(pmin, pr);
                }
            if (row_ok) {
                vmax = fmaxf((float)pmax[0], (float)pmax[1]);
                vmin = fminf((float)pmin[0], (float)pmin[1]);
            }
        } else {
            if (out.post_scale != 0.0f) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float p = Y[nt][r] * ps;
                        asm volatile("" : "+v"(p));   // an fp32 VALUE (no fusion with a later rounding)
                        Y[nt][r] = p;
                    }
            }
            if (out.rt_flags & FQ_ROUND_Y_F16) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Y[nt][r] = (float)(f16)Y[nt][r];
            }
            float a = -INFINITY, b = INFINITY;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    a = fq_max3(a, Y[nt][r], Y[nt][r + 1]);
                    b = fq_min3(b, Y[nt][r], Y[nt][r + 1]);
                }
            if (row_ok) {
                vmax = a;
                vmin = b;
            }
        }
        FQ_TALL_WAIT_ROWS();   // the next token's rows have had GEMM 2 and the extrema to arrive; nothing waits for the stores below
        if (YOUT && row_ok) {   // the transformed activation as well (the launch the parity tests read; kronecker_matmul + quant)
            f16* yrow = out.y + tok * d + (int64_t)(w * 32 + c) * N + h * 32;   // n' = h*32 + nt*16 + r: 16 consecutive values per tile
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                uint32_t v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (H16) v[j] = H[H16 ? nt : 0][j];
                    else v[j] = __builtin_bit_cast(uint32_t, f16x2{(f16)Y[nt][2 * j], (f16)Y[nt][2 * j + 1]});
                }
                u32x4* dst = reinterpret_cast<u32x4*>(yrow + nt * 16);
                dst[0] = u32x4{v[0], v[1], v[2], v[3]};
                dst[1] = u32x4{v[4], v[5], v[6], v[7]};
            }
        }
        vmax = fq_wave_max(vmax);
        vmin = fq_wave_min(vmin);
        if (lane == 0) {
            red[buf][0][w] = vmax;
            red[buf][1][w] = vmin;
        }
        FQ_TALL_LDS_BARRIER();
        vmax = red[buf][0][0];
        vmin = red[buf][1][0];
#pragma unroll
        for (int i = 1; i < MT; ++i) {
            vmax = fmaxf(vmax, red[buf][0][i]);
            vmin = fminf(vmin, red[buf][1][i]);
        }
        // ================= scale, quantiser, pack, store =================
        for (int ci = 0; ci < out.n_clips; ++ci) {
            float sig_max, sig_min;
            fq_token_sigs(out, ci, tok, gcur, sig_max, sig_min);
            float scale;
            if (H16) scale = fq_token_scale<FQ_QUANT_F16>(vmax, vmin, sig_max, sig_min, out.rt_flags);
            else scale = fq_token_scale<0>(vmax, vmin, sig_max, sig_min, out.rt_flags);
            const float inv = fq_fast_inv(scale);
            const FqH16Recip rc = H16 ? fq_h16_recip(scale) : FqH16Recip{0.0f, 0.0f};
            const bool magic = H16 || fq_magic_ok(vmax, vmin, inv);
            const bool clampq = fq_needs_clamp(vmax, vmin, inv);
            uint2 pk[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (H16) {
                    const uint32_t(&hv)[8] = H[H16 ? nt : 0];
                    if (clampq) {
                        pk[nt].x = fq_quant8_h16<true>(hv[0], hv[1], hv[2], hv[3], rc);
                        pk[nt].y = fq_quant8_h16<true>(hv[4], hv[5], hv[6], hv[7], rc);
                    } else {
                        pk[nt].x = fq_quant8_h16<false>(hv[0], hv[1], hv[2], hv[3], rc);
                        pk[nt].y = fq_quant8_h16<false>(hv[4], hv[5], hv[6], hv[7], rc);
                    }
                } else {
                    const f32x16& yv = Y[nt];
                    unsigned long long d0m = ~0ull, d1m = ~0ull;
                    pk[nt] = uint2{0u, 0u};
                    if (magic) {
                        const float ilo = fq_inv_lo(inv), ihi = fq_inv_hi(inv);
                        if (clampq) {
                            pk[nt].x = fq_quant8<true>(yv[0], yv[1], yv[2], yv[3], yv[4], yv[5], yv[6], yv[7], inv, ilo, ihi, d0m);
                            pk[nt].y = fq_quant8<true>(yv[8], yv[9], yv[10], yv[11], yv[12], yv[13], yv[14], yv[15], inv, ilo, ihi, d1m);
                        } else {
                            pk[nt].x = fq_quant8<false>(yv[0], yv[1], yv[2], yv[3], yv[4], yv[5], yv[6], yv[7], inv, ilo, ihi, d0m);
                            pk[nt].y = fq_quant8<false>(yv[8], yv[9], yv[10], yv[11], yv[12], yv[13], yv[14], yv[15], inv, ilo, ihi, d1m);
                        }
                    }
                    if (d0m)   // rare: an ambiguous digit somewhere in the wave -> the true division for this dword
                        pk[nt].x = fq_pack8(fq_qexact(yv[0], scale), fq_qexact(yv[1], scale), fq_qexact(yv[2], scale), fq_qexact(yv[3], scale),
                                            fq_qexact(yv[4], scale), fq_qexact(yv[5], scale), fq_qexact(yv[6], scale), fq_qexact(yv[7], scale));
                    if (d1m)
                        pk[nt].y = fq_pack8(fq_qexact(yv[8], scale), fq_qexact(yv[9], scale), fq_qexact(yv[10], scale), fq_qexact(yv[11], scale),
                                            fq_qexact(yv[12], scale), fq_qexact(yv[13], scale), fq_qexact(yv[14], scale), fq_qexact(yv[15], scale));
                }
            }
            // row m' = 32 w + c holds 32 bytes: n' = h*32 + nt*16 + r -> byte h*16 + nt*8 + r/2: a lane's two tiles are 16 contiguous
            // bytes, the wave's 64 lanes 1 KB of contiguous output
            if (row_ok)
                *reinterpret_cast<u32x4*>(out.q[ci] + tok * (d >> 1) + (w * 32 + c) * (N / 2) + h * 16) =
                    u32x4{pk[0].x, pk[0].y, pk[1].x, pk[1].y};
            // (FQ_RATIO_POST — deploy.nn.Quantizer(lac=False) behind a Hadamard rotation, fq_kron_quant_ex_f16: no zero guard, an all-zero
            //  token stores scale 0 as fq_rowquant_f16 does, deploy/nn/quantization.py:30)
            if (w == 0 && lane == 0)
                out.scale[ci][tok] = ((out.rt_flags & FQ_RATIO_POST) && vmax == 0.0f && vmin == 0.0f) ? (f16)0.0f : (f16)scale;
        }
    };

    int64_t tok = blockIdx.x;
    const int64_t step = gridDim.x;
    f16x8 A[KS1];
#if TALL_ASM_LOADS
    fetch(tok, A);   // (blockIdx.x < rows: the launcher starts at most one workgroup per token. Unconditional on purpose — a branch
                     //  round the untracked loads makes A a phi, and a phi is a register COPY that may run before the data has arrived)
#else
    if (tok < rows) fetch(tok, A);
#endif
    FQ_TALL_WAIT_ROWS();
    for (int it = 0; tok < rows; tok += step, ++it) {
        const int64_t n1 = tok + step;
        token(tok, it, A, (n1 < rows && !(TALL_ABL & 1)) ? n1 : tok);
    }
}

template <int MT, bool H16, bool YOUT>
int launch_tall(const f16* x, const uint4* ws, int64_t rows, int M, const FqQuantOut& out, int n_cu, hipStream_t stream) {
    constexpr int WG_PER_CU = MT >= 5 ? 2 : MT == 4 ? 3 : 4;   // 12 waves per CU (168 VGPRs each)
    int64_t blocks = rows < (int64_t)n_cu * WG_PER_CU ? rows : (int64_t)n_cu * WG_PER_CU;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((fq_kron_tall_kernel<MT, H16, YOUT>), dim3((unsigned)blocks), dim3(MT * 64), 0, stream, x, ws, rows, M, out);
    return (int)hipGetLastError();
}

}  // namespace

// Returns -1000 when the shape / output set is not one this kernel covers (the caller goes on to the generic kernels).
// ws: fragment workspace already filled by fq_kron_prepare_kernel (rfrag [2][4][64], lfrag [2MT][MT][64]).
int fq_launch_kron_tall(int flags, const f16* x, const void* ws, const f16* diag, int64_t rows, int M, int N,
                        const FqQuantOut& out, int n_cu, hipStream_t stream) {
    if (N != TALL_N || M <= 64 || M > 192 || diag != nullptr) return -1000;
    if ((out.rt_flags & FQ_GROUP128) || out.ws_group_stride != 0) return -1000;
    const int ct = flags & FQ_CT_MASK, cq = ct & ~FQ_OUT_TRANSFORM;
    const bool yout = (ct & FQ_OUT_TRANSFORM) != 0;
    // (round 4) the transform ALONE (kronecker_matmul; the rotation of matmul_hadU_cuda run as a Kronecker pair): the fp16-epilogue
    // instantiation with no clip set — the quantiser loop does not run, the rounded pairs go out as they are
    const bool yonly = yout && (cq & ~FQ_QUANT_F16) == 0 && out.n_clips == 0;
    const bool h16 = yonly || (cq == (FQ_OUT_PACKED | FQ_QUANT_F16) && (flags & FQ_ROUND_Y_F16));
    if (cq != FQ_OUT_PACKED && !h16) return -1000;
    if (yout && out.y == nullptr) return -1000;
    const uint4* wp = reinterpret_cast<const uint4*>(ws);
    const int MT = (M + 31) / 32;
#define FQ_TALL(MT_)                                                                                              \
    if (MT == MT_) {                                                                                              \
        if (yout)                                                                                                 \
            return h16 ? launch_tall<MT_, true, true>(x, wp, rows, M, out, n_cu, stream)                          \
                       : launch_tall<MT_, false, true>(x, wp, rows, M, out, n_cu, stream);                        \
        return h16 ? launch_tall<MT_, true, false>(x, wp, rows, M, out, n_cu, stream)                             \
                   : launch_tall<MT_, false, false>(x, wp, rows, M, out, n_cu, stream);                           \
    }
    FQ_TALL(3) FQ_TALL(4) FQ_TALL(5) FQ_TALL(6)
#undef FQ_TALL
    return -1000;
}
