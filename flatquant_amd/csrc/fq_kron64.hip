// fq_kron64.hip — fused Kronecker transform + per-token INT4 quantisation for d = 4096 (M = N = 64).
//
// Replaces, for the Llama-3-8B hidden size, deploy/kernels/kron_matmul.py:24-130 (Triton matmul_kernel,
// non-split branch) and the flat_utils.py:6-17 + quant_utils.py:71-119 op sequence.
//
// Design (gfx950, wave64):
//   * ONE WAVE OWNS ONE TOKEN. A token is a 64x64 fp16 matrix X (8 KB). Nothing about a token ever
//     touches LDS: X is loaded from HBM straight into MFMA A-operand fragments (64 contiguous bytes per
//     lane), both small GEMMs run on v_mfma_f32_32x32x16_f16, and the fp32 result is quantised and
//     packed in registers, then stored with 16-byte coalesced stores.
//   * GEMM 1:  U = X . R          (contraction over n, the contiguous axis of X)      32x32x16, K = n
//     GEMM 2:  Y^T = U^T . L      (contraction over m)                               32x32x16, K = m
//     The C/D fragment of GEMM 1 (lane holds one column n', 16 rows m) is, after fp16 conversion,
//     EXACTLY an A fragment of GEMM 2 (row n', 8 k-slots m per K=16 step) because the order of the
//     contraction index inside an MFMA is free as long as A and B agree. So the intermediate never
//     leaves registers and needs no cross-lane traffic. The fp16 rounding of U is the rounding
//     flat_utils.py:15 performs (torch.matmul output dtype) — path-A arithmetic.
//   * The permutations are absorbed into the one-time gather of the L and R fragments (LDS-staged,
//     once per workgroup; the grid is persistent):
//       - columns of R are permuted so that, after GEMM 2, lane (h, c) holds for output row m' = c
//         the 32 CONSECUTIVE columns n' = 32h .. 32h+31 -> one 16-byte store of packed nibbles.
//   * Prefetch: the next token's 8 KB are loaded into a second register set before the current token's
//     MFMAs start; 8 waves/CU x 8 KB in flight x 2 = 128 KB per CU of outstanding HBM reads.
//
// Algorithmic traffic per token: 8192 B read + 2048 B packed + 2 B scale = 10242 B.
#include "fq_common.hpp"

namespace {

constexpr int KM = 64, KN = 64, KD = KM * KN;

__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// physical column n' held at (n-tile nt, tile position pos) of GEMM-1's output / GEMM-2's output rows.
__device__ __forceinline__ int nperm(int nt, int pos) {
    return ((pos >> 2) & 1) * 32 + nt * 16 + (pos & 3) + 4 * (pos >> 3);
}

template <int FLAGS>
__global__ __launch_bounds__(256, 2) void fq_kron64_kernel(const f16* __restrict__ x,
                                                           const f16* __restrict__ left,
                                                           const f16* __restrict__ right,
                                                           const f16* __restrict__ diag,
                                                           int64_t rows, FqQuantOut out) {
    __shared__ __attribute__((aligned(16))) f16 smem[2 * KD];  // [0,KD) = right, [KD,2KD) = left
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int h = lane >> 5;
    const int c = lane & 31;

    // ---- stage the two 64x64 matrices in LDS (coalesced 16-byte loads), gather fragments once ----
    {
        const uint4* gr = reinterpret_cast<const uint4*>(right);
        const uint4* gl = reinterpret_cast<const uint4*>(left);
        uint4* s4 = reinterpret_cast<uint4*>(smem);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            s4[tid + i * 256] = gr[tid + i * 256];
            s4[512 + tid + i * 256] = gl[tid + i * 256];
        }
    }
    __syncthreads();

    f16x8 Rf[2][4];   // [nt][s]   B operand of GEMM 1: R[n = 32h + 8s + j][n' = nperm(nt, c)]
    f16x8 Lf[4][2];   // [ks][mt'] B operand of GEMM 2: L[m(ks, h, j)][m' = 32 mt' + c]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int np = nperm(nt, c);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) Rf[nt][s][j] = smem[(h * 32 + s * 8 + j) * KN + np];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int mt = ks >> 1, p = ks & 1;
#pragma unroll
        for (int mo = 0; mo < 2; ++mo)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = mt * 32 + 16 * p + 8 * (j >> 2) + 4 * h + (j & 3);
                Lf[ks][mo][j] = smem[KD + m * KM + mo * 32 + c];
            }
    }

    const int wave = tid >> 6;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave;
    const int64_t n_waves = (int64_t)gridDim.x * 4;

    // lane's 64 contiguous bytes inside each 32-row half of X: row (mt*32 + c), columns 32h .. 32h+31
    const int lane_off = c * KN + h * 32;

    uint4 Xn[2][4];
    int64_t tok = wave_id;
    if (tok < rows) {
        const uint4* xp = reinterpret_cast<const uint4*>(x + tok * KD + lane_off);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int s = 0; s < 4; ++s) Xn[mt][s] = xp[mt * (32 * KN / 8) + s];
    }

    for (; tok < rows; tok += n_waves) {
        f16x8 Xc[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int s = 0; s < 4; ++s) Xc[mt][s] = __builtin_bit_cast(f16x8, Xn[mt][s]);

        const int64_t nxt = tok + n_waves;
        if (nxt < rows) {
            const uint4* xp = reinterpret_cast<const uint4*>(x + nxt * KD + lane_off);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int s = 0; s < 4; ++s) Xn[mt][s] = xp[mt * (32 * KN / 8) + s];
        }

        if (diag != nullptr) {  // x * diag_scale, rounded to fp16 (trans_utils.py:86-90)
            const uint4* dp = reinterpret_cast<const uint4*>(diag + lane_off);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    f16x8 dv = __builtin_bit_cast(f16x8, dp[mt * (32 * KN / 8) + s]);
                    Xc[mt][s] = Xc[mt][s] * dv;
                }
        }

        f32x16 Y[2][2];  // [nt][mt']: Y^T[n' = 32h + 16nt + r][m' = 32mt' + c]
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f32x16 U[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x16 acc = {0};
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = mfma32(Xc[mt][s], Rf[nt][s], acc);
                U[mt] = acc;
            }
            f16x8 Uh[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) Uh[ks][j] = (f16)U[ks >> 1][(ks & 1) * 8 + j];
#pragma unroll
            for (int mo = 0; mo < 2; ++mo) {
                f32x16 acc = {0};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = mfma32(Uh[ks], Lf[ks][mo], acc);
                Y[nt][mo] = acc;
            }
        }

        if (out.rt_flags & FQ_ROUND_Y_F16) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mo = 0; mo < 2; ++mo)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Y[nt][mo][r] = (float)(f16)Y[nt][mo][r];
        }

        if (FLAGS & FQ_OUT_TRANSFORM) {
#pragma unroll
            for (int mo = 0; mo < 2; ++mo) {
                uint4* yp = reinterpret_cast<uint4*>(out.y + tok * KD + (mo * 32 + c) * KN + h * 32);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        f16x8 v;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (f16)Y[nt][mo][w * 8 + e];
                        yp[nt * 2 + w] = __builtin_bit_cast(uint4, v);
                    }
            }
        }

        if (FLAGS & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT)) {
            float vmax = Y[0][0][0], vmin = Y[0][0][0];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mo = 0; mo < 2; ++mo)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        vmax = fmaxf(vmax, Y[nt][mo][r]);
                        vmin = fminf(vmin, Y[nt][mo][r]);
                    }
            vmax = fq_wave_max(vmax);
            vmin = fq_wave_min(vmin);

            for (int ci = 0; ci < out.n_clips; ++ci) {
                const float scale = fq_token_scale<FLAGS>(vmax, vmin, out.sig_max[ci], out.sig_min[ci], out.rt_flags);
                if (FLAGS & FQ_OUT_PACKED) {
                    if (lane == 0) out.scale[ci][tok] = (f16)scale;
#pragma unroll
                    for (int mo = 0; mo < 2; ++mo) {
                        uint4 pk;
                        uint32_t* pw = reinterpret_cast<uint32_t*>(&pk);
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            uint32_t d = 0;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const int q = fq_quant1<FLAGS>(Y[w >> 1][mo][(w & 1) * 8 + e], scale);
                                d |= (uint32_t)(q & 15) << (4 * e);
                            }
                            pw[w] = d;
                        }
                        *reinterpret_cast<uint4*>(out.q[ci] + tok * (KD / 2) + (mo * 32 + c) * (KN / 2) +
                                                  h * 16) = pk;
                    }
                }
                if (FLAGS & FQ_OUT_FAKEQUANT) {
#pragma unroll
                    for (int mo = 0; mo < 2; ++mo) {
                        uint4* fp =
                            reinterpret_cast<uint4*>(out.fq[ci] + tok * KD + (mo * 32 + c) * KN + h * 32);
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int w = 0; w < 2; ++w) {
                                f16x8 v;
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    const int q = fq_quant1<FLAGS>(Y[nt][mo][w * 8 + e], scale);
                                    v[e] = fq_dequant1<FLAGS>(q, scale);
                                }
                                fp[nt * 2 + w] = __builtin_bit_cast(uint4, v);
                            }
                    }
                }
            }
        }
    }
}

}  // namespace

// Host-side launcher used by the C ABI (fq_capi.hip). Returns hipError_t as int.
template <int FLAGS>
static int launch_kron64(const f16* x, const f16* left, const f16* right, const f16* diag, int64_t rows,
                         const FqQuantOut& out, int n_cu, hipStream_t stream) {
    int64_t blocks = (rows + 3) / 4;
    const int64_t cap = (int64_t)n_cu * 2;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fq_kron64_kernel<FLAGS>, dim3((unsigned)blocks), dim3(256), 0, stream, x, left,
                       right, diag, rows, out);
    return (int)hipGetLastError();
}

int fq_launch_kron64(int flags, const f16* x, const f16* left, const f16* right, const f16* diag,
                     int64_t rows, const FqQuantOut& out, int n_cu, hipStream_t stream) {
    // Compile-time specialisations: output set x fp16-quant arithmetic. Everything else is run-time.
#define FQ_CASE(F)                                                                     \
    case (F):                                                                          \
        return launch_kron64<(F)>(x, left, right, diag, rows, out, n_cu, stream);      \
    case (F) | FQ_QUANT_F16:                                                           \
        return launch_kron64<(F) | FQ_QUANT_F16>(x, left, right, diag, rows, out, n_cu, stream);
    switch (flags & FQ_CT_MASK) {
        FQ_CASE(FQ_OUT_PACKED)
        FQ_CASE(FQ_OUT_FAKEQUANT)
        FQ_CASE(FQ_OUT_PACKED | FQ_OUT_FAKEQUANT)
        FQ_CASE(FQ_OUT_TRANSFORM | FQ_OUT_PACKED)
        FQ_CASE(FQ_OUT_TRANSFORM | FQ_OUT_FAKEQUANT)
        FQ_CASE(FQ_OUT_TRANSFORM | FQ_OUT_PACKED | FQ_OUT_FAKEQUANT)
        case FQ_OUT_TRANSFORM:
        case FQ_OUT_TRANSFORM | FQ_QUANT_F16:
            return launch_kron64<FQ_OUT_TRANSFORM>(x, left, right, diag, rows, out, n_cu, stream);
        default:
            return -1000;
    }
#undef FQ_CASE
}
