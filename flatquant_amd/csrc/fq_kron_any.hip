// fq_kron_any.hip — last-resort Kronecker transform + INT4 quantisation for the factor pairs no MFMA kernel takes:
// N % 16 != 0, M > 128 or an odd byte count per token (e.g. Qwen2.5 ffn widths: 18944 = 128 x 148, 27648 = 144 x 192,
// 29568 = 168 x 176). The reference's Triton kernels mask their way through any (M, N) (kron_matmul.py:29-110); this is
// the counterpart of that generality: same mathematics and rounding points as the other kernels (U = fp16(X . R) with
// fp32 accumulation, Y = L^T . U in fp32), plain FMA loops instead of matrix instructions. Correct, not fast
// (measured 24 ms for 16384 tokens of 128 x 148, ~100x an MFMA kernel of that size: one LDS and one L2 load per FMA
// issue-bound on memory instructions); every shape of the BASELINE configs has a fast kernel.
//
// One 1024-thread workgroup per token. X and U live in LDS as fp16 [M][N]; thread t owns the outputs t, t + 1024, ...
// (<= 32 per thread: M * N <= 32768) in registers through statistics and quantisation; nibbles meet their neighbours
// through an LDS byte image before they are packed.
#include "fq_common.hpp"

namespace {

constexpr int ANY_T = 1024, ANY_MAXJ = 32;

__global__ __launch_bounds__(ANY_T) void fq_kron_any_kernel(const f16* __restrict__ x, const f16* __restrict__ L,
                                                            const f16* __restrict__ R, const f16* __restrict__ diag,
                                                            int64_t rows, int M, int N, FqQuantOut out, int flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int d = M * N;
    f16* X = reinterpret_cast<f16*>(smem);                 // [M][N], later the byte image of the nibbles
    f16* U = X + ((d + 7) & ~7);                           // [M][N]
    float* red = reinterpret_cast<float*>(U + ((d + 7) & ~7));  // [2][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nj = (d + ANY_T - 1) / ANY_T;

    for (int64_t tok = blockIdx.x; tok < rows; tok += gridDim.x) {
        __syncthreads();  // the previous token's readers of X / U / red are done
        const f16* xp = x + tok * (int64_t)d;
        for (int i = tid; i < d; i += ANY_T) X[i] = diag != nullptr ? xp[i] * diag[i] : xp[i];  // fp16 product (trans_utils.py:86-90)
        __syncthreads();
        for (int i = tid; i < d; i += ANY_T) {  // U[m][n'] = fp16(sum_n X[m][n] R[n][n'])
            const int m = i / N, np = i - m * N;
            const f16* xr = X + m * N;
            const f16* rc = R + np;
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;  // four chains: the loads of 8 steps are in flight together
            int n = 0;
#pragma unroll 2
            for (; n + 4 <= N; n += 4) {
                a0 = __builtin_fmaf((float)xr[n], (float)rc[n * N], a0);
                a1 = __builtin_fmaf((float)xr[n + 1], (float)rc[(n + 1) * N], a1);
                a2 = __builtin_fmaf((float)xr[n + 2], (float)rc[(n + 2) * N], a2);
                a3 = __builtin_fmaf((float)xr[n + 3], (float)rc[(n + 3) * N], a3);
            }
            for (; n < N; ++n) a0 = __builtin_fmaf((float)xr[n], (float)rc[n * N], a0);
            U[i] = (f16)((a0 + a1) + (a2 + a3));
        }
        __syncthreads();
        float y[ANY_MAXJ];
        float vmax = -INFINITY, vmin = INFINITY;
#pragma unroll
        for (int j = 0; j < ANY_MAXJ; ++j) {  // Y[m'][n] = sum_m L[m][m'] U[m][n]   (unrolled: y[] stays in registers)
            y[j] = 0.0f;
            const int i = tid + j * ANY_T;
            if (j < nj && i < d) {
                const int mp = i / N, n = i - mp * N;
                const f16* lc = L + mp;
                const f16* uc = U + n;
                float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
                int m = 0;
#pragma unroll 2
                for (; m + 4 <= M; m += 4) {
                    a0 = __builtin_fmaf((float)lc[m * M], (float)uc[m * N], a0);
                    a1 = __builtin_fmaf((float)lc[(m + 1) * M], (float)uc[(m + 1) * N], a1);
                    a2 = __builtin_fmaf((float)lc[(m + 2) * M], (float)uc[(m + 2) * N], a2);
                    a3 = __builtin_fmaf((float)lc[(m + 3) * M], (float)uc[(m + 3) * N], a3);
                }
                for (; m < M; ++m) a0 = __builtin_fmaf((float)lc[m * M], (float)uc[m * N], a0);
                float acc = (a0 + a1) + (a2 + a3);
                if (flags & FQ_ROUND_Y_F16) acc = (float)(f16)acc;
                y[j] = acc;
                vmax = fmaxf(vmax, acc);
                vmin = fminf(vmin, acc);
            }
        }
        vmax = fq_wave_max(vmax);
        vmin = fq_wave_min(vmin);
        if (lane == 0) {
            red[wave] = vmax;
            red[16 + wave] = vmin;
        }
        __syncthreads();  // also: U and X are no longer read
        vmax = red[0];
        vmin = red[16];
        for (int w = 1; w < ANY_T / 64; ++w) {
            vmax = fmaxf(vmax, red[w]);
            vmin = fminf(vmin, red[16 + w]);
        }
        if (flags & FQ_OUT_TRANSFORM) {
#pragma unroll
            for (int j = 0; j < ANY_MAXJ; ++j) {
                const int i = tid + j * ANY_T;
                if (j < nj && i < d) out.y[tok * (int64_t)d + i] = (f16)y[j];
            }
        }
        unsigned char* qb = reinterpret_cast<unsigned char*>(X);
        for (int ci = 0; ci < out.n_clips; ++ci) {
            if (!(flags & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT))) break;
            const bool h16 = (flags & FQ_QUANT_F16) != 0;
            const float scale = h16 ? fq_token_scale<FQ_QUANT_F16>(vmax, vmin, out.sig_max[ci], out.sig_min[ci], flags)
                                    : fq_token_scale<0>(vmax, vmin, out.sig_max[ci], out.sig_min[ci], flags);
#pragma unroll
            for (int j = 0; j < ANY_MAXJ; ++j) {
                const int i = tid + j * ANY_T;
                if (j >= nj || i >= d) continue;
                const int q = h16 ? fq_quant1<FQ_QUANT_F16>(y[j], scale) : fq_quant1<0>(y[j], scale);
                if (flags & FQ_OUT_PACKED) qb[i] = (unsigned char)(q & 15);
                if (flags & FQ_OUT_FAKEQUANT)
                    out.fq[ci][tok * (int64_t)d + i] = h16 ? fq_dequant1<FQ_QUANT_F16>(q, scale) : fq_dequant1<0>(q, scale);
            }
            if (flags & FQ_OUT_PACKED) {
                __syncthreads();
                if (tid == 0) out.scale[ci][tok] = (f16)scale;
                uint8_t* qp = out.q[ci] + tok * (int64_t)(d >> 1);
                for (int b = tid; b < (d >> 1); b += ANY_T) qp[b] = (uint8_t)(qb[2 * b] | (qb[2 * b + 1] << 4));
                __syncthreads();
            }
        }
    }
}

}  // namespace

// -1000: outside even this kernel's range (M * N > 32768, M or N > 256, odd N with packed output)
int fq_launch_kron_any(int flags, const f16* x, const f16* left, const f16* right, const f16* diag, int64_t rows, int M,
                       int N, const FqQuantOut& out, int n_cu, hipStream_t stream) {
    if (out.group_offsets != nullptr || (out.rt_flags & FQ_GROUP128)) return -1000;  // not in this kernel
    if (M < 1 || N < 1 || M > 256 || N > 256 || (int64_t)M * N > ANY_T * ANY_MAXJ) return -1000;
    if ((flags & FQ_OUT_PACKED) && ((M * N) & 1)) return -1000;
    const int d8 = (M * N + 7) & ~7;
    const size_t lds = (size_t)d8 * 4 + 32 * sizeof(float);
    FQ_RAISE_LDS_CAP(fq_kron_any_kernel, 160 * 1024);
    int64_t blocks = n_cu;
    if (blocks > rows) blocks = rows;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fq_kron_any_kernel, dim3((unsigned)blocks), dim3(ANY_T), lds, stream, x, left, right, diag, rows, M, N,
                       out, flags & ~FQ_WS_PREPARED);
    return (int)hipGetLastError();
}
