// fq_quant.hip — standalone quantisation kernels (HBM-bound byte work, no MFMA).
//
//   fq_rowquant_kernel      deploy/nn/quantization.py:13-36 (Quantizer.forward: 5-8 torch launches + the
//                           CUDA pack kernel) and flatquant/quant_utils.py:77-119 in ONE pass: a row is
//                           read once into registers, reduced, quantised, packed, written.
//   fq_rowquant_asym_kernel flatquant/quant_utils.py:33-46,109-117 (ActivationQuantizer(sym=False): the K / V / Q cache
//                           quantisers of llama_utils.py:124-132 under --k_asym --v_asym), fake-quant output, one pass
//   fq_sym_quant_kernel     deploy/kernels/quant.cu:13-47  (fp16 division, rn, clamp, low nibble first)
//   fq_sym_dequant_kernel   deploy/kernels/quant.cu:66-85
#include "fq_common.hpp"

namespace {

// Running extrema of the 16-byte chunks a lane holds. fp16: packed pairs (v_pk_max/min_f16, exact: the data is fp16);
// bf16: each dword's halves widened to fp32 (one shift, one and) and v_max3 / v_min3_f32.
constexpr int RQ_THREADS = 256;
constexpr int RQ_MAXCH = 16;  // 16-byte chunks per thread: cols <= 256 * 16 * 8 = 32768

// One workgroup per row; each thread keeps its 16-byte chunks (8 fp16) in registers.
template <int FLAGS, int NCH, typename T = f16>
__global__ __launch_bounds__(RQ_THREADS) void fq_rowquant_kernel(const T* __restrict__ x, int64_t rows,
                                                                 int cols, FqQuantOut out) {
    typedef typename FqVec<T>::x8 X8;
    constexpr bool IS16 = FqVec<T>::is_f16;
    __shared__ float red[2][RQ_THREADS / 64];
    const int tid = threadIdx.x;
    const int nchunks = cols >> 3;  // cols % 8 == 0 enforced by the host
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const uint4* xp = reinterpret_cast<const uint4*>(x + row * (int64_t)cols);
        X8 v[NCH];
        float vmax = -INFINITY, vmin = INFINITY;
        RowExtrema<T> ext;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = tid + k * RQ_THREADS;
            if (ch < nchunks) {
                v[k] = __builtin_bit_cast(X8, xp[ch]);
                ext.take(v[k]);
            }
        }
        vmax = fq_wave_max(ext.vmax());
        vmin = fq_wave_min(ext.vmin());
        __syncthreads();  // protect `red` against the previous iteration's readers
        if ((tid & 63) == 0) {
            red[0][tid >> 6] = vmax;
            red[1][tid >> 6] = vmin;
        }
        __syncthreads();
        vmax = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
        vmin = fminf(fminf(red[1][0], red[1][1]), fminf(red[1][2], red[1][3]));

        for (int ci = 0; ci < out.n_clips; ++ci) {
            const float scale = fq_token_scale<FLAGS, T>(vmax, vmin, out.sig_max[ci], out.sig_min[ci], out.rt_flags);
            const float h16_inv = fq_fast_inv(scale);
            const FqH16Recip h16rc = ((FLAGS & FQ_QUANT_F16) && IS16) ? fq_h16_recip(scale) : FqH16Recip{0.0f, 0.0f};
            const bool h16_clamp = fq_h16_needs_clamp(vmax, vmin, h16_inv);
            if (FLAGS & FQ_OUT_PACKED) {
                if (tid == 0) reinterpret_cast<T*>(out.scale[ci])[row] = ((out.rt_flags & FQ_RATIO_POST) && vmax == 0.0f && vmin == 0.0f) ? (T)0.0f : (T)scale;
                uint32_t* qp = reinterpret_cast<uint32_t*>(out.q[ci] + row * (int64_t)(cols >> 1));
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int ch = tid + k * RQ_THREADS;
                    if (ch < nchunks) {
                        uint32_t d = 0;
                        if ((FLAGS & FQ_QUANT_F16) && IS16) {   // packed pairs, exact fp16 quotient without a division (fq_quant8_h16)
                            const u32x4 xv = __builtin_bit_cast(u32x4, v[k]);
                            d = h16_clamp ? fq_quant8_h16<true>(xv[0], xv[1], xv[2], xv[3], h16rc)
                                          : fq_quant8_h16<false>(xv[0], xv[1], xv[2], xv[3], h16rc);
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                d |= (uint32_t)(fq_quant1<FLAGS, T>((float)v[k][e], scale) & 15) << (4 * e);
                        }
                        qp[ch] = d;
                    }
                }
            }
            if (FLAGS & FQ_OUT_FAKEQUANT) {
                uint4* fp = reinterpret_cast<uint4*>(out.fq[ci] + row * (int64_t)cols);
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int ch = tid + k * RQ_THREADS;
                    if (ch < nchunks) {
                        X8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            o[e] = fq_dequant1<FLAGS, T>((FLAGS & FQ_QUANT_F16) ? fq_quant1_h(v[k][e], (T)scale)
                                                                                : fq_quant1<FLAGS, T>((float)v[k][e], scale), scale);
                        fp[ch] = __builtin_bit_cast(uint4, o);
                    }
                }
            }
        }
    }
}

// q = clamp(rn(x /h scale[row])) — fp16 division exactly as __hdiv (quant.cu:40): the fp32 quotient of
// two fp16 values rounded to fp16 is the correctly rounded fp16 quotient (24 >= 2*11+2).
__device__ __forceinline__ int symq1(f16 x, f16 s) {
    return fq_quant1_h(x, s);  // native fp16 division == correctly rounded (fq_common.hpp)
}

// Fast path: cols % 8 == 0. One thread packs 8 values (16-byte load, 4-byte store).
__global__ __launch_bounds__(256) void fq_sym_quant_vec_kernel(const f16* __restrict__ x,
                                                               const f16* __restrict__ scale,
                                                               int64_t rows, int cols,
                                                               uint8_t* __restrict__ q) {
    const int cpr = cols >> 3;  // chunks per row
    const int64_t total = rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cpr;
        const f16 s = scale[row];
        const f16x8 v = __builtin_bit_cast(f16x8, reinterpret_cast<const uint4*>(x)[i]);
        uint32_t d = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) d |= (uint32_t)(symq1(v[e], s) & 15) << (4 * e);
        reinterpret_cast<uint32_t*>(q)[i] = d;
    }
}

// General path: any cols (odd tail -> high nibble 0, as quant.cu:27-46 memset + `safe`).
__global__ __launch_bounds__(256) void fq_sym_quant_gen_kernel(const f16* __restrict__ x,
                                                               const f16* __restrict__ scale,
                                                               int64_t rows, int cols, int cols_dst,
                                                               uint8_t* __restrict__ q) {
    const int64_t total = rows * cols_dst;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cols_dst;
        const int cd = (int)(i - row * cols_dst);
        const f16 s = scale[row];
        const f16* xr = x + row * (int64_t)cols;
        uint32_t b = (uint32_t)(symq1(xr[2 * cd], s) & 15);
        if (2 * cd + 1 < cols) b |= (uint32_t)(symq1(xr[2 * cd + 1], s) & 15) << 4;
        q[i] = (uint8_t)b;
    }
}

// x = scale_row[r] * scale_col[c] * half(int(q / 10.0f) clamped to +-65176) * half(10), every product
// rounded to fp16 left-to-right (quant.cu:5-10, 83-84).
__global__ __launch_bounds__(256) void fq_sym_dequant_kernel(const int32_t* __restrict__ q,
                                                             const f16* __restrict__ srow,
                                                             const f16* __restrict__ scol, int64_t rows,
                                                             int cols, f16* __restrict__ x) {
    const int64_t total = rows * (int64_t)cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cols;
        const int col = (int)(i - row * cols);
        int iv = (int)((float)q[i] / 10.0f);  // C truncation toward zero
        iv = max(-65176, min(65176, iv));
        const f16 xe = (f16)iv;  // __int2half_rn
        f16 r = srow[row] * scol[col];
        r = r * xe;
        r = r * (f16)10.0f;
        x[i] = r;
    }
}

// Wave-per-row variant for cols <= 64 * 8 * NCH: the whole row sits in one wave's registers (NCH 16-byte chunks per
// lane, all loads issued back to back), max/min by a register wave all-reduce, no LDS and no workgroup barrier.
// (The one-workgroup-per-row kernel above keeps only 2 loads per lane in flight and pays two barriers per row:
// 55 us for 16384 x 4096, where moving the same bytes takes 28 us.) Rows are handed out grid-stride per wave.
// MULTI (short rows, cols <= 256: heads, 128-element groups): a row occupies 2^lg lanes, 64 >> lg rows per wave — one row per
// wave would leave 3 of 4 lanes idle at cols = 128 (measured 0.6-1.1 TB/s against 3.5-3.8 with the lanes filled); the
// extrema are reduced by xor butterflies inside the lane group and every row carries its own scale through the same epilogue.
template <int FLAGS, int NCH, bool MULTI = false, typename T = f16>
// (occupancy bound for the packed fp16-quantiser builds only — the deploy Quantizer: their epilogue is 5 VALU per element on
//  packed pairs and fits; the fp32-quantiser builds keep the compiler's own choice)
__global__ __launch_bounds__(256, ((FLAGS & (FQ_QUANT_F16 | FQ_OUT_FAKEQUANT)) == FQ_QUANT_F16 && FqVec<T>::is_f16) ? (NCH > 24 ? 2 : NCH > 16 ? 3 : 4) : 1)
void fq_rowquant_wave_kernel(const T* __restrict__ x, int64_t rows, int cols,
                                                               FqQuantOut out, int lg) {
    static_assert(!MULTI || NCH == 1, "short rows: one chunk per lane");
    typedef typename FqVec<T>::x8 X8;
    constexpr bool IS16 = FqVec<T>::is_f16;
    const int wl = threadIdx.x & 63;
    const int lpr = MULTI ? (1 << lg) : 64, rpw = MULTI ? (64 >> lg) : 1;
    const int lane = MULTI ? (wl & (lpr - 1)) : wl;            // the lane's position inside its row
    const int nchunks = cols >> 3;
    const int64_t nw = (int64_t)gridDim.x * 4 * rpw;
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw; row0 < rows; row0 += nw) {
        const int64_t rown = row0 + (MULTI ? (wl >> lg) : 0);
        const bool live = !MULTI || rown < rows;
        const int64_t row = live ? rown : rows - 1;              // (idle lane groups of the last wave re-read the last row, store nothing)
        const u32x4* xp = reinterpret_cast<const u32x4*>(x + row * (int64_t)cols);
        X8 v[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + k * lpr;
            v[k] = __builtin_bit_cast(X8, (ch < nchunks) ? __builtin_nontemporal_load(xp + ch) : u32x4{0, 0, 0, 0});
        }
        float vmax = -INFINITY, vmin = INFINITY;
        {   // extrema on packed fp16 pairs (the data IS fp16: exact), two values per instruction; bf16: RowExtrema
            RowExtrema<T> ext;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                if (lane + k * lpr < nchunks) ext.take(v[k]);
            }
            vmax = ext.vmax();
            vmin = ext.vmin();
        }
        if (MULTI) {
            for (int m = 1; m < lpr; m <<= 1) {
                vmax = fmaxf(vmax, __shfl_xor(vmax, m));
                vmin = fminf(vmin, __shfl_xor(vmin, m));
            }
        } else {
            vmax = fq_wave_max(vmax);
            vmin = fq_wave_min(vmin);
        }
        for (int ci = 0; ci < out.n_clips; ++ci) {
            const float scale = fq_token_scale<FLAGS, T>(vmax, vmin, out.sig_max[ci], out.sig_min[ci], out.rt_flags);
            const float inv = fq_fast_inv(scale);
            const FqH16Recip h16rc = ((FLAGS & FQ_QUANT_F16) && IS16) ? fq_h16_recip(scale) : FqH16Recip{0.0f, 0.0f};
            const bool h16_clamp = fq_h16_needs_clamp(vmax, vmin, inv);
            if (FLAGS & FQ_OUT_PACKED) {
                if (lane == 0 && live) reinterpret_cast<T*>(out.scale[ci])[row] = ((out.rt_flags & FQ_RATIO_POST) && vmax == 0.0f && vmin == 0.0f) ? (T)0.0f : (T)scale;
                uint32_t* qp = reinterpret_cast<uint32_t*>(out.q[ci] + row * (int64_t)(cols >> 1));
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int ch = lane + k * lpr;
                    uint32_t d;
                    if ((FLAGS & FQ_QUANT_F16) && IS16) {
                        const u32x4 xv = __builtin_bit_cast(u32x4, v[k]);   // packed pairs, exact fp16 quotient (fq_quant8_h16)
                        d = h16_clamp ? fq_quant8_h16<true>(xv[0], xv[1], xv[2], xv[3], h16rc)
                                      : fq_quant8_h16<false>(xv[0], xv[1], xv[2], xv[3], h16rc);
                    } else if (FLAGS & FQ_QUANT_F16) {   // bf16 arithmetic: the quotient rounded to bf16, element by element
                        d = 0;
#pragma unroll
                        for (int e = 0; e < 8; ++e) d |= (uint32_t)(fq_quant1_h(v[k][e], (T)scale) & 15) << (4 * e);
                    } else {
                        float dmax = 0.0f;
                        const f32x2 inv2 = {inv, inv};
                        const f32x2 q01 = fq_qfast2(f32x2{(float)v[k][0], (float)v[k][1]}, inv2, dmax);
                        const f32x2 q23 = fq_qfast2(f32x2{(float)v[k][2], (float)v[k][3]}, inv2, dmax);
                        const f32x2 q45 = fq_qfast2(f32x2{(float)v[k][4], (float)v[k][5]}, inv2, dmax);
                        const f32x2 q67 = fq_qfast2(f32x2{(float)v[k][6], (float)v[k][7]}, inv2, dmax);
                        d = fq_pack8(q01.x, q01.y, q23.x, q23.y, q45.x, q45.y, q67.x, q67.y);
                        if (fq_wave_needs_exact(dmax))
                            d = fq_pack8(fq_qexact((float)v[k][0], scale), fq_qexact((float)v[k][1], scale),
                                         fq_qexact((float)v[k][2], scale), fq_qexact((float)v[k][3], scale),
                                         fq_qexact((float)v[k][4], scale), fq_qexact((float)v[k][5], scale),
                                         fq_qexact((float)v[k][6], scale), fq_qexact((float)v[k][7], scale));
                    }
                    if (ch < nchunks && live) qp[ch] = d;
                }
            }
            if (FLAGS & FQ_OUT_FAKEQUANT) {
                uint4* fp = reinterpret_cast<uint4*>(out.fq[ci] + row * (int64_t)cols);
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int ch = lane + k * lpr;
                    X8 o;
                    if (FLAGS & FQ_QUANT_F16) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            o[e] = fq_dequant1<FLAGS, T>(fq_quant1_h(v[k][e], (T)scale), scale);
                    } else {
                        float dmax = 0.0f;
                        float r[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) r[e] = fq_qfast((float)v[k][e], inv, dmax);
                        if (fq_wave_needs_exact(dmax)) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) r[e] = fq_qexact((float)v[k][e], scale);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = fq_fake<T>(scale, r[e]);
                    }
                    if (ch < nchunks && live) fp[ch] = __builtin_bit_cast(uint4, o);
                }
            }
        }
    }
}

// deploy.nn.RMSNorm alone (deploy/nn/normalization.py:16-23): a wave per row, the row in registers.
template <int NCH>
__global__ __launch_bounds__(256) void fq_rmsnorm_kernel(const f16* __restrict__ x, f16* __restrict__ y, int64_t rows,
                                                         int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int nchunks = cols >> 3;
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += nw) {
        const u32x4* xp = reinterpret_cast<const u32x4*>(x + row * (int64_t)cols);
        f16x8 v[NCH];
        float ss[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + k * 64;
            v[k] = (ch < nchunks) ? __builtin_bit_cast(f16x8, __builtin_nontemporal_load(xp + ch)) : f16x8{0};
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) ss[e & 3] = __builtin_fmaf((float)v[k][e], (float)v[k][e], ss[e & 3]);
        const float tot = fq_wave_sum((ss[0] + ss[1]) + (ss[2] + ss[3]));
        const float rinv = __builtin_amdgcn_rsqf(tot / (float)cols + eps);
        uint4* yp = reinterpret_cast<uint4*>(y + row * (int64_t)cols);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + k * 64;
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fq_mul_to_f16((float)v[k][e], rinv);
            if (ch < nchunks) yp[ch] = __builtin_bit_cast(uint4, o);
        }
    }
}

template <int FLAGS, typename T>
int launch_rowquant(const T* x, int64_t rows, int cols, const FqQuantOut& out, int n_cu,
                    hipStream_t stream) {
    if ((cols >> 3) <= 32) {   // short rows: several rows per wave (2^lg lanes each)
        int lg = 0;
        while ((1 << lg) < (cols >> 3)) ++lg;
        const int rpw = 64 >> lg;
        int64_t wb = ((rows + rpw - 1) / rpw + 3) / 4;
        if (wb > (int64_t)n_cu * 8) wb = (int64_t)n_cu * 8;
        if (wb < 1) wb = 1;
        hipLaunchKernelGGL((fq_rowquant_wave_kernel<FLAGS, 1, true, T>), dim3((unsigned)wb), dim3(256), 0, stream, x, rows, cols, out, lg);
        return (int)hipGetLastError();
    }
    {   // wave-per-row fast path: the row fits one wave's registers (up to 32 chunks of 16 bytes per lane)
        const int nchw = ((cols >> 3) + 63) / 64;
        int64_t wb = (rows + 3) / 4;
        if (wb > (int64_t)n_cu * 8) wb = (int64_t)n_cu * 8;
        if (wb < 1) wb = 1;
#define FQ_RW(N)                                                                                                  \
    if (nchw <= (N)) {                                                                                            \
        hipLaunchKernelGGL((fq_rowquant_wave_kernel<FLAGS, (N), false, T>), dim3((unsigned)wb), dim3(256), 0, stream, x, rows, \
                           cols, out, 6);                                                                            \
        return (int)hipGetLastError();                                                                            \
    }
        FQ_RW(4) FQ_RW(8) FQ_RW(16) FQ_RW(24)  // beyond 24 chunks per lane the one-workgroup-per-row kernel measured faster
                                               // (round 2, fp16 quantiser, 28 chunks at two waves per SIMD: 111.7 vs 110-114 us — no gain)
#undef FQ_RW
    }
    const int nch = ((cols >> 3) + RQ_THREADS - 1) / RQ_THREADS;
    int64_t blocks = rows;
    const int64_t cap = (int64_t)n_cu * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    dim3 g((unsigned)blocks), b(RQ_THREADS);
#define FQ_RQ(N)                                                                                 \
    if (nch <= (N)) {                                                                            \
        hipLaunchKernelGGL((fq_rowquant_kernel<FLAGS, (N), T>), g, b, 0, stream, x, rows, cols, out); \
        return (int)hipGetLastError();                                                           \
    }
    FQ_RQ(2) FQ_RQ(4) FQ_RQ(8) FQ_RQ(RQ_MAXCH)
#undef FQ_RQ
    return -1000;
}

// ---------------------------------------------------------------------------------------------------
// Asymmetric per-row fake quantisation (FQ_ASYM): rows of `cols` fp16 values, LPR = 2^k lanes per row (cols = 128, a
// head: 16 lanes, four rows per wave), each lane keeps its <= NV 16-byte vectors in registers (NV == 0: rows longer than
// the register budget are read a second time — from L2). Pinned arithmetic, as torch evaluates the reference
// (quant_utils.py:86-117, 33-46) on an fp16 activation:
//   xmax = max(amax, 0), xmin = min(amin, 0); times the clip factors; both zero -> (-1, +1);
//   scale = (xmax - xmin) / 15; zero = rint(-xmin / scale); q = clamp(rint(x / scale) + zero, 0, 15);
//   out = fp16(scale * (q - zero))
// F16A == false (lac with fp32 clip parameters: the (1,)-shaped fp32 sigmoid promotes extrema, scale, zero, the quotient
// and the product to fp32): every operation rounds to fp32. F16A == true (no lac, clip_ratio, or a half()'ed module):
// every operation rounds to fp16 — the extremum x factor product (fp32 opmath, then fp16: two roundings), the
// difference, both quotients (fp16(a / b) from the correctly rounded fp32 quotient IS the correctly rounded fp16
// quotient: 24 >= 2 * 11 + 2 bits), the product. (round_ste's (r - t) + t is r exactly in both widths.)
template <bool F16A, int NV, typename T = f16>
__global__ __launch_bounds__(256) void fq_rowquant_asym_kernel(const T* __restrict__ x, int64_t rows, int cols, int lpr_log2,
                                                               FqQuantOut out) {
    typedef typename FqVec<T>::x8 X8;
    const int lane = threadIdx.x & 63;
    const int lpr = 1 << lpr_log2, sub = lane & (lpr - 1), rpw = 64 >> lpr_log2;
    const int nvec = cols >> 3;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r0 = wave * rpw; r0 < rows; r0 += nwaves * rpw) {
        const int64_t row = r0 + (lane >> lpr_log2);
        const bool live = row < rows;
        const uint4* xp = reinterpret_cast<const uint4*>(x + (live ? row : 0) * (int64_t)cols);
        X8 v[NV > 0 ? NV : 1];
        RowExtrema<T> ext;
        auto take = [&](const X8& w) { ext.take(w); };
        if (NV > 0) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int i = sub + k * lpr;
                if (live && i < nvec) {
                    v[k] = __builtin_bit_cast(X8, xp[i]);
                    take(v[k]);
                }
            }
        } else {
            for (int i = sub; live && i < nvec; i += lpr) take(__builtin_bit_cast(X8, xp[i]));
        }
        float vmax = ext.vmax(), vmin = ext.vmin();
        for (int m = 1; m < lpr; m <<= 1) {  // the lanes of a row are an aligned group of 2^k: xor butterflies stay inside it
            vmax = fmaxf(vmax, __shfl_xor(vmax, m));
            vmin = fminf(vmin, __shfl_xor(vmin, m));
        }
        vmax = fmaxf(vmax, 0.0f);
        vmin = fminf(vmin, 0.0f);
        for (int ci = 0; ci < out.n_clips; ++ci) {
            float xmax, xmin;
            if (F16A) {
                xmax = (float)fq_mul_to<T>(vmax, out.sig_max[ci]);
                xmin = (float)fq_mul_to<T>(vmin, out.sig_min[ci]);
            } else {
                xmax = vmax * out.sig_max[ci];
                xmin = vmin * out.sig_min[ci];
            }
            if (xmax == 0.0f && xmin == 0.0f) {
                xmin = -1.0f;
                xmax = 1.0f;
            }
            float scale, zero;
            if (F16A) {
                const float d = (float)(T)(xmax - xmin);
                scale = (float)(T)(d / 15.0f);
                zero = __builtin_rintf((float)(T)(-xmin / scale));
            } else {
                const float d = xmax - xmin;
                scale = d / 15.0f;
                zero = __builtin_rintf(-xmin / scale);
            }
            uint4* op = reinterpret_cast<uint4*>(out.fq[ci] + (live ? row : 0) * (int64_t)cols);
            // fp32 route without a division (the two-sided magic-number test of fq_quant8_two, fq_common.hpp): with
            // ilo / ihi = v_rcp_f32(scale) (1 -+ 2^-21), u = fma(x, ilo, 1.5 * 2^23) and v = fma(x, ihi, ...) bracket
            // rint(fl(x / scale)); u == v on all eight elements (of every lane at work) proves u, else the vector is redone
            // with the division. u - (1.5 * 2^23 - zero) = rint(x / scale) + zero exactly (integers below 2^24).
            const float inv = fq_fast_inv(scale), ilo = fq_inv_lo(inv), ihi = fq_inv_hi(inv), cz = FQ_MAGIC - zero;
            const bool fast_ok = !F16A && fmaxf(vmax, -vmin) * inv < 2097152.0f;
            auto emit = [&](const X8& w, int i) {
                X8 o;
                if (fast_ok) {
                    float u[8];
                    bool differ = false;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xf = (float)w[e];
                        u[e] = __builtin_fmaf(xf, ilo, FQ_MAGIC);
                        differ |= u[e] != __builtin_fmaf(xf, ihi, FQ_MAGIC);
                    }
                    if (!__any(differ)) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float q = __builtin_amdgcn_fmed3f(u[e] - cz, 0.0f, 15.0f);
                            o[e] = fq_mul_to<T>(scale, q - zero);
                        }
                        op[i] = __builtin_bit_cast(uint4, o);
                        return;
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = (float)w[e] / scale;            // correctly rounded fp32 division
                    if (F16A) t = (float)(T)t;
                    float q = __builtin_rintf(t) + zero;      // (integers of small magnitude: exact in either width)
                    q = __builtin_amdgcn_fmed3f(q, 0.0f, 15.0f);
                    o[e] = fq_mul_to<T>(scale, q - zero);     // fp32 product rounded to fp32, then to fp16; F16A: the product
                }                                             // of an 11-bit by a 5-bit significand is exact in fp32
                op[i] = __builtin_bit_cast(uint4, o);
            };
            if (NV > 0) {
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int i = sub + k * lpr;
                    if (live && i < nvec) emit(v[k], i);
                }
            } else {
                for (int i = sub; live && i < nvec; i += lpr) emit(__builtin_bit_cast(X8, xp[i]), i);
            }
        }
    }
}

template <bool F16A, typename T>
int launch_rowquant_asym(const T* x, int64_t rows, int cols, const FqQuantOut& out, int n_cu, hipStream_t stream) {
    const int nvec = cols >> 3;
    int lg = 0;
    while (lg < 6 && (1 << lg) < nvec) ++lg;  // lanes per row: the power of two >= cols / 8, at most a wave
    const int per_lane = (nvec + (1 << lg) - 1) >> lg;
    const int64_t waves = (rows + (64 >> lg) - 1) / (64 >> lg);
    int64_t blocks = (waves + 3) / 4;
    if (blocks > (int64_t)n_cu * 8) blocks = (int64_t)n_cu * 8;
    if (blocks < 1) blocks = 1;
#define FQ_ASYM_LAUNCH(NV_)                                                                                             \
    hipLaunchKernelGGL((fq_rowquant_asym_kernel<F16A, NV_, T>), dim3((unsigned)blocks), dim3(256), 0, stream, x, rows, cols, lg, out)
    if (per_lane <= 1) FQ_ASYM_LAUNCH(1);
    else if (per_lane <= 4) FQ_ASYM_LAUNCH(4);
    else if (per_lane <= 8) FQ_ASYM_LAUNCH(8);
    else FQ_ASYM_LAUNCH(0);
#undef FQ_ASYM_LAUNCH
    return (int)hipGetLastError();
}

}  // namespace

template <typename T>
static int launch_rowquant_any(int flags, const T* x, int64_t rows, int cols, const FqQuantOut& out, int n_cu,
                               hipStream_t stream) {
#define FQ_CASE(F)                                                              \
    case (F):                                                                   \
        return launch_rowquant<(F), T>(x, rows, cols, out, n_cu, stream);       \
    case (F) | FQ_QUANT_F16:                                                    \
        return launch_rowquant<(F) | FQ_QUANT_F16, T>(x, rows, cols, out, n_cu, stream);
    if (flags & FQ_ASYM) {  // (fq_capi admits it with FQ_OUT_FAKEQUANT alone)
        if ((flags & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT | FQ_OUT_TRANSFORM)) != FQ_OUT_FAKEQUANT) return -1000;
        return (flags & FQ_QUANT_F16) ? launch_rowquant_asym<true, T>(x, rows, cols, out, n_cu, stream)
                                      : launch_rowquant_asym<false, T>(x, rows, cols, out, n_cu, stream);
    }
    switch (flags & FQ_CT_MASK) {
        FQ_CASE(FQ_OUT_PACKED)
        FQ_CASE(FQ_OUT_FAKEQUANT)
        FQ_CASE(FQ_OUT_PACKED | FQ_OUT_FAKEQUANT)
        default:
            return -1000;
    }
#undef FQ_CASE
}

int fq_launch_rowquant(int flags, const f16* x, int64_t rows, int cols, const FqQuantOut& out, int n_cu,
                       hipStream_t stream) {
    if (flags & FQ_DT_BF16) return launch_rowquant_any<bf16>(flags & ~FQ_DT_BF16, (const bf16*)x, rows, cols, out, n_cu, stream);
    return launch_rowquant_any<f16>(flags, x, rows, cols, out, n_cu, stream);
}

int fq_launch_sym_quant(const f16* x, const f16* scale, int64_t rows, int cols, uint8_t* q, int n_cu,
                        hipStream_t stream) {
    if ((cols & 7) == 0) {
        const int64_t total = rows * (cols >> 3);
        int64_t blocks = (total + 255) / 256;
        if (blocks > (int64_t)n_cu * 16) blocks = (int64_t)n_cu * 16;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(fq_sym_quant_vec_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, scale,
                           rows, cols, q);
    } else {
        const int cols_dst = (cols + 1) / 2;
        const int64_t total = rows * cols_dst;
        int64_t blocks = (total + 255) / 256;
        if (blocks > (int64_t)n_cu * 16) blocks = (int64_t)n_cu * 16;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(fq_sym_quant_gen_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, scale,
                           rows, cols, cols_dst, q);
    }
    return (int)hipGetLastError();
}

int fq_launch_sym_dequant(const int32_t* q, const f16* srow, const f16* scol, int64_t rows, int cols,
                          f16* x, int n_cu, hipStream_t stream) {
    const int64_t total = rows * (int64_t)cols;
    int64_t blocks = (total + 255) / 256;
    if (blocks > (int64_t)n_cu * 16) blocks = (int64_t)n_cu * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fq_sym_dequant_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, q, srow, scol,
                       rows, cols, x);
    return (int)hipGetLastError();
}

// x_up * act_fn(x_gate) alone (modeling_llama.py:277-278): 16 bytes per lane per tensor, grid-stride.
__global__ __launch_bounds__(256) void fq_silu_mul_kernel(const f16* __restrict__ gate, const f16* __restrict__ up,
                                                          f16* __restrict__ y, int64_t chunks) {
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < chunks; i += step) {
        const f16x8 g = __builtin_bit_cast(f16x8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(gate) + i));
        const f16x8 u = __builtin_bit_cast(f16x8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(up) + i));
        reinterpret_cast<uint4*>(y)[i] = __builtin_bit_cast(uint4, fq_silu_mul8(g, u));
    }
}

int fq_launch_silu_mul(const f16* gate, const f16* up, f16* y, int64_t n, int n_cu, hipStream_t stream) {
    if (n & 7) return -1000;
    const int64_t chunks = n >> 3;
    int64_t blocks = (chunks + 255) / 256;
    if (blocks > (int64_t)n_cu * 16) blocks = (int64_t)n_cu * 16;
    hipLaunchKernelGGL(fq_silu_mul_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, gate, up, y, chunks);
    return (int)hipGetLastError();
}

int fq_launch_rmsnorm(const f16* x, f16* y, int64_t rows, int cols, float eps, int n_cu, hipStream_t stream) {
    if ((cols & 7) || cols < 8 || cols > 16384) return -1000;
    const int nchw = ((cols >> 3) + 63) / 64;
    int64_t wb = (rows + 3) / 4;
    if (wb > (int64_t)n_cu * 8) wb = (int64_t)n_cu * 8;
    if (wb < 1) wb = 1;
#define FQ_RN(N)                                                                                                     \
    if (nchw <= (N)) {                                                                                               \
        hipLaunchKernelGGL((fq_rmsnorm_kernel<(N)>), dim3((unsigned)wb), dim3(256), 0, stream, x, y, rows, cols, eps); \
        return (int)hipGetLastError();                                                                               \
    }
    FQ_RN(4) FQ_RN(8) FQ_RN(16) FQ_RN(32)
#undef FQ_RN
    return -1000;
}

// ---------------------------------------------------------------------------------------------------
// ActivationQuantizer with bits != 4 (round 4): get_qmin_qmax (quant_utils.py:10-16) makes the grid a parameter — --a_bits / --q_bits /
// --k_bits / --v_bits of args_utils.py:38,101,108,116 — and fake_quant (quant_utils.py:76-119, 18-46) is otherwise the same arithmetic.
// Fake-quant output only (the packed format is the INT4 one). No reference script runs anything but 4 bits, so this is the plain form:
// a wave per row, the row read twice (extrema, then values: the second read comes from L2), IEEE divisions. Pinned arithmetic = the
// 4-bit kernels' with 7 / 15 replaced by qmax (oracle: rowquant(bits=) / rowquant_asym(bits=), pinned by tests/golden/act_bits.npz):
//   sym:  xmax = max(amax, 0), xmin = min(amin, 0), times the clip factors, m = max(|xmin|, xmax), scale = m / qmax (1 if m == 0),
//         q = clamp(rint(x / scale), -qmax - 1, qmax), out = T(scale q);      qmax = 2^(bits-1) - 1
//   asym: both zero -> (-1, +1); scale = (xmax - xmin) / qmax, zero = rint(-xmin / scale), q = clamp(rint(x / scale) + zero, 0, qmax),
//         out = T(scale (q - zero));                                            qmax = 2^bits - 1
// F16A == false: lac with fp32 (1,)-shaped clip parameters — every operation in fp32. F16A == true: no lac / clip_ratio / a module in the
// activation dtype — every operation rounds to T (sig_lowp: the extremum x factor product too; with factor 1 it is exact either way).
template <typename T, bool ASYM, bool F16A>
__global__ __launch_bounds__(256) void fq_fakequant_bits_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t rows, int cols,
                                                                float sig_max, float sig_min, float qmax, int sig_lowp) {
    typedef typename FqVec<T>::x8 X8;
    const int lane = threadIdx.x & 63, nvec = cols >> 3;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    auto rnd = [](float v) -> float { return F16A ? (float)(T)v : v; };
    for (int64_t row = wave; row < rows; row += nwaves) {
        const uint4* xp = reinterpret_cast<const uint4*>(x + row * (int64_t)cols);
        uint4* yp = reinterpret_cast<uint4*>(y + row * (int64_t)cols);
        float vmax = -INFINITY, vmin = INFINITY;
        for (int i = lane; i < nvec; i += 64) {
            const X8 w = __builtin_bit_cast(X8, xp[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                vmax = fmaxf(vmax, (float)w[e]);
                vmin = fminf(vmin, (float)w[e]);
            }
        }
        vmax = fmaxf(fq_wave_max(vmax), 0.0f);
        vmin = fminf(fq_wave_min(vmin), 0.0f);
        float xmax, xmin;
        if (F16A && sig_lowp) {
            xmax = (float)fq_mul_to<T>(vmax, sig_max);
            xmin = (float)fq_mul_to<T>(vmin, sig_min);
        } else {
            xmax = vmax * sig_max;
            xmin = vmin * sig_min;
        }
        float scale, zero = 0.0f;
        if (ASYM) {
            if (xmax == 0.0f && xmin == 0.0f) {
                xmin = -1.0f;
                xmax = 1.0f;
            }
            scale = rnd(rnd(xmax - xmin) / qmax);
            zero = __builtin_rintf(rnd(-xmin / scale));
        } else {
            const float m = fmaxf(fabsf(xmin), xmax);
            scale = rnd(m / qmax);
            if (m == 0.0f) scale = 1.0f;
        }
        for (int i = lane; i < nvec; i += 64) {
            const X8 w = __builtin_bit_cast(X8, xp[i]);
            X8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = rnd((float)w[e] / scale);   // correctly rounded fp32 division (then to T: the 16-bit quotient)
                float q = __builtin_rintf(t);
                if (ASYM) {
                    q = __builtin_amdgcn_fmed3f(rnd(q + zero), 0.0f, qmax);
                    o[e] = fq_mul_to<T>(scale, q - zero);
                } else {
                    q = __builtin_amdgcn_fmed3f(q, -qmax - 1.0f, qmax);
                    o[e] = fq_fake<T>(scale, q);            // (+0.0 for a zero digit, like the 4-bit kernels)
                }
            }
            yp[i] = __builtin_bit_cast(uint4, o);
        }
    }
}

template <typename T>
static int launch_fakequant_bits_t(const T* x, T* y, int64_t rows, int cols, float sig_max, float sig_min, int bits, int flags, int n_cu,
                                   hipStream_t stream) {
    int64_t wb = (rows + 3) / 4;
    if (wb > (int64_t)n_cu * 8) wb = (int64_t)n_cu * 8;
    if (wb < 1) wb = 1;
    const bool asym = (flags & FQ_ASYM) != 0, f16a = (flags & FQ_QUANT_F16) != 0;
    const float qmax = asym ? (float)((1 << bits) - 1) : (float)((1 << (bits - 1)) - 1);
    const int sl = (flags & FQ_SIG_F16) || asym ? 1 : 0;   // (the asymmetric route always rounds the 16-bit route's products: rowquant_asym)
#define FQ_FB(A_, F_)                                                                                                         \
    hipLaunchKernelGGL((fq_fakequant_bits_kernel<T, A_, F_>), dim3((unsigned)wb), dim3(256), 0, stream, x, y, rows, cols, sig_max, \
                       sig_min, qmax, sl)
    if (asym) { if (f16a) FQ_FB(true, true); else FQ_FB(true, false); }
    else { if (f16a) FQ_FB(false, true); else FQ_FB(false, false); }
#undef FQ_FB
    return (int)hipGetLastError();
}

int fq_launch_fakequant_bits(int bf16_dtype, const void* x, void* y, int64_t rows, int cols, float sig_max, float sig_min, int bits, int flags,
                             int n_cu, hipStream_t stream) {
    return bf16_dtype ? launch_fakequant_bits_t<bf16>((const bf16*)x, (bf16*)y, rows, cols, sig_max, sig_min, bits, flags, n_cu, stream)
                      : launch_fakequant_bits_t<f16>((const f16*)x, (f16*)y, rows, cols, sig_max, sig_min, bits, flags, n_cu, stream);
}
