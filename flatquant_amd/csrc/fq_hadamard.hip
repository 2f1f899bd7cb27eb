// fq_hadamard.hip — online Hadamard rotation over the last axis, n = K * P (P = 2^p >= 8):
//     y = hadK [K,K] @ FWHT_P( x.view(rows, K, P) ) * scale
//
// Replaces flatquant/hadamard_utils.py:132-141 / deploy/functional/online_trans.py:144-151 (third-party
// fast_hadamard_transform CUDA kernel + a batched cuBLAS matmul: two launches, two HBM round trips) and the
// pure-torch matmul_hadU (hadamard_utils.py:89-110: log2(P) full-tensor passes) with ONE launch that reads each
// row once and writes it once.
//
// One 4-wave workgroup per row, the row lives in LDS as fp32:
//   * pass 1 reads 16 bytes (8 fp16) per lane from HBM and does butterfly stages 0-2 in registers;
//   * further radix-8 passes (3 stages each, the last one 1-3) run LDS -> registers -> LDS;
//     stage order is stride 1, 2, 4, ... exactly as hadamard_utils.py:94-101 and the fast_hadamard_transform
//     kernel (in-thread, then across lanes, then across warps) — fp32 add/sub only, so results are
//     bit-identical to the oracle's fwht_f32;
//   * K == 1: the last pass scales, rounds to fp16 and stores;
//   * K > 1: the scaled fp16 values are laid out [p][k] in LDS and the K x K factor is applied by MFMA
//     (v_mfma_f32_32x32x16_f16: A = 32 positions p x 16 k, B = hadK^T fragments, fp32 accumulate — the
//     fp16-in/fp32-accumulate arithmetic of the reference's `hadK @ input` in fp16). The A rows are permuted so
//     each lane ends with 16 consecutive p of one output row k' -> 32-byte stores.
#include "fq_common.hpp"
#include <stdlib.h>

namespace {

__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <int W>  // radix 2^W butterfly, stages in ascending bit order
__device__ __forceinline__ void bfly(float (&v)[8]) {
#pragma unroll
    for (int b = 0; b < W; ++b) {
        const int s = 1 << b;
#pragma unroll
        for (int i = 0; i < (1 << W); ++i)
            if (!(i & s)) {
                const float a = v[i], c = v[i + s];
                v[i] = a + c;
                v[i + s] = a - c;
            }
    }
}

struct HadGeom {
    int n, K, P, log2P;
    int KP;       // K padded to a multiple of 16 (MFMA K-steps)
    int KT;       // 32-wide output tiles over k'
    int vpitch;   // fp16 elements per p-row of the [p][k] image (odd number of 16-byte chunks)
    int frag_off; // byte offset of the hadK fragment image in LDS
};

__global__ __launch_bounds__(256) void fq_hadamard_kernel(const f16* __restrict__ x, f16* __restrict__ y,
                                                          int64_t rows, HadGeom g, const f16* __restrict__ hadK,
                                                          float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* buf = reinterpret_cast<float*>(smem);
    f16* vT = reinterpret_cast<f16*>(smem);  // aliases buf once the butterflies are done
    uint4* kfrag = reinterpret_cast<uint4*>(smem + g.frag_off);
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = g.n, K = g.K, P = g.P, log2P = g.log2P;
    const int n8 = n >> 3;

    if (K > 1) {
        // B fragments of hadK^T: frag (s, kt): B[k = 16 s + 8 h + j][k' = 32 kt + c] = hadK[k'][k]
        const int nfr = (g.KP / 16) * g.KT * 64;
        for (int item = tid; item < nfr; item += 256) {
            const int f = item >> 6, ln = item & 63, fh = ln >> 5, fc = ln & 31;
            const int s = f / g.KT, kt = f - s * g.KT;
            const int kp = kt * 32 + fc;
            f16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = s * 16 + fh * 8 + j;
                v[j] = (k < K && kp < K) ? hadK[kp * K + k] : (f16)0.0f;
            }
            kfrag[item] = __builtin_bit_cast(uint4, v);
        }
    }

    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        __syncthreads();  // previous row's consumers are done with LDS
        // ---- pass 1: HBM -> registers, stages 0..2, -> LDS fp32 ----
        const uint4* xp = reinterpret_cast<const uint4*>(x + row * (int64_t)n);
        for (int q = tid; q < n8; q += 256) {
            const f16x8 hv = __builtin_bit_cast(f16x8, xp[q]);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)hv[e];
            bfly<3>(v);
            float4* bp = reinterpret_cast<float4*>(buf + q * 8);
            bp[0] = make_float4(v[0], v[1], v[2], v[3]);
            bp[1] = make_float4(v[4], v[5], v[6], v[7]);
        }
        // ---- middle passes: first the odd-width one (log2P - 3 not a multiple of 3), then radix 8, so that the
        //      LAST pass is radix 8 whenever log2P >= 6 (few items per thread to carry across the barrier) ----
        int b = 3;
        auto lds_pass = [&](int bb, int w) {
            const int S = 1 << bb, rad = 1 << w, items = n >> w;
            for (int id = tid; id < items; id += 256) {
                const int lo = id & (S - 1), hi = id >> bb;
                float* p = buf + ((hi << (bb + w)) + lo);
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (j < rad) ? p[j << bb] : 0.0f;
                if (w == 3) bfly<3>(v);
                else if (w == 2) bfly<2>(v);
                else bfly<1>(v);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < rad) p[j << bb] = v[j];
            }
        };
        {
            const int rem = (log2P - 3) % 3;
            if (log2P > 6 && rem) {
                __syncthreads();
                lds_pass(b, rem);
                b += rem;
            }
            for (; b + 3 < log2P; b += 3) {
                __syncthreads();
                lds_pass(b, 3);
            }
        }
        __syncthreads();
        // ---- last pass: W = log2P - b stages (3 when log2P >= 6), then scale + fp16 ----
        const int W = log2P - b;  // 1..3
        const int S = 1 << b;
        const int rad = 1 << W;
        const int items = n >> W;
        if (K == 1) {
            f16* yp = y + row * (int64_t)n;
            for (int id = tid; id < items; id += 256) {
                const int lo = id & (S - 1), hi = id >> b;
                const int base = (hi << (b + W)) + lo;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (j < rad) ? buf[base + (j << b)] : 0.0f;
                if (W == 3) bfly<3>(v);
                else if (W == 2) bfly<2>(v);
                else if (W == 1) bfly<1>(v);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < rad) yp[base + (j << b)] = fq_mul_to_f16(v[j], scale);
            }
            continue;
        }
        // K > 1: results go to the [p][k] fp16 image, which aliases buf -> keep them in registers across a barrier.
        constexpr int MAXI = 20;  // items per thread (launcher guarantees ceil(items / 256) <= MAXI)
        f16 res[MAXI][8];
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int id = tid + i * 256;
            if (id < items) {
                const int lo = id & (S - 1), hi = id >> b;
                const int base = (hi << (b + W)) + lo;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (j < rad) ? buf[base + (j << b)] : 0.0f;
                if (W == 3) bfly<3>(v);
                else if (W == 2) bfly<2>(v);
                else if (W == 1) bfly<1>(v);
#pragma unroll
                for (int j = 0; j < 8; ++j) res[i][j] = fq_mul_to_f16(v[j], scale);
            }
        }
        __syncthreads();  // all reads of buf are done: overwrite it with the [p][k] image
        for (int i = tid; i < P * (g.vpitch / 8); i += 256) reinterpret_cast<uint4*>(vT)[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int id = tid + i * 256;
            if (id < items) {
                const int lo = id & (S - 1), hi = id >> b;
                const int base = (hi << (b + W)) + lo;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < rad) {
                        const int e = base + (j << b);
                        const int k = e >> log2P, pp = e & (P - 1);
                        vT[pp * g.vpitch + k] = res[i][j];
                    }
            }
        }
        __syncthreads();
        // ---- K x K factor on the matrix cores: out^T[p][k'] = sum_k vT[p][k] hadK[k'][k] ----
        f16* yp = y + row * (int64_t)n;
        const int ptiles = P >> 5, ksteps = g.KP >> 4;
        for (int t = wave; t < ptiles * g.KT; t += 4) {
            const int pt = t / g.KT, kt = t - pt * g.KT;
            // A row supplied by this lane: p = 32 pt + pmap(c), pmap -> lane (h, .) ends with p = 32 pt + 16 h + reg
            const int prow = pt * 32 + ((c >> 2) & 1) * 16 + (c & 3) + 4 * (c >> 3);
            const uint4* ap = reinterpret_cast<const uint4*>(vT + prow * g.vpitch) + h;
            f32x16 acc = {0};
            for (int s = 0; s < ksteps; ++s)
                acc = mfma32(__builtin_bit_cast(f16x8, ap[s * 2]), __builtin_bit_cast(f16x8, kfrag[(s * g.KT + kt) * 64 + lane]),
                             acc);
            const int kp = kt * 32 + c;
            if (kp < K) {
                f16x8 v0, v1;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v0[e] = (f16)acc[e];
                    v1[e] = (f16)acc[8 + e];
                }
                uint4* op = reinterpret_cast<uint4*>(yp + (int64_t)kp * P + pt * 32 + h * 16);
                op[0] = __builtin_bit_cast(uint4, v0);
                op[1] = __builtin_bit_cast(uint4, v1);
            }
        }
    }
}

}  // namespace

int fq_launch_hadamard_reg(const f16* x, f16* y, int64_t rows, int n, int K, const f16* hadK, float scale, int n_cu,
                           hipStream_t stream);  // fq_hadamard_reg.hip

int fq_launch_hadamard(const f16* x, f16* y, int64_t rows, int n, int K, const f16* hadK, float scale, int n_cu,
                       hipStream_t stream) {
    {  // register FWHT where it applies (P = 64 ... 512 * 2^q); this file is the general case
        const int rc = fq_launch_hadamard_reg(x, y, rows, n, K, hadK, scale, n_cu, stream);
        if (rc != -1000) return rc;
    }
    HadGeom g;
    g.n = n;
    g.K = K;
    g.P = n / K;
    if (g.P < 32 || (n & 7) || n > 40960 || K > 192) return -1000;
    g.log2P = 0;
    while ((1 << g.log2P) < g.P) ++g.log2P;
    g.KP = (K + 15) / 16 * 16;
    g.KT = (K + 31) / 32;
    g.vpitch = (((g.KP / 8) | 1)) * 8;
    size_t main_bytes = (size_t)n * 4;
    if (K > 1) {
        const size_t vt = (size_t)g.P * g.vpitch * 2;
        if (vt > main_bytes) main_bytes = vt;
        // items held in registers across the barrier: n >> W per 256 threads, W >= 1 when log2P > 3
        const int W = g.log2P >= 6 ? 3 : g.log2P - 3;  // width of the last butterfly pass (see kernel)
        if (((n >> W) + 255) / 256 > 20) return -1000;
    }
    main_bytes = (main_bytes + 15) & ~(size_t)15;
    g.frag_off = (int)main_bytes;
    const size_t lds = main_bytes + (K > 1 ? (size_t)(g.KP / 16) * g.KT * 1024 : 0);
    if (lds > 160 * 1024) return -1000;
    FQ_RAISE_LDS_CAP(fq_hadamard_kernel, 160 * 1024);
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    int64_t blocks = (int64_t)n_cu * per_cu;
    if (blocks > rows) blocks = rows;
    hipLaunchKernelGGL(fq_hadamard_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, x, y, rows, g, hadK, scale);
    return (int)hipGetLastError();
}
