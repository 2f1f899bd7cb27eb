// fq_kvquant.hip — the KV-cache side of the online transforms: per (token, head) row of head_dim values,
//   K:  y = fp16( x . trans_matrix_k )            (deploy/transformers/kv_cache.py:268, torch.matmul on fp16)
//   K, V: asymmetric INT4 quantisation + pack      (kv_cache.py:11-51, asym_quantize_and_pack_i4, fp16 arithmetic)
// in one launch: the reference runs the GEMM and then ~15 element-wise / reduction launches over the same tensor.
//
// One wave per 32 rows. The product is formed TRANSPOSED on the matrix cores (A = fragments of trans^T from LDS with
// the row permutation of the other kernels, B = the 32 rows straight from HBM as 16-byte fragment loads), so that lane
// (h, c) ends with half h of row c: per n-tile 16 consecutive columns -> the row extrema are an in-lane reduction plus
// ONE exchange with lane ^ 32, and the packed nibbles leave as 8-byte stores. Every quantiser step rounds to fp16
// exactly where the torch expression does (native _Float16 ops; the fp16 division is the correctly rounded one,
// fq_common.hpp). V (no transform) takes the same path with the loaded fragments as values.
#include "fq_kv_common.hpp"

namespace {
using namespace fqkv;

// Inputs / outputs of one launch. which = blockIdx.y: 0 = keys (transformed when TRANS), 1 = values (never transformed;
// present when x[1] != nullptr: K and V of a layer step in one launch). Dense destinations q / param / y, or — data != nullptr —
// straight into the paged cache (the scatter of fq_kv_append_kernel, fq_kvcache.hip: every request appends `added` tokens
// at the end of its current length, each source head feeds `group` cache heads).
struct KvIO {
    const f16* x[2];
    uint8_t* q[2];
    f16* param[2];
    f16* y;
    float cmax[2], cmin[2];
    uint8_t* data;
    f16* pparam;
    const int* indptr;
    const int* indices;
    const int* last;
    int num_layers, layer_idx, num_heads, page_size, added, group, src_heads;
};

template <int HD, bool TRANS, bool LAC>
__global__ __launch_bounds__(256) void fq_kv_quant_kernel(KvIO io, const f16* __restrict__ T, int64_t rows) {
    const int which = blockIdx.y;
    const bool do_trans = TRANS && which == 0;
    const f16* __restrict__ x = io.x[which];
    uint8_t* __restrict__ q_out = io.q[which];
    f16* __restrict__ param = io.param[which];
    f16* __restrict__ y_out = which == 0 ? io.y : nullptr;
    const float clip_max = io.cmax[which], clip_min = io.cmin[which];
    constexpr int KS = HD / 16, NTL = HD / 32;
    __shared__ __attribute__((aligned(16))) uint4 tfrag[TRANS ? KS * NTL * 64 : 1];  // [(s * NTL + nt)][lane]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    __shared__ __attribute__((aligned(16))) unsigned short traw[TRANS ? (HD / 2) * (HD + 2) : 2];
    const f16 cmax = (f16)clip_max, cmin = (f16)clip_min;
    const int64_t n_tiles = (rows + 31) / 32;
    // the next tile's row is requested before this one is worked on: a wave's loads overlap its own arithmetic (all waves
    // of a launch start together, so without this the launch is a load phase followed by an arithmetic phase)
    const int64_t tstep = (int64_t)gridDim.x * 4;
    auto fetch = [&](int64_t t, f16x8 (&dst)[KS]) {
        int64_t r = t * 32 + c;
        r = r < rows ? r : rows - 1;
        const uint4* xp = reinterpret_cast<const uint4*>(x + r * HD);
#pragma unroll
        for (int s = 0; s < KS; ++s) dst[s] = __builtin_bit_cast(f16x8, xp[2 * s + h]);
    };
    const bool small = rows <= 0x7FFFFFFF;   // 32-bit index arithmetic (a 64-bit division is ~100 instructions, four of them per row)
    // destination(s) of a lane's row: the dense row, or `group` rows of the paged cache (two dependent page-table round trips)
    struct Dest {       // group copy g lives g * page_size entries further (cache head hs * group + g of the same page and entry)
        uint8_t* q;
        unsigned* p;
        int n;
    };
    auto locate = [&](int64_t tile, Dest& d) {
        const int64_t row = tile * 32 + c;
        const int64_t lrow = row < rows ? row : rows - 1;
        d.n = 1;
        if (io.data == nullptr) {
            d.q = q_out + row * (HD / 2);
            d.p = reinterpret_cast<unsigned*>(param + row * 2);
            return;
        }
        int64_t t;
        int hs, b, j;
        if (small) {
            const unsigned lr = (unsigned)lrow, t32 = lr / (unsigned)io.src_heads, b32 = t32 / (unsigned)io.added;
            t = t32, hs = (int)(lr - t32 * (unsigned)io.src_heads), b = (int)b32, j = (int)(t32 - b32 * (unsigned)io.added);
        } else {
            t = lrow / io.src_heads, hs = (int)(lrow - t * io.src_heads);
            b = (int)(t / io.added), j = (int)(t - (int64_t)b * io.added);
        }
        const int pgb = io.indptr[b];
        const int64_t seq_len = (int64_t)(io.indptr[b + 1] - pgb - 1) * io.page_size + io.last[b];
        const int64_t pos = seq_len - io.added + j;
        int64_t pq;
        int entry_i;
        if ((uint64_t)pos <= 0x7FFFFFFFu) {
            const unsigned p32 = (unsigned)pos, q32 = p32 / (unsigned)io.page_size;
            pq = q32, entry_i = (int)(p32 - q32 * (unsigned)io.page_size);
        } else {
            pq = pos / io.page_size, entry_i = (int)(pos - pq * io.page_size);
        }
        const size_t page = (size_t)io.indices[pgb + pq];
        const size_t entry = (size_t)entry_i;
        d.n = io.group;
        const size_t e = (((page * io.num_layers + io.layer_idx) * 2 + which) * io.num_heads + (size_t)hs * io.group) * io.page_size + entry;
        d.q = io.data + e * (HD / 2);
        d.p = reinterpret_cast<unsigned*>(io.pparam) + e;
    };
    // A decode step is ONE tile per wave and two workgroups per launch: every dependent round trip of this kernel is exposed there. So
    // the first rows and their destinations are requested before the matrix is staged (the values' workgroups, which == 1, never use it),
    // and the next tile's during the current one.
    f16x8 xn[KS];
    Dest dn;
    {
        const int64_t t0 = (int64_t)blockIdx.x * 4 + (tid >> 6);
        if (t0 < n_tiles) {
            fetch(t0, xn);
            locate(t0, dn);
        }
    }
    if (TRANS && do_trans) kv_stage_tfrag<HD, f16>(T, tfrag, traw, tid);
    for (int64_t tile = (int64_t)blockIdx.x * 4 + (tid >> 6); tile < n_tiles; tile += tstep) {
        const int64_t row = tile * 32 + c;
        const bool ok = row < rows;
        uint8_t* qdst[8];      // (up to eight copies per source head: Llama-2/3-70B have 64 query heads on 8 KV heads)
        unsigned* pdst[8];
        const int ndst = dn.n;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            qdst[g] = dn.q + (size_t)g * io.page_size * (HD / 2);
            pdst[g] = dn.p + (size_t)g * io.page_size;
        }
        f16x8 xf[KS];  // chunk 2s + h of the row (8 consecutive columns each)
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[s] = xn[s];
        if (tile + tstep < n_tiles) {
            fetch(tile + tstep, xn);
            locate(tile + tstep, dn);
        }

        f16 v[NTL][16];  // TRANS: columns nt*32 + 16h + r;  else: v[s/2][(s&1)*8 + j] = column (2s + h)*8 + j
        if (TRANS && do_trans) {
            int foff = lane;
            asm volatile("" : "+v"(foff));
            const uint4* tf = tfrag + foff;
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                f32x16 acc = {0};
#pragma unroll
                for (int s = 0; s < KS; ++s) acc = mfma32(__builtin_bit_cast(f16x8, tf[(s * NTL + nt) * 64]), xf[s], acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) v[nt][r] = (f16)acc[r];
            }
            if (y_out != nullptr && ok) {
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) {
                    f16x8 a, b;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        a[e] = v[nt][e];
                        b[e] = v[nt][8 + e];
                    }
                    uint4* yp = reinterpret_cast<uint4*>(y_out + row * HD + nt * 32 + 16 * h);
                    yp[0] = __builtin_bit_cast(uint4, a);
                    yp[1] = __builtin_bit_cast(uint4, b);
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[s >> 1][(s & 1) * 8 + j] = xf[s][j];
        }

        f16x2 pmx = {v[0][0], v[0][1]}, pmn = pmx;
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f16x2 pr = {v[nt][r], v[nt][r + 1]};
                pmx = fq_pk_max(pmx, pr);
                pmn = fq_pk_min(pmn, pr);
            }
        f16 mx = pmx[0] > pmx[1] ? pmx[0] : pmx[1], mn = pmn[0] < pmn[1] ? pmn[0] : pmn[1];
        const f16 omx = xchg32(mx, lane), omn = xchg32(mn, lane);
        mx = omx > mx ? omx : mx;
        mn = omn < mn ? omn : mn;
        const KvParams p = kv_params<LAC>(mx, mn, cmax, cmin);

        const float sc = (float)p.scale, rc = fq_fast_inv(sc);
        const uint32_t zero2 = __builtin_bit_cast(uint32_t, f16x2{p.zero, p.zero});
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            uint32_t pr[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pr[j] = __builtin_bit_cast(uint32_t, f16x2{v[nt][2 * j], v[nt][2 * j + 1]});
            const unsigned w0 = kv_q8<LAC>(pr[0], pr[1], pr[2], pr[3], rc, sc, zero2);
            const unsigned w1 = kv_q8<LAC>(pr[4], pr[5], pr[6], pr[7], rc, sc, zero2);
            if (ok) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    if (g >= ndst) break;
                    if (TRANS && do_trans) {
                        *reinterpret_cast<uint2*>(qdst[g] + nt * 16 + 8 * h) = make_uint2(w0, w1);
                    } else {  // chunk s = 2nt (w0), 2nt + 1 (w1): bytes (2s + h) * 4
                        *reinterpret_cast<unsigned*>(qdst[g] + (4 * nt + h) * 4) = w0;
                        *reinterpret_cast<unsigned*>(qdst[g] + (4 * nt + 2 + h) * 4) = w1;
                    }
                }
            }
        }
        if (ok && h == 0) {
            const unsigned short s16 = __builtin_bit_cast(unsigned short, p.scale), z16 = __builtin_bit_cast(unsigned short, p.zero);
#pragma unroll
            for (int g = 0; g < 8; ++g)
                if (g < ndst) *pdst[g] = (unsigned)s16 | ((unsigned)z16 << 16);
        }
    }
}

// kv_cache.py:54-61: 8 columns per thread
template <bool LAC>
__global__ __launch_bounds__(256) void fq_kv_dequant_kernel(const uint8_t* __restrict__ q, const f16* __restrict__ param,
                                                            int64_t rows, int hd, f16* __restrict__ y) {
    const int cpr = hd >> 3;
    const int64_t total = rows * cpr, step = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
        const int64_t row = i / cpr;
        const unsigned w = reinterpret_cast<const unsigned*>(q)[i];
        const f16 s = param[row * 2], z = param[row * 2 + 1];
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const f16 qv = (f16)(float)((w >> (4 * e)) & 15u);
            if (LAC) {
                const f16 t = qv - z;
                o[e] = s * t;
            } else {
                const f16 t = qv * s;
                o[e] = t - z;
            }
        }
        reinterpret_cast<uint4*>(y)[i] = __builtin_bit_cast(uint4, o);
    }
}

template <int HD>
int launch_kv(const KvIO& io, const f16* T, int64_t rows, bool lac, int n_cu, hipStream_t stream) {
    int64_t blocks = (rows + 127) / 128;
#ifndef FQ_KV_OCC
#define FQ_KV_OCC 2
#endif
    if (blocks > (int64_t)n_cu * FQ_KV_OCC) blocks = (int64_t)n_cu * FQ_KV_OCC;
    if (blocks < 1) blocks = 1;
    const dim3 grid((unsigned)blocks, io.x[1] != nullptr ? 2u : 1u);
#define FQ_KV(TR, LC) hipLaunchKernelGGL((fq_kv_quant_kernel<HD, TR, LC>), grid, dim3(256), 0, stream, io, T, rows)
    if (T != nullptr) {
        if (lac) FQ_KV(true, true);
        else FQ_KV(true, false);
    } else {
        if (lac) FQ_KV(false, true);
        else FQ_KV(false, false);
    }
#undef FQ_KV
    return (int)hipGetLastError();
}

}  // namespace

static KvIO kv_io_dense(const f16* x, float cmax, float cmin, uint8_t* q, f16* param, f16* y) {
    KvIO io = {};
    io.x[0] = x;
    io.q[0] = q;
    io.param[0] = param;
    io.y = y;
    io.cmax[0] = cmax;
    io.cmin[0] = cmin;
    return io;
}

int fq_launch_kv_quant(const f16* x, const f16* T, int64_t rows, int hd, float cmax, float cmin, bool lac, uint8_t* q,
                       f16* param, f16* y, int n_cu, hipStream_t stream) {
    const KvIO io = kv_io_dense(x, cmax, cmin, q, param, y);
    if (hd == 128) return launch_kv<128>(io, T, rows, lac, n_cu, stream);
    if (hd == 64) return launch_kv<64>(io, T, rows, lac, n_cu, stream);
    return -1000;
}

// The K transform as the fragment image fq_kv_quant_kernel builds in LDS at the start of every launch ([(s * NTL + nt)][lane] uint4, 32 KB at
// head_dim 128): written once per deployed layer for the decode launch that quantises the step's own row (fq_kvcache.hip, KvNew) — there the
// image's fragments are single coalesced 16-byte loads in front of eight MFMAs instead of a staging pass with two barriers per workgroup.
template <int HD>
__global__ __launch_bounds__(256) void fq_kv_timage_kernel(const f16* __restrict__ T, uint4* __restrict__ img) {
    constexpr int KS = HD / 16, NTL = HD / 32;
    __shared__ __attribute__((aligned(16))) uint4 tfrag[KS * NTL * 64];
    __shared__ __attribute__((aligned(16))) unsigned short traw[(HD / 2) * (HD + 2)];
    kv_stage_tfrag<HD, f16>(T, tfrag, traw, threadIdx.x);
    for (int i = threadIdx.x; i < KS * NTL * 64; i += 256) img[i] = tfrag[i];
}

int64_t fq_kv_timage_bytes(int hd) { return (hd == 128 || hd == 64) ? (int64_t)(hd / 16) * (hd / 32) * 64 * 16 : -1; }

int fq_launch_kv_timage(const f16* T, int hd, void* img, hipStream_t stream) {
    if (hd == 128) hipLaunchKernelGGL(fq_kv_timage_kernel<128>, dim3(1), dim3(256), 0, stream, T, reinterpret_cast<uint4*>(img));
    else if (hd == 64) hipLaunchKernelGGL(fq_kv_timage_kernel<64>, dim3(1), dim3(256), 0, stream, T, reinterpret_cast<uint4*>(img));
    else return -1000;
    return (int)hipGetLastError();
}

// K (transformed when T != nullptr) and V of one layer step, quantised and written straight into the paged cache.
int fq_launch_kv_quant_append(const f16* k, const f16* v, const f16* T, int64_t tokens, int src_heads, int hd, const float* clip4,
                              bool lac, void* kv_data, void* kv_param, const int* indptr, const int* indices, const int* last,
                              int num_layers, int layer_idx, int num_heads, int page_size, int added, int group, int n_cu,
                              hipStream_t stream) {
    if (group < 1 || group > 8 || src_heads * group != num_heads || added < 1) return -1000;
    KvIO io = {};
    io.x[0] = k;
    io.x[1] = v;
    io.cmax[0] = clip4[0];
    io.cmin[0] = clip4[1];
    io.cmax[1] = clip4[2];
    io.cmin[1] = clip4[3];
    io.data = (uint8_t*)kv_data;
    io.pparam = (f16*)kv_param;
    io.indptr = indptr;
    io.indices = indices;
    io.last = last;
    io.num_layers = num_layers;
    io.layer_idx = layer_idx;
    io.num_heads = num_heads;
    io.page_size = page_size;
    io.added = added;
    io.group = group;
    io.src_heads = src_heads;
    const int64_t rows = tokens * src_heads;
    if (hd == 128) return launch_kv<128>(io, T, rows, lac, n_cu, stream);
    if (hd == 64) return launch_kv<64>(io, T, rows, lac, n_cu, stream);
    return -1000;
}

// ---- the single-matrix transform alone: y = x.reshape(-1, n) @ matrix in the activation's dtype (round 4) ----
// {SVD,Inv}SingleTransMatrix.forward (flatquant/trans_utils.py:21-25, 136-151) at n = head_dim: the fake-quant eval path applies
// kcache_trans(q, inv_t=True), kcache_trans(k) and vcache_trans(v) to [.., heads, head_dim] activations on every forward
// (flatquant/model_tools/llama_utils.py:181-199) — rows = (token, head), n = 64 / 128, a torch.matmul on fp16 / bf16 in the reference.
// Same scheme as fq_kv_quant_kernel's transform (the product formed transposed on the matrix pipe, A = fragments of matrix^T in LDS,
// B = 32 rows straight from HBM, fp32 accumulation, one rounding to the dtype), without the quantiser; fp16 and bf16.
template <int HD, typename T>
__global__ __launch_bounds__(256) void fq_rowmm_kernel(const T* __restrict__ x, const T* __restrict__ Tm, T* __restrict__ y, int64_t rows) {
    typedef typename FqVec<T>::x8 X8;
    constexpr int KS = HD / 16, NTL = HD / 32;
    __shared__ __attribute__((aligned(16))) uint4 tfrag[KS * NTL * 64];  // [(s * NTL + nt)][lane]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    __shared__ __attribute__((aligned(16))) unsigned short traw[(HD / 2) * (HD + 2)];
    kv_stage_tfrag<HD, T>(Tm, tfrag, traw, tid);
    const int64_t n_tiles = (rows + 31) / 32;
    const int64_t tstep = (int64_t)gridDim.x * 4;
    auto fetch = [&](int64_t t, X8 (&dst)[KS]) {
        int64_t r = t * 32 + c;
        r = r < rows ? r : rows - 1;
        const uint4* xp = reinterpret_cast<const uint4*>(x + r * HD);
#pragma unroll
        for (int s = 0; s < KS; ++s) dst[s] = __builtin_bit_cast(X8, xp[2 * s + h]);
    };
    X8 xn[KS];
    {
        const int64_t t0 = (int64_t)blockIdx.x * 4 + (tid >> 6);
        if (t0 < n_tiles) fetch(t0, xn);
    }
    for (int64_t tile = (int64_t)blockIdx.x * 4 + (tid >> 6); tile < n_tiles; tile += tstep) {
        const int64_t row = tile * 32 + c;
        X8 xf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[s] = xn[s];
        if (tile + tstep < n_tiles) fetch(tile + tstep, xn);
        int foff = lane;
        asm volatile("" : "+v"(foff));
        const uint4* tf = tfrag + foff;
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            f32x16 acc = {0};
#pragma unroll
            for (int s = 0; s < KS; ++s) acc = fq_mfma32<T>(__builtin_bit_cast(X8, tf[(s * NTL + nt) * 64]), xf[s], acc);
            X8 a, b;   // columns nt*32 + 16h + r of row c
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a[e] = (T)acc[e];
                b[e] = (T)acc[8 + e];
            }
            if (row < rows) {
                uint4* yp = reinterpret_cast<uint4*>(y + row * HD + nt * 32 + 16 * h);
                yp[0] = __builtin_bit_cast(uint4, a);
                yp[1] = __builtin_bit_cast(uint4, b);
            }
        }
    }
}

template <typename T>
static int launch_rowmm(const T* x, const T* Tm, T* y, int64_t rows, int n, int n_cu, hipStream_t stream) {
    int64_t blocks = (rows + 127) / 128;
    if (blocks > (int64_t)n_cu * 8) blocks = (int64_t)n_cu * 8;
    if (blocks < 1) blocks = 1;
    if (n == 128) hipLaunchKernelGGL((fq_rowmm_kernel<128, T>), dim3((unsigned)blocks), dim3(256), 0, stream, x, Tm, y, rows);
    else if (n == 64) hipLaunchKernelGGL((fq_rowmm_kernel<64, T>), dim3((unsigned)blocks), dim3(256), 0, stream, x, Tm, y, rows);
    else return -1000;
    return (int)hipGetLastError();
}
int fq_launch_rowmm(int bf16_dtype, const void* x, const void* Tm, void* y, int64_t rows, int n, int n_cu, hipStream_t stream) {
    return bf16_dtype ? launch_rowmm<bf16>((const bf16*)x, (const bf16*)Tm, (bf16*)y, rows, n, n_cu, stream)
                      : launch_rowmm<f16>((const f16*)x, (const f16*)Tm, (f16*)y, rows, n, n_cu, stream);
}

int fq_launch_kv_dequant(const uint8_t* q, const f16* param, int64_t rows, int hd, bool lac, f16* y, int n_cu,
                         hipStream_t stream) {
    if (hd & 7) return -1000;
    int64_t blocks = (rows * (hd >> 3) + 255) / 256;
    if (blocks > (int64_t)n_cu * 8) blocks = (int64_t)n_cu * 8;
    if (blocks < 1) blocks = 1;
    if (lac) hipLaunchKernelGGL(fq_kv_dequant_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, q, param, rows, hd, y);
    else hipLaunchKernelGGL(fq_kv_dequant_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, q, param, rows, hd, y);
    return (int)hipGetLastError();
}
