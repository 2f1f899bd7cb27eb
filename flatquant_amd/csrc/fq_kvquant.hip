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
#include "fq_common.hpp"

namespace {

__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f16 xchg32(f16 v, int lane) {  // value of lane ^ 32
    const int iv = (int)__builtin_bit_cast(unsigned short, v);
    const int r = __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, iv);
    return __builtin_bit_cast(f16, (unsigned short)r);
}

struct KvParams {
    f16 scale, zero;
};

// kv_cache.py:13-46 for one row whose extrema are (xmax, xmin)
template <bool LAC>
__device__ __forceinline__ KvParams kv_params(f16 xmax, f16 xmin, f16 cmax, f16 cmin) {
    KvParams p;
    if (LAC) {
        xmax = xmax > (f16)0 ? xmax : (f16)0;
        xmin = xmin < (f16)0 ? xmin : (f16)0;
        xmax = xmax * cmax;
        xmin = xmin * cmin;
        if (xmin == (f16)0 && xmax == (f16)0) {
            xmin = (f16)-1.0f;
            xmax = (f16)1.0f;
        }
        const f16 d = xmax - xmin;
        p.scale = d / (f16)15.0f;
        const f16 nx = (f16)-1.0f * xmin;
        p.zero = (f16)__builtin_rintf((float)(f16)(nx / p.scale));
    } else {
        f16 d = xmax - xmin;
        const f16 floor_ = (f16)1e-5f;  // .clamp(min=1e-5) on an fp16 tensor
        d = d > floor_ ? d : floor_;
        p.scale = d / (f16)15.0f;
        p.zero = -xmin;
    }
    return p;
}

template <bool LAC>
__device__ __forceinline__ unsigned kv_q1(f16 x, KvParams p) {
    float r;
    if (LAC) {
        const f16 t = (f16)__builtin_rintf((float)(f16)(x / p.scale));
        r = (float)(f16)(t + p.zero);
    } else {
        const f16 t = x + p.zero;
        r = __builtin_rintf((float)(f16)(t / p.scale));
    }
    r = __builtin_amdgcn_fmed3f(r, 0.0f, 15.0f);
    return (unsigned)(int)r;
}

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
    if (TRANS) {
        for (int item = tid; item < KS * NTL * 64; item += 256) {
            const int f = item >> 6, ln = item & 63, fh = ln >> 5, fc = ln & 31;
            const int s = f / NTL, nt = f - s * NTL;
            const int n = nt * 32 + 16 * ((fc >> 2) & 1) + 4 * (fc >> 3) + (fc & 3);  // output column of A row fc
            f16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = T[(s * 16 + fh * 8 + j) * HD + n];
            tfrag[item] = __builtin_bit_cast(uint4, v);
        }
        __syncthreads();
    }
    const f16 cmax = (f16)clip_max, cmin = (f16)clip_min;
    const int64_t n_tiles = (rows + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + (tid >> 6); tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t row = tile * 32 + c;
        const bool ok = row < rows;
        const int64_t lrow = ok ? row : rows - 1;
        const uint4* xp = reinterpret_cast<const uint4*>(x + lrow * HD);
        f16x8 xf[KS];  // chunk 2s + h of the row (8 consecutive columns each)
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[s] = __builtin_bit_cast(f16x8, xp[2 * s + h]);

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

        f16 mx = v[0][0], mn = v[0][0];
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                mx = v[nt][r] > mx ? v[nt][r] : mx;
                mn = v[nt][r] < mn ? v[nt][r] : mn;
            }
        const f16 omx = xchg32(mx, lane), omn = xchg32(mn, lane);
        mx = omx > mx ? omx : mx;
        mn = omn < mn ? omn : mn;
        const KvParams p = kv_params<LAC>(mx, mn, cmax, cmin);

        // destination(s) of this lane's row: the dense row, or `group` rows of the paged cache
        uint8_t* qdst[4];
        unsigned* pdst[4];
        int ndst = 1;
        if (io.data == nullptr) {
            qdst[0] = q_out + row * (HD / 2);
            pdst[0] = reinterpret_cast<unsigned*>(param + row * 2);
        } else {
            const int64_t t = lrow / io.src_heads;
            const int hs = (int)(lrow - t * io.src_heads);
            const int b = (int)(t / io.added), j = (int)(t - (int64_t)b * io.added);
            const int64_t seq_len = (int64_t)(io.indptr[b + 1] - io.indptr[b] - 1) * io.page_size + io.last[b];
            const int64_t pos = seq_len - io.added + j;
            const size_t page = (size_t)io.indices[io.indptr[b] + pos / io.page_size];
            const size_t entry = (size_t)(pos % io.page_size);
            ndst = io.group;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const size_t e = (((page * io.num_layers + io.layer_idx) * 2 + which) * io.num_heads + (size_t)hs * io.group + (g < ndst ? g : 0)) *
                                     io.page_size + entry;
                qdst[g] = io.data + e * (HD / 2);
                pdst[g] = reinterpret_cast<unsigned*>(io.pparam) + e;
            }
        }
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            unsigned w0 = 0, w1 = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                w0 |= kv_q1<LAC>(v[nt][e], p) << (4 * e);
                w1 |= kv_q1<LAC>(v[nt][8 + e], p) << (4 * e);
            }
            if (ok) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
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
            for (int g = 0; g < 4; ++g)
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
    if (blocks > (int64_t)n_cu * 2) blocks = (int64_t)n_cu * 2;
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

// K (transformed when T != nullptr) and V of one layer step, quantised and written straight into the paged cache.
int fq_launch_kv_quant_append(const f16* k, const f16* v, const f16* T, int64_t tokens, int src_heads, int hd, const float* clip4,
                              bool lac, void* kv_data, void* kv_param, const int* indptr, const int* indices, const int* last,
                              int num_layers, int layer_idx, int num_heads, int page_size, int added, int group, int n_cu,
                              hipStream_t stream) {
    if (group < 1 || group > 4 || src_heads * group != num_heads || added < 1) return -1000;
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
