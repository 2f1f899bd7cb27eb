// fq_kvcache.hip — the paged INT4 KV cache behind MultiLayerPagedKVCache4Bit (deploy/transformers/kv_cache.py:166-359):
//   append  (init_kv_i4 / append_kv_i4  -> kernels/flashinfer.cu:47-96 -> include/flashinfer/page.cuh:118-214)
//   decode  (batch_decode_i4            -> kernels/flashinfer.cu:9-45  -> include/flashinfer/decode.cuh:492-683)
// Layouts (page.cuh:75-103): kv_data  [pages, layers, 2 (k, v), heads, page_size, head_dim / 2] bytes (two INT4 per byte,
// low nibble = even feature), kv_param [pages, layers, 2, heads, page_size] half2 = (scale, zero). A request b owns the
// pages indices[indptr[b] .. indptr[b+1]); its length is (n_pages - 1) * page_size + last_page_offset[b].
//
// Decode attention, one query token per request (decode.cuh:492-683 with rotary_mode none): for every (request, head)
//   k_i = float(n) * scale_i - zero_i  (quantization.cuh:58-80),  x_i = (q . k_i) / sqrt(head_dim),
//   o = sum_i softmax(x)_i * v_i, fp32 throughout, fp16 out.
// Here: one 4-wave workgroup per (request, head). A cached row is 64 bytes; a QUAD of lanes owns one row per step (16
// bytes = 32 features per lane, a wave reads 16 consecutive rows = 1 KB per instruction) and runs its OWN online softmax
// over the rows it sees — no cross-lane traffic in the loop beyond the 4-lane dot-product reduction. The 64 partial
// states (m, d, o[head_dim]) are merged once at the end through LDS (the standard max-rescaled merge, state.cuh).
// The dequantisation is folded into the dot product: q . k = scale * sum(q_j n_j) - zero * sum(q_j).
#include "fq_common.hpp"

namespace {

struct PagedKv {
    uint8_t* data;
    f16* param;            // half2 per entry
    const int* indptr;
    const int* indices;
    const int* last_page_offset;
    int num_layers, layer_idx, num_heads, page_size, head_dim, batch_size;
};

__device__ __forceinline__ size_t k_entry(const PagedKv& p, size_t page, size_t head, size_t entry) {
    return ((page * p.num_layers + p.layer_idx) * 2 * p.num_heads + head) * p.page_size + entry;
}
__device__ __forceinline__ size_t v_entry(const PagedKv& p, size_t page, size_t head, size_t entry) {
    return (((page * p.num_layers + p.layer_idx) * 2 + 1) * p.num_heads + head) * p.page_size + entry;
}

// page.cuh:118-214. seqlen_indptr == nullptr: one token per request (append_kv_i4), else request b appends
// seqlen_indptr[b+1] - seqlen_indptr[b] tokens that END at its current length (init_kv_i4).
// One thread per 16 bytes of one (token, head) row of k and of v. F16: the fp16 configuration of the cache
// (disable_quant=True, init_kv_f16 / append_kv_f16): rows of head_dim fp16 values instead of head_dim / 2 bytes.
template <bool F16>
__global__ __launch_bounds__(256) void fq_kv_append_kernel(PagedKv p, const uint8_t* __restrict__ key,
                                                           const uint8_t* __restrict__ value, const f16* __restrict__ kparam,
                                                           const f16* __restrict__ vparam, const int* __restrict__ seqlen_indptr,
                                                           int64_t total_tokens, int group) {
    const int row_bytes = F16 ? p.head_dim * 2 : p.head_dim / 2;
    const int cpr = row_bytes / 16;  // 16-byte chunks per cached row
    const int64_t items = total_tokens * p.num_heads * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)(i % cpr);
        const int64_t th = i / cpr;
        const int head = (int)(th % p.num_heads);
        const int64_t tok = th / p.num_heads;
        int b;
        int64_t j, n_app;
        if (seqlen_indptr == nullptr) {
            b = (int)tok;
            j = 0;
            n_app = 1;
        } else {
            b = 0;
            while (b + 1 < p.batch_size && seqlen_indptr[b + 1] <= tok) ++b;
            j = tok - seqlen_indptr[b];
            n_app = seqlen_indptr[b + 1] - seqlen_indptr[b];
        }
        const int64_t seq_len = (int64_t)(p.indptr[b + 1] - p.indptr[b] - 1) * p.page_size + p.last_page_offset[b];
        const int64_t pos = seq_len - n_app + j;
        const size_t page = (size_t)p.indices[p.indptr[b] + pos / p.page_size];
        const size_t entry = (size_t)(pos % p.page_size);
        const size_t ke = k_entry(p, page, head, entry), ve = v_entry(p, page, head, entry);
        // grouped-query attention (kv_cache.py:286-296): the inputs hold num_heads / group heads, cache head h copies head h / group
        const size_t shead = (size_t)tok * (p.num_heads / group) + head / group;
        const size_t src = shead * row_bytes + (size_t)ch * 16;
        *reinterpret_cast<uint4*>(p.data + ke * row_bytes + ch * 16) = *reinterpret_cast<const uint4*>(key + src);
        *reinterpret_cast<uint4*>(p.data + ve * row_bytes + ch * 16) = *reinterpret_cast<const uint4*>(value + src);
        if (ch == 0) {
            reinterpret_cast<uint32_t*>(p.param)[ke] = reinterpret_cast<const uint32_t*>(kparam)[shead];
            reinterpret_cast<uint32_t*>(p.param)[ve] = reinterpret_cast<const uint32_t*>(vparam)[shead];
        }
    }
}

// F16: the fp16 configuration of the cache (batch_decode_f16): a cached row is head_dim fp16 values, no (scale, zero); a lane
// still owns 32 features of a row (64 bytes: four 16-byte loads).
template <int HD, int NW, bool F16 = false>  // NW waves per workgroup: 4, or 8 when there are too few (request, head) pairs to fill the chip
__global__ __launch_bounds__(NW * 64, 2) void fq_kv_decode_kernel(f16* __restrict__ o, const f16* __restrict__ q, PagedKv p,
                                                               const f16* __restrict__ qt, int transpose_out) {
    constexpr int QL = HD / 32;        // lanes per cached row (16 bytes = 32 features each): 4 for head_dim 128
    constexpr int RPW = 64 / QL;       // rows per wave and step
    constexpr int NS = NW * RPW;       // partial softmax states per workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char kv_smem[];
    float (*s_o)[HD + 1] = reinterpret_cast<float (*)[HD + 1]>(kv_smem);
    float* s_m = reinterpret_cast<float*>(kv_smem + sizeof(float) * NS * (HD + 1));
    float* s_d = s_m + NS;
    const int b = blockIdx.x, head = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int part = lane % QL, slot = lane / QL;
    const float sm_scale = 1.44269504088896340736f / __builtin_sqrtf((float)HD);  // log2(e) / sqrt(head_dim): exp2 below
    const int pg0 = p.indptr[b], pg1 = p.indptr[b + 1];
    const int64_t seq_len = (int64_t)(pg1 - pg0 - 1) * p.page_size + p.last_page_offset[b];

    float qv[32], qsum = 0.0f;
    if (qt != nullptr) {
        // the query side of the K transform (kv_cache.py:139-140: torch.matmul(q.half(), trans_matrix_k_inv_t)) in here:
        // q' = fp16(q . qt), fp32 accumulation, one output feature per thread, handed round through LDS
        float* s_q = s_d + NS;
        const f16* qrow = q + ((size_t)b * p.num_heads + head) * HD;
        for (int j = tid; j < HD; j += NW * 64) {
            float a = 0.0f;
            for (int i = 0; i < HD; ++i) a = __builtin_fmaf((float)qrow[i], (float)qt[i * HD + j], a);
            s_q[j] = (float)(f16)a;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            qv[j] = s_q[part * 32 + j];
            qsum += qv[j];
        }
    } else {
        const f16* qp = q + ((size_t)b * p.num_heads + head) * HD + part * 32;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            qv[j] = (float)qp[j];
            qsum += qv[j];
        }
    }
    float m = -INFINITY, d = 0.0f, acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = 0.0f;

    const size_t page_stride = (size_t)p.num_layers * 2 * p.num_heads * p.page_size;
    const size_t k_off = ((size_t)p.layer_idx * 2 * p.num_heads + head) * p.page_size, kv_off = (size_t)p.num_heads * p.page_size;
    // (page, entry) of this quad's row advance incrementally: a 64-bit division per row would cost more than the row
    int pit = 0, ent = wave * RPW + slot;
    while (ent >= p.page_size) {
        ent -= p.page_size;
        ++pit;
    }
    for (int64_t pos = (int64_t)wave * RPW + slot; pos < seq_len; pos += NS) {
        const size_t page = (size_t)p.indices[pg0 + pit];
        const size_t entry = (size_t)ent;
        ent += NS;
        while (ent >= p.page_size) {
            ent -= p.page_size;
            ++pit;
        }
        const size_t ke = page * page_stride + k_off + entry, ve = ke + kv_off;   // = k_entry / v_entry, constants hoisted
        if (F16) {
            const uint4* kp = reinterpret_cast<const uint4*>(p.data + ke * (HD * 2) + part * 64);
            const uint4* vp = reinterpret_cast<const uint4*>(p.data + ve * (HD * 2) + part * 64);
            uint4 kq[4], vq[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                kq[w] = kp[w];
                vq[w] = vp[w];
            }
            float part_dot = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const f16x8 kh = __builtin_bit_cast(f16x8, kq[w]);
#pragma unroll
                for (int e = 0; e < 8; ++e) part_dot = __builtin_fmaf(qv[w * 8 + e], (float)kh[e], part_dot);
            }
#pragma unroll
            for (int off = 1; off < QL; off <<= 1) part_dot += __shfl_xor(part_dot, off, 64);
            const float x = part_dot * sm_scale;
            const float m_new = fmaxf(m, x);
            const float alpha = __builtin_amdgcn_exp2f(m - m_new), pr = __builtin_amdgcn_exp2f(x - m_new);
            d = d * alpha + pr;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const f16x8 vh = __builtin_bit_cast(f16x8, vq[w]);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[w * 8 + e] = __builtin_fmaf((float)vh[e], pr, acc[w * 8 + e] * alpha);
            }
            m = m_new;
            continue;
        }
        const uint4 kq = *reinterpret_cast<const uint4*>(p.data + ke * (HD / 2) + part * 16);
        const uint4 vq = *reinterpret_cast<const uint4*>(p.data + ve * (HD / 2) + part * 16);
        const uint32_t kpar = reinterpret_cast<const uint32_t*>(p.param)[ke];
        const uint32_t vpar = reinterpret_cast<const uint32_t*>(p.param)[ve];
        const float ks = (float)__builtin_bit_cast(f16, (unsigned short)(kpar & 0xFFFF)), kz = (float)__builtin_bit_cast(f16, (unsigned short)(kpar >> 16));
        const float vs = (float)__builtin_bit_cast(f16, (unsigned short)(vpar & 0xFFFF)), vz = (float)__builtin_bit_cast(f16, (unsigned short)(vpar >> 16));
        const uint32_t kw[4] = {kq.x, kq.y, kq.z, kq.w}, vw[4] = {vq.x, vq.y, vq.z, vq.w};
        float dotn = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int e = 0; e < 8; ++e) dotn = __builtin_fmaf(qv[w * 8 + e], (float)((kw[w] >> (4 * e)) & 15u), dotn);
        float part_dot = ks * dotn - kz * qsum;       // this lane's 32 features of q . k
#pragma unroll
        for (int off = 1; off < QL; off <<= 1) part_dot += __shfl_xor(part_dot, off, 64);
        const float x = part_dot * sm_scale;
        const float m_new = fmaxf(m, x);
        const float alpha = __builtin_amdgcn_exp2f(m - m_new), pr = __builtin_amdgcn_exp2f(x - m_new);
        d = d * alpha + pr;
        const float pvs = pr * vs, pvz = pr * vz;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                acc[w * 8 + e] = __builtin_fmaf((float)((vw[w] >> (4 * e)) & 15u), pvs, acc[w * 8 + e] * alpha - pvz);
        m = m_new;
    }
    const int st = wave * RPW + slot;
    if (part == 0) {
        s_m[st] = m;
        s_d[st] = d;
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) s_o[st][part * 32 + j] = acc[j];
    __syncthreads();
    if (tid < HD) {   // state.cuh merge: rescale every partial state to the common maximum
        // (bounded unrolling: fully unrolled for NS = 64, these loops hoisted 128 LDS reads and pushed the 4-wave build from
        //  198 to 260 VGPRs — one wave per SIMD, 48 -> 71 us at 16 x 2048 — when the output select below was added)
        float mm = -INFINITY;
#pragma unroll 8
        for (int s = 0; s < NS; ++s) mm = fmaxf(mm, s_m[s]);
        float dd = 0.0f, oo = 0.0f;
#pragma unroll 8
        for (int s = 0; s < NS; ++s) {
            const float w = s_m[s] == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(s_m[s] - mm);
            dd += s_d[s] * w;
            oo += s_o[s][tid] * w;
        }
        // transpose_out: [batch, head_dim, heads] — the layout the o_proj head transform takes (modeling_llama.py:147-149
        // transposes and copies the attention output before block_matmul)
        const size_t oi = transpose_out ? ((size_t)b * HD + tid) * p.num_heads + head : ((size_t)b * p.num_heads + head) * HD + tid;
        o[oi] = dd > 0.0f ? (f16)(oo / dd) : (f16)0.0f;  // an empty sequence attends to nothing: zeros, not 0/0
    }
}

}  // namespace

static PagedKv make_kv(void* kv_data, void* kv_param, const int* indptr, const int* indices, const int* last, int num_layers,
                       int layer_idx, int num_heads, int page_size, int head_dim, int batch) {
    PagedKv p;
    p.data = (uint8_t*)kv_data;
    p.param = (f16*)kv_param;
    p.indptr = indptr;
    p.indices = indices;
    p.last_page_offset = last;
    p.num_layers = num_layers;
    p.layer_idx = layer_idx;
    p.num_heads = num_heads;
    p.page_size = page_size;
    p.head_dim = head_dim;
    p.batch_size = batch;
    return p;
}

int fq_launch_kv_append(void* kv_data, void* kv_param, const int* indptr, const int* indices, const int* last, const uint8_t* k,
                        const uint8_t* v, const f16* kparam, const f16* vparam, const int* seqlen_indptr, int64_t total_tokens,
                        int num_layers, int layer_idx, int num_heads, int page_size, int head_dim, int batch, int group, int n_cu,
                        hipStream_t stream, bool f16_cache) {
    if (head_dim % 32 || group < 1 || num_heads % group) return -1000;
    const PagedKv p = make_kv(kv_data, kv_param, indptr, indices, last, num_layers, layer_idx, num_heads, page_size, head_dim, batch);
    const int64_t items = total_tokens * num_heads * (f16_cache ? head_dim / 8 : head_dim / 32);
    int64_t blocks = (items + 255) / 256;
    if (blocks > (int64_t)n_cu * 16) blocks = (int64_t)n_cu * 16;
    if (blocks < 1) blocks = 1;
    if (f16_cache)
        hipLaunchKernelGGL(fq_kv_append_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, p, k, v, kparam, vparam,
                           seqlen_indptr, total_tokens, group);
    else
        hipLaunchKernelGGL(fq_kv_append_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, p, k, v, kparam, vparam,
                           seqlen_indptr, total_tokens, group);
    return (int)hipGetLastError();
}

int fq_launch_kv_decode(f16* o, const f16* q, void* kv_data, void* kv_param, const int* indptr, const int* indices, const int* last,
                        int num_layers, int layer_idx, int num_heads, int page_size, int head_dim, int batch, const f16* qt,
                        int transpose_out, hipStream_t stream, bool f16_cache) {
    const PagedKv p = make_kv(kv_data, kv_param, indptr, indices, last, num_layers, layer_idx, num_heads, page_size, head_dim, batch);
    const dim3 grid((unsigned)batch, (unsigned)num_heads);
    const bool wide = (int64_t)batch * num_heads < 512;  // fewer than two workgroups per CU: 8 waves each instead of 4 (measured: 158 -> 113 us at 8 x 8192; no gain from 512 up)
#define FQ_DEC(HD_, NW_)                                                                                              \
    {                                                                                                                 \
        constexpr size_t lds = sizeof(float) * ((size_t)(NW_ * (64 / (HD_ / 32))) * (HD_ + 1 + 2) + HD_);                     \
        if (f16_cache) {                                                                                              \
            FQ_RAISE_LDS_CAP((fq_kv_decode_kernel<HD_, NW_, true>), lds);                                             \
            hipLaunchKernelGGL((fq_kv_decode_kernel<HD_, NW_, true>), grid, dim3(NW_ * 64), lds, stream, o, q, p, qt, transpose_out); \
        } else {                                                                                                      \
            FQ_RAISE_LDS_CAP((fq_kv_decode_kernel<HD_, NW_>), lds);                                                   \
            hipLaunchKernelGGL((fq_kv_decode_kernel<HD_, NW_>), grid, dim3(NW_ * 64), lds, stream, o, q, p, qt, transpose_out); \
        }                                                                                                             \
    }
    if (head_dim == 128) {
        if (wide) FQ_DEC(128, 8) else FQ_DEC(128, 4)
    } else if (head_dim == 64) {
        if (wide) FQ_DEC(64, 8) else FQ_DEC(64, 4)
    } else {
        return -1000;
    }
#undef FQ_DEC
    return (int)hipGetLastError();
}
