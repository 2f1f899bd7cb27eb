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
// Here: one 4-wave workgroup per (request, head); a cached row is 64 bytes, a lane owns 16 bytes = 32 features of one row per step and a
// wave reads 16 consecutive rows = 1 KB per instruction. Every row's four lanes run their OWN online softmax over the rows they see; the
// 64 partial states (m, d, o[head_dim]) are merged once at the end through LDS (the standard max-rescaled merge, state.cuh).
//   * fp16 cache (fq_kv_decode_kernel): a quad of neighbouring lanes owns a row, the dot product is 32 fp32 FMAs + a 4-lane reduction.
//   * INT4 cache (fq_kv_decode_i4_kernel, round 5): the scalar form of rounds 1-4 spent ~9 VALU operations per cached byte (extract,
//     convert, FMA per nibble, twice) and was VALU-bound at 0.33-0.39 of the HBM roofline. Now (a) the eight nibbles of a dword become the
//     eight EXACT fp16 values 16 + n with eight operations (a shift and a v_and_or_b32 per pair: the nibble lands in the top mantissa
//     bits of 16.0; the offset leaves both sums as one scalar correction per row); (b) q . k runs on the matrix pipe:
//     v_mfma_f32_16x16x32_f16 with the query in all 16 rows of A and the wave's 16 cached rows as the columns of B — lane l = 16 g + i
//     supplies features 32 g .. 32 g + 31 of row i, exactly the 16 bytes it loaded, and receives row i's full 128-feature sum (exact
//     fp16 x fp16 products, fp32 accumulation: no 4-lane reduction); (c) p . v reads the fp16 halves directly (v_fma_mix_f32: one
//     operation per feature) and the dequantisation stays folded: q . k = scale * sum(q_j n_j) - zero * sum(q_j), o = sum_i (p_i scale_i)
//     n_i - sum_i p_i zero_i with the second sum a scalar per lane; (d) a lane's running softmax refers to a bound 2^8 above the row that
//     last raised it, not to its exact maximum (which moves in most steps of a 2048-token request and costs 34 register rescales in every
//     lane each time), rescaled in a wave-uniform branch; (e) the page index is a scalar load where a wave's 16 rows cannot straddle a
//     page, the row addresses a scalar base + a loop-invariant lane offset, and the next step's rows are requested before the current
//     step's arithmetic inside a condition-free steady-state loop (counted vmcnt waits); (f) the lane states merge on ALL threads (a wave
//     reduction for the maximum, every state's weight once, chunked sums), and with few (request, head) pairs a request's rows are split
//     over several workgroups (SPLIT below). head_dim 64: two lanes per row, 32 rows per step, v_mfma_f32_32x32x16_f16.
//     64 requests x 2048 tokens x 32 heads: 180.7 -> 102.7 us (0.39 -> 0.69 of 8 TB/s; FETCH_SIZE 571 MB for 570 MB of rows, 136 VALU
//     instructions per 16-row step where the scalar kernel had ~300), 8 x 8192: 108 -> 46 us (0.77), 128 x 4096: 0.74, one request x 2048:
//     20.8 -> 8.6 us (profiles/r05_kvdecode_timing.txt, r05_kvdecode_pmc.txt).
#include "fq_kv_common.hpp"
#include <type_traits>

namespace {

// (round 6, third session) THE STEP'S OWN ROW INSIDE THE DECODE LAUNCH. A decode step appends one token per request and then attends over the
// cache including it (kv_cache.py:283-359): until now two launches — fq_kv_quant_append_i4 (K transform + K / V INT4 quantisation + scatter,
// 5 - 11 us of dependent round trips for 64 bytes per head) and the attention, which cannot start before the first has finished. With `k` set the
// decode launch does the first one itself: the workgroup whose rows include the request's LAST row (length - 1: the new token; the host has
// already advanced the lengths, as for fq_kv_quant_append_i4) transforms and quantises the new K / V row of its cache head in its prologue — the
// arithmetic of fq_kv_quant_kernel to the bit: the same MFMA on the same fragment image of the transform, the same extrema, (scale, zero) and
// kv_q8 — keeps the 64 + 64 bytes and the two parameters in LDS, hands them to the lanes of that row in the tail steps of the row loop INSTEAD of
// what the cache holds there, and (one workgroup per cache head) writes them to the cache for the steps to come. No workgroup waits for another.
struct KvNew {
    const f16* k;        // [batch, src_heads, head_dim] fp16, the new token's keys (nullptr: nothing to append — the plain decode launch)
    const f16* v;        // ... values
    const uint4* timg;   // the K transform as the fragment image fq_kv_transform_image_f16 wrote (32 KB at head_dim 128), or nullptr: no transform
    int src_heads;       // heads of k / v: cache head c is a copy of source head c / (cache heads / src_heads) (kv_cache.py:286-296)
};

struct PagedKv {
    uint8_t* data;
    f16* param;            // half2 per entry
    const int* indptr;
    const int* indices;
    const int* last_page_offset;
    int num_layers, layer_idx, num_heads, page_size, head_dim, batch_size;
    int copies;            // decode only: cache heads c * copies .. c * copies + copies - 1 hold IDENTICAL rows (the reference's replicated layout,
                           // kv_cache.py:286-296) and the launch reads the first of them for all of the group's query heads; 1: every head its own rows
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

typedef const int __attribute__((address_space(4))) kv_const_int;
constexpr int KV_PERM[8] = {0, 4, 1, 5, 2, 6, 3, 7};   // feature (within the dword's eight) of packed slot e
constexpr float KV_OFF = 16.0f;
constexpr float KV_MARGIN = 8.0f;   // see the step of fq_kv_decode_kernel
__device__ __forceinline__ uint32_t kv_and_or(uint32_t x, uint32_t mask, uint32_t bits) {
    uint32_t r;   // (gfx9: one constant-bus operand per VOP3 — the mask in an SGPR, the exponent bits in a VGPR)
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(mask), "v"(bits));
    return r;
}
__device__ __forceinline__ void kv_unpack8(uint32_t w, uint32_t ebits, uint32_t (&r)[4]) {
    constexpr uint32_t M = 0x03C003C0u;
    r[0] = kv_and_or(w << 6, M, ebits);
    r[1] = kv_and_or(w << 2, M, ebits);
    r[2] = kv_and_or(w >> 2, M, ebits);
    r[3] = kv_and_or(w >> 6, M, ebits);
}
// acc += float(half h of the packed pair) * s: v_fma_mix_f32 reads the fp16 half directly (a convert + an FMA otherwise)
template <int HI>
__device__ __forceinline__ void kv_fma_half(float& acc, uint32_t pair, float s) {
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(pair), "v"(s));
    else asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(pair), "v"(s));
}

// sum_j q_j n_ij for the wave's RPW cached rows on the matrix pipe: A = the query in every row, B = the rows as columns. Lane
// l = RPW g + i holds features 32 g .. 32 g + 31 of row i (one dword per K-step) and gets row i's sum over ALL features back.
template <int QL>
__device__ __forceinline__ float kv_qk_mfma(const f16x8 (&qa)[4], const f16x8 (&kb)[4]) {
    if constexpr (QL == 4) {
        f32x4 c = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int w = 0; w < 4; ++w) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[w], kb[w], c, 0, 0, 0);
        return c[0];
    } else {
        f32x16 c = {0};
#pragma unroll
        for (int w = 0; w < 4; ++w) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[w], kb[w], c, 0, 0, 0);
        return c[0];
    }
}
// The same for head_dim 128 with the first four D rows kept: with A row m = the query of head m % QG (fq_kv_decode_kernel<.., QG>), lane
// (part, i) — which holds D rows 4 part .. 4 part + 3 of column i — gets the sums of heads 0 .. 3 for ITS row i in c[0..3], no lane exchange.
__device__ __forceinline__ f32x4 kv_qk_mfma4(const f16x8 (&qa)[4], const f16x8 (&kb)[4]) {
    f32x4 c = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int w = 0; w < 4; ++w) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[w], kb[w], c, 0, 0, 0);
    return c;
}
template <int QL>
__device__ __forceinline__ float kv_qk(const f16x8 (&qa)[4], const uint4 kq, uint32_t ebits) {   // INT4 row -> sum_j q_j (16 + n_ij)
    const uint32_t kw[4] = {kq.x, kq.y, kq.z, kq.w};
    f16x8 kb[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        uint32_t r[4];
        kv_unpack8(kw[w], ebits, r);
        kb[w] = __builtin_bit_cast(f16x8, u32x4{r[0], r[1], r[2], r[3]});
    }
    return kv_qk_mfma<QL>(qa, kb);
}

__device__ __forceinline__ size_t kv_uniform64(size_t v) {
    return ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

// the split decode's workspace: a counter per (request, head) pair first (a fixed place, whatever the split count), the states behind them
__device__ __forceinline__ float* kv_states(float* ws, size_t pairs) { return ws + ((pairs + 3) & ~(size_t)3); }

#ifndef KV_MERGE_HEADS
#define KV_MERGE_HEADS 1   // (measurement knob) 0: a workgroup per query head on a shared cache too (fq_launch_kv_decode)
#endif
#ifndef KV_ABL
#define KV_ABL 0   // measurement builds: 1 no row loop, 2 no state sums in the merge
#endif
#ifndef KV_DEPTH
#define KV_DEPTH 1   // steps of rows in flight under a step's arithmetic (measured: 1, 2, 3 within 2 % — profiles/r05_kvdecode_timing.txt)
#endif
// Decode attention (batch_decode_i4 / batch_decode_f16): see the head of this file. Lane l = RPW part + slot: row `slot` of the wave's
// RPW rows, chunk `part` (32 features) of that row: 16 bytes of the INT4 cache (a wave's load is RPW consecutive rows = 1 KB), or 64
// bytes = four 16-byte loads of the fp16 configuration (F16: no (scale, zero), no unpacking — the loaded dwords ARE the B fragments).
// SPLIT (round 5): a request's rows over gridDim.z workgroups — with few (request, head) pairs the launch fills a fraction of the chip
// (one request x 32 heads: 32 of 256 CUs, ~20 us for 9 MB). The workgroups of a pair take the wave-steps round robin (virtual wave
// blockIdx.z * NW + wave of gridDim.z * NW), each leaves its un-normalised (m, d, o[HD]) in `ws`, and the LAST one to arrive (a counter
// per pair behind the states, left at zero again) merges them like the states of one workgroup and writes the output.
// QG (round 6): QUERY HEADS PER WORKGROUP over a cache that holds the KV heads once (grouped-query attention, fq_kv_batch_decode_gqa):
// blockIdx.y is the CACHE head, the workgroup serves its QG query heads from ONE pass over the rows — the rows are loaded and their
// nibbles unpacked once, the q . k MFMA that had one useful A row of sixteen carries all QG queries (A row m = query m % QG), and only the
// per-head softmax state and the p . v accumulation are QG-fold. head_dim 128 only (the 16 x 16 x 32 shape's D rows). QG = 1: every
// workgroup one query head (qgroup: how many of them share a cache head; 1 = the reference's replicated cache).
#ifndef KV_PV_MFMA
#define KV_PV_MFMA 1
#endif
#ifndef KV_PVM_OCC4
#define KV_PVM_OCC4 1      // the matrix-pipe form of the merged launch held to 128 VGPRs (four waves per SIMD; two registers spill) and run on eight-wave workgroups up to 1024 of them:
                           // the launch 36.2 -> 33.8 us at 64 requests, the step 140.1 -> 135.8 (r06c61); 0: 134 VGPRs, four-wave workgroups from 512 on
#endif
#ifndef KV_PVM_WIDE_MAX
#define KV_PVM_WIDE_MAX 512    // the merged matrix-pipe launch runs eight-wave workgroups up to this many of them (two per CU); beyond, four-wave workgroups, four per CU:
                               // the step at 80 / 96 / 128 requests 164.5 / 172.0 / 210.8 -> 159.4 / 168.1 / 206.5 us against a bound of 1024 (r06c75)
#endif
template <int HD, int NW, bool UNI, bool F16, bool SPLIT, int QG = 1, bool APPEND = false>
__global__ __launch_bounds__(NW * 64, (KV_PVM_OCC4 && KV_PV_MFMA && QG == 4 && !F16 && UNI) ? 4 : 2) void fq_kv_decode_kernel(f16* __restrict__ o, const f16* __restrict__ q, PagedKv p,
                                                               const f16* __restrict__ qt, int transpose_out, float* ws, int qgroup, KvNew nw) {
    constexpr int QL = HD / 32;        // lanes per cached row: 4 for head_dim 128, 2 for 64
    constexpr int RPW = 64 / QL;       // rows per wave and step: the N of the MFMA (16 / 32)
    constexpr int NS = NW * RPW;       // partial softmax states per workgroup
    static_assert(QL == 4 || QL == 2, "head_dim 128 (16x16x32) or 64 (32x32x16)");
    static_assert(!APPEND || (HD == 128 && !F16 && UNI), "the step's own row inside the launch: INT4 cache, head_dim 128, rows that never straddle a page");
    extern __shared__ __attribute__((aligned(16))) unsigned char kv_smem[];
    float (*s_o)[HD + 1] = reinterpret_cast<float (*)[HD + 1]>(kv_smem);
    float* s_m = reinterpret_cast<float*>(kv_smem + sizeof(float) * NS * (HD + 1));
    float* s_d = s_m + NS;
    float* s_q = s_d + NS;
    // (round 6) qgroup > 1: a cache that holds the KV heads ONCE (grouped-query attention without the reference's per-query-head copies,
    // kv_cache.py:286-296): query head `head` of QH = gridDim.y reads cache head head / qgroup of p.num_heads = QH / qgroup. Same values
    // read, same arithmetic: the output is bit-identical to the replicated cache's; the cache is qgroup times smaller and the qgroup
    // workgroups of a KV head re-read rows the memory-side cache still holds.
    static_assert(QG == 1 || (HD == 128 && (QG == 2 || QG == 4)), "several query heads per workgroup: head_dim 128, groups of 2 or 4");
    const int b = blockIdx.x, head = QG > 1 ? (int)blockIdx.y * QG : (int)blockIdx.y;        // (QG > 1: the FIRST query head of this workgroup)
    const int QH = QG > 1 ? (int)gridDim.y * QG : (int)gridDim.y, chead = (head / qgroup) * p.copies;       // (QG > 1: the workgroup's QG query heads lie inside one group: QG divides qgroup)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int part = lane / RPW, slot = lane % RPW;
    const float sm_scale = 1.44269504088896340736f / __builtin_sqrtf((float)HD);  // log2(e) / sqrt(head_dim): exp2 below
    const int pg0 = p.indptr[b], pg1 = p.indptr[b + 1];
    const int64_t seq_len = (int64_t)(pg1 - pg0 - 1) * p.page_size + p.last_page_offset[b];

    // the query as fp16 values in LDS: q itself, or q' = fp16(q . qt) (kv_cache.py:139-140: torch.matmul(q.half(),
    // trans_matrix_k_inv_t)), fp32 accumulation — every thread sums HD / NCHQ terms of one output feature (one thread per feature
    // walked all HD terms in a dependent chain: microseconds in front of every workgroup, and a split launch has many), the NCHQ
    // partial sums meet in LDS (s_o is free until the states are written)
#ifndef KV_QT_ONCE
#define KV_QT_ONCE 1    // QG > 1: the query transform of the workgroup's heads in ONE pass over qt (0: a pass and two barriers per head, rounds 5-6)
#endif
    if (QG > 1 && KV_QT_ONCE && qt != nullptr) {
        // (third session) the same sums in the same order for every head — qt is read once instead of QG times and the heads share one barrier pair
        constexpr int NCHQ = NW * 64 / HD, CHQ = HD / NCHQ;
        static_assert(QG * NCHQ <= NS, "partial sums of all heads fit the scratch rows");
        const int j = tid % HD, c = tid / HD;
        const f16* qrow0 = q + ((size_t)b * QH + head) * HD;
        float a[QG];
#pragma unroll
        for (int g = 0; g < QG; ++g) a[g] = 0.0f;
#pragma unroll 8
        for (int i = c * CHQ; i < (c + 1) * CHQ; ++i) {
            const float t = (float)qt[i * HD + j];
#pragma unroll
            for (int g = 0; g < QG; ++g) a[g] = __builtin_fmaf((float)qrow0[(size_t)g * HD + i], t, a[g]);
        }
#pragma unroll
        for (int g = 0; g < QG; ++g) s_o[g * NCHQ + c][j] = a[g];
        __syncthreads();
        for (int idx = tid; idx < QG * HD; idx += NW * 64) {
            const int g = idx / HD, f = idx - g * HD;
            float t = 0.0f;
#pragma unroll
            for (int cc = 0; cc < NCHQ; ++cc) t += s_o[g * NCHQ + cc][f];
            s_q[g * HD + f] = (float)(f16)t;
        }
        __syncthreads();
    } else {
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const f16* qrow = q + ((size_t)b * QH + head + g) * HD;
        if (qt != nullptr) {
            constexpr int NCHQ = NW * 64 / HD, CHQ = HD / NCHQ;
            const int j = tid % HD, c = tid / HD;
            float a = 0.0f;
#pragma unroll 8
            for (int i = c * CHQ; i < (c + 1) * CHQ; ++i) a = __builtin_fmaf((float)qrow[i], (float)qt[i * HD + j], a);
            s_o[c][j] = a;
            __syncthreads();
            if (tid < HD) {
                float t = 0.0f;
#pragma unroll
                for (int cc = 0; cc < NCHQ; ++cc) t += s_o[cc][tid];
                s_q[g * HD + tid] = (float)(f16)t;
            }
        } else {
            for (int j = tid; j < HD; j += NW * 64) s_q[g * HD + j] = (float)qrow[j];
        }
        __syncthreads();
    }
    }
    // the A operand of K-step w: features 32 part + 8 w + feat(e) — INT4: the order kv_unpack8 leaves the nibbles in
    auto feat = [](int e) { return F16 ? e : KV_PERM[e]; };
    f16x8 qa[4];
    float qsum[QG], qoff[QG];
    {
        const int ga = slot % QG;   // the query this lane's A row carries (QG = 1: the one query in every row)
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int e = 0; e < 8; ++e) qa[w][e] = (f16)s_q[ga * HD + part * 32 + w * 8 + feat(e)];
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int e = 0; e < 8; ++e) t += s_q[g * HD + part * 32 + w * 8 + feat(e)];
#pragma unroll
            for (int off = RPW; off < 64; off <<= 1) t += __shfl_xor(t, off, 64);   // over the row's QL lanes: sum of ALL features
            qsum[g] = t;
            qoff[g] = KV_OFF * t;
        }
    }
    uint32_t ebits = 0x4C004C00u;   // (16.0, 16.0): kept in a VGPR
    asm volatile("" : "+v"(ebits));

    // APPEND: the new row (see KvNew). s_new (behind every other LDS array): [0..15] the K row's dwords, [16..31] the V row's, [32] / [33] the
    // (scale, zero) pairs, [36..] the per-wave extrema of the prologue.
    constexpr size_t NEW_OFF = (sizeof(float) * ((size_t)NS * (HD + 1 + 2) + QG * HD + NS + NW + (size_t)(NW * 64 / HD) * (HD + 2)) + 15) & ~(size_t)15;
    uint32_t* s_new = reinterpret_cast<uint32_t*>(kv_smem + NEW_OFF);
    const int64_t last_row = seq_len - 1;
    bool own = false;    // (workgroup-uniform) this workgroup's rows include last_row
    if constexpr (APPEND) {
        if (nw.k != nullptr && seq_len > 0) {
            const int wsteps = SPLIT ? (int)gridDim.z * NW : NW;                    // waves that share the pair's rows, 16 rows a step each
            own = !SPLIT || (int)(((last_row / RPW) % wsteps) / NW) == (int)blockIdx.z;
        }
        if (own) {
            using namespace fqkv;
            unsigned short* s_ext = reinterpret_cast<unsigned short*>(s_new + 36);   // [k, v][wave 0..3][max, min]
            const int cgroup = p.num_heads / nw.src_heads, shead = chead / cgroup;
            const f16* krow = nw.k + ((size_t)b * nw.src_heads + shead) * HD;
            const f16* vrow = nw.v + ((size_t)b * nw.src_heads + shead) * HD;
            const int h = lane >> 5;
            f16 kv16[2][16];   // [k, v][r]: feature 32 wave + 16 h + r of the (transformed) row — the columns lane (h, .) of fq_kv_quant_kernel's wave ends with
            if (wave < 4) {
                if (nw.timg != nullptr) {
                    f16x8 xf[8], tf[8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        tf[s] = __builtin_bit_cast(f16x8, nw.timg[(s * 4 + wave) * 64 + lane]);
                        xf[s] = __builtin_bit_cast(f16x8, reinterpret_cast<const uint4*>(krow)[2 * s + h]);
                    }
                    f32x16 a = {0};
#pragma unroll
                    for (int s = 0; s < 8; ++s) a = __builtin_amdgcn_mfma_f32_32x32x16_f16(tf[s], xf[s], a, 0, 0, 0);   // (every B column is the one row)
#pragma unroll
                    for (int r = 0; r < 16; ++r) kv16[0][r] = (f16)a[r];
                } else {
                    const f16x8 k0 = __builtin_bit_cast(f16x8, reinterpret_cast<const uint4*>(krow + wave * 32 + 16 * h)[0]);
                    const f16x8 k1 = __builtin_bit_cast(f16x8, reinterpret_cast<const uint4*>(krow + wave * 32 + 16 * h)[1]);
#pragma unroll
                    for (int r = 0; r < 8; ++r) kv16[0][r] = k0[r], kv16[0][8 + r] = k1[r];
                }
                const f16x8 v0 = __builtin_bit_cast(f16x8, reinterpret_cast<const uint4*>(vrow + wave * 32 + 16 * h)[0]);
                const f16x8 v1 = __builtin_bit_cast(f16x8, reinterpret_cast<const uint4*>(vrow + wave * 32 + 16 * h)[1]);
#pragma unroll
                for (int r = 0; r < 8; ++r) kv16[1][r] = v0[r], kv16[1][8 + r] = v1[r];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f16x2 pmx = {kv16[t][0], kv16[t][1]}, pmn = pmx;
#pragma unroll
                    for (int r = 2; r < 16; r += 2) {
                        const f16x2 pr = {kv16[t][r], kv16[t][r + 1]};
                        pmx = fq_pk_max(pmx, pr);
                        pmn = fq_pk_min(pmn, pr);
                    }
                    f16 mx = pmx[0] > pmx[1] ? pmx[0] : pmx[1], mn = pmn[0] < pmn[1] ? pmn[0] : pmn[1];
                    const f16 omx = xchg32(mx, lane), omn = xchg32(mn, lane);
                    mx = omx > mx ? omx : mx;
                    mn = omn < mn ? omn : mn;
                    if (lane == 0) {
                        s_ext[(t * 4 + wave) * 2 + 0] = __builtin_bit_cast(unsigned short, mx);
                        s_ext[(t * 4 + wave) * 2 + 1] = __builtin_bit_cast(unsigned short, mn);
                    }
                }
            }
            __syncthreads();
            if (wave < 4) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f16 mx = __builtin_bit_cast(f16, s_ext[(t * 4) * 2]), mn = __builtin_bit_cast(f16, s_ext[(t * 4) * 2 + 1]);
#pragma unroll
                    for (int w = 1; w < 4; ++w) {
                        const f16 a = __builtin_bit_cast(f16, s_ext[(t * 4 + w) * 2]), c = __builtin_bit_cast(f16, s_ext[(t * 4 + w) * 2 + 1]);
                        mx = a > mx ? a : mx;
                        mn = c < mn ? c : mn;
                    }
                    const KvParams pp = kv_params<false>(mx, mn, (f16)1.0f, (f16)1.0f);   // (kv_cache.py:283-284: the cache's own calls leave lac off)
                    const float sc = (float)pp.scale, rc = fq_fast_inv(sc);
                    const uint32_t zero2 = __builtin_bit_cast(uint32_t, f16x2{pp.zero, pp.zero});
                    uint32_t pr[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) pr[j] = __builtin_bit_cast(uint32_t, f16x2{kv16[t][2 * j], kv16[t][2 * j + 1]});
                    const unsigned w0 = kv_q8<false>(pr[0], pr[1], pr[2], pr[3], rc, sc, zero2);
                    const unsigned w1 = kv_q8<false>(pr[4], pr[5], pr[6], pr[7], rc, sc, zero2);
                    if ((lane & 31) == 0) {      // features 32 wave + 16 h .. + 15 = dwords 4 wave + 2 h, + 1
                        s_new[t * 16 + wave * 4 + 2 * h] = w0;
                        s_new[t * 16 + wave * 4 + 2 * h + 1] = w1;
                        if (wave == 0 && h == 0)
                            s_new[32 + t] = (uint32_t)__builtin_bit_cast(unsigned short, pp.scale) | ((uint32_t)__builtin_bit_cast(unsigned short, pp.zero) << 16);
                    }
                }
            }
            __syncthreads();
            // the cache gets the row: the workgroup of the cache head's FIRST query head (a replicated cache: every head its copy)
            if (head % qgroup == 0 && tid < 10 * p.copies) {     // (p.copies > 1: this workgroup reads the group's first copy and writes them all)
                const int cpy = tid / 10, t = tid - cpy * 10;
                const size_t page = (size_t)p.indices[pg0 + (int)(last_row / p.page_size)];
                const size_t entry = (size_t)(last_row % p.page_size);
                const size_t ke = k_entry(p, page, (size_t)(chead + cpy), entry), ve = v_entry(p, page, (size_t)(chead + cpy), entry);
                if (t < 4) reinterpret_cast<uint4*>(p.data + ke * (HD / 2))[t] = reinterpret_cast<const uint4*>(s_new)[t];
                else if (t < 8) reinterpret_cast<uint4*>(p.data + ve * (HD / 2))[t - 4] = reinterpret_cast<const uint4*>(s_new + 16)[t - 4];
                else reinterpret_cast<uint32_t*>(p.param)[t == 8 ? ke : ve] = s_new[32 + (t - 8)];
            }
        }
    }

    float m[QG], d[QG], zacc[QG], acc[QG][32];   // acc[g][8 w + e]: feature 32 part + 8 w + KV_PERM[e] of sum_i p_i s_i (16 + n_i); zacc: sum_i p_i (z_i + 16 s_i)
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        m[g] = -INFINITY, d[g] = 0.0f, zacc[g] = 0.0f;
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[g][j] = 0.0f;
    }
    // (round 6, third session) PVM — p . v ON THE MATRIX PIPE where one workgroup serves four query heads. That launch is VALU-bound (302 VALU
    // instructions per 16-row wave-step, the pipe ~88 % busy at 2.9 TB/s: profiles/r06_read_one_copy.txt): 128 of them are the fp32 FMAs of
    // p . v for four heads — 8192 multiply-adds per step, the work of ONE MFMA. The contraction runs over ROWS, which sit in different lanes,
    // so the rows' unpacked values (exact fp16 16 + n) go through LDS as eight [16 rows][16 features] tiles and come back TRANSPOSED
    // (ds_read_b64_tr_b16 at byte 8 * lane of a tile = the B fragment of v_mfma_f32_16x16x16_f16: lane (G, n) gets rows 4G .. 4G + 3 of
    // column n; tools/microbench/tr_probe.hip), the four heads' weights p_i s_i 2^8 (fp16: the one rounding this form adds) go the same way
    // as a [16 rows][16] tile whose columns 4 .. 15 stay zero (the A fragment: lane (G, m) gets rows 4G .. 4G + 3 of head m), and eight
    // MFMAs accumulate O[head][feature] — 32 accumulator registers instead of 128. All rows of a wave feed one accumulator, so the softmax
    // reference m[g] is WAVE-uniform here (a bound KV_MARGIN above the largest score seen when it last moved — a wave reduction then, rare),
    // the denominators stay per lane. The zero-point sum uses the ROUNDED weights, so the 16-offset cancels exactly as before.
#ifndef KV_PV_MFMA
#define KV_PV_MFMA 1
#endif
    constexpr bool PVM = KV_PV_MFMA && QG == 4 && !F16 && UNI;
    typedef __fp16 h4 __attribute__((__vector_size__(8)));
    typedef __attribute__((address_space(3))) h4 lds_h4;
    f32x4 oacc[PVM ? 8 : 1];
    unsigned char* pv_vt = kv_smem + (size_t)wave * 4096;                       // (the tiles alias s_o, which is free until the states are written)
    unsigned char* pv_pt = kv_smem + (size_t)NW * 4096 + (size_t)wave * 512;
    if constexpr (PVM) {
#pragma unroll
        for (int t = 0; t < 8; ++t) oacc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        reinterpret_cast<uint2*>(pv_pt)[lane] = make_uint2(0u, 0u);             // (wave-private: LDS operations of a wave execute in order)
    }

    const size_t page_stride = (size_t)p.num_layers * 2 * p.num_heads * p.page_size;
    const size_t k_off = ((size_t)p.layer_idx * 2 * p.num_heads + chead) * p.page_size, kv_off = (size_t)p.num_heads * p.page_size;
    // Where the wave's next RPW rows live, advanced incrementally in request order (a 64-bit division per row would cost more than
    // the row). UNI (page_size % RPW == 0: the wave's RPW consecutive rows never straddle a page): ONE page index per wave and step, a
    // scalar load on its own counter, and the lanes differ by `slot` entries; rows of the last step beyond the sequence lie in the same
    // (allocated) page and are masked. Otherwise every lane walks its own (page, entry) and a row beyond the sequence reads entry 0
    // of the request's last page instead. The loop itself is WAVE-uniform either way (the MFMA wants every lane).
    const int vwave = SPLIT ? (int)blockIdx.z * NW + wave : wave;        // this wave's place among the pair's waves
    const int STRIDE = SPLIT ? (int)gridDim.z * NS : NS;                 // rows between two steps of a wave
    int pit, ent;
    if (UNI) {
        pit = (vwave * RPW) / p.page_size;
        ent = vwave * RPW - pit * p.page_size;
    } else {
        pit = 0, ent = vwave * RPW + slot;
        while (ent >= p.page_size) {
            ent -= p.page_size;
            ++pit;
        }
    }
    struct Rows {
        uint4 kq[F16 ? 4 : 1], vq[F16 ? 4 : 1];
        uint32_t kpar, vpar;
    };
    constexpr int ROWB = F16 ? HD * 2 : HD / 2;                          // bytes of a cached row
    const unsigned lane_row = (unsigned)slot * ROWB + (unsigned)part * (F16 ? 64 : 16);   // UNI: this lane's bytes behind the wave's first row
    auto request = [&](int64_t base, Rows& r) {
        const uint8_t *kp_, *vp_;
        const uint32_t *kpar_, *vpar_;
        if (UNI) {
            // (the page index through the constant address space: a scalar load on lgkmcnt — as a vector load it would sit in front of
            //  the rows on vmcnt, and waiting for it would wait for every row in flight. Everything up to the wave's first row is
            //  wave-uniform — scalar arithmetic, an SGPR base for the loads — and the lane adds a loop-invariant 32-bit offset:
            //  the 64-bit address arithmetic per lane and request was ~25 of the ~160 VALU instructions of a step, rocprofv3 r05c31)
            const size_t page = (size_t)reinterpret_cast<kv_const_int*>(reinterpret_cast<uintptr_t>(p.indices))[pg0 + pit];
            const size_t e0 = kv_uniform64(page * page_stride + k_off + (size_t)ent);   // (told to the compiler: it is wave-uniform)
            const uint8_t* kb = p.data + e0 * ROWB;
            kp_ = kb + lane_row;
            vp_ = kb + kv_off * ROWB + lane_row;
            kpar_ = reinterpret_cast<const uint32_t*>(p.param) + e0 + slot;
            vpar_ = kpar_ + kv_off;
        } else {
            const bool valid = base + slot < seq_len;
            const size_t page = (size_t)p.indices[valid ? pg0 + pit : pg1 - 1];
            const size_t ke = page * page_stride + k_off + (valid ? (size_t)ent : 0), ve = ke + kv_off;   // = k_entry / v_entry, constants hoisted
            kp_ = p.data + ke * ROWB + part * (F16 ? 64 : 16);
            vp_ = p.data + ve * ROWB + part * (F16 ? 64 : 16);
            kpar_ = reinterpret_cast<const uint32_t*>(p.param) + ke;
            vpar_ = reinterpret_cast<const uint32_t*>(p.param) + ve;
        }
        ent += STRIDE;
        while (ent >= p.page_size) {
            ent -= p.page_size;
            ++pit;
        }
        if constexpr (F16) {
#pragma unroll
            for (int w = 0; w < 4; ++w) r.kq[w] = reinterpret_cast<const uint4*>(kp_)[w];
#pragma unroll
            for (int w = 0; w < 4; ++w) r.vq[w] = reinterpret_cast<const uint4*>(vp_)[w];
        } else {
            r.kq[0] = *reinterpret_cast<const uint4*>(kp_);
            r.vq[0] = *reinterpret_cast<const uint4*>(vp_);
            r.kpar = *kpar_;
            r.vpar = *vpar_;
        }
    };
    // (third session of round 6) ALLV: every row of the step lies inside the sequence — true for every step of the steady-state loop (it stops two
    // strides short of the length) — so the selects that mask rows beyond it fold away: ~17 of the ~290 VALU instructions of a four-head step.
    auto step = [&](int64_t base, const Rows& r, auto allv) {
#ifndef KV_ALLV
#define KV_ALLV 1
#endif
        constexpr bool ALLV = KV_ALLV && decltype(allv)::value;
        const bool valid = ALLV || base + slot < seq_len;
        float x[QG], vs = 1.0f, vz = 0.0f;
        if constexpr (F16) {
            f16x8 kb[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) kb[w] = __builtin_bit_cast(f16x8, r.kq[w]);
            if constexpr (QG > 1) {
                const f32x4 c4 = kv_qk_mfma4(qa, kb);
#pragma unroll
                for (int g = 0; g < QG; ++g) x[g] = valid ? c4[g] * sm_scale : -INFINITY;
            } else {
                x[0] = valid ? kv_qk_mfma<QL>(qa, kb) * sm_scale : -INFINITY;
            }
        } else {
            const float ks = (float)__builtin_bit_cast(f16, (unsigned short)(r.kpar & 0xFFFF)), kz = (float)__builtin_bit_cast(f16, (unsigned short)(r.kpar >> 16));
            vs = (float)__builtin_bit_cast(f16, (unsigned short)(r.vpar & 0xFFFF)), vz = (float)__builtin_bit_cast(f16, (unsigned short)(r.vpar >> 16));
            if constexpr (QG > 1) {
                const uint32_t kw[4] = {r.kq[0].x, r.kq[0].y, r.kq[0].z, r.kq[0].w};
                f16x8 kb[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    uint32_t u[4];
                    kv_unpack8(kw[w], ebits, u);
                    kb[w] = __builtin_bit_cast(f16x8, u32x4{u[0], u[1], u[2], u[3]});
                }
                const f32x4 c4 = kv_qk_mfma4(qa, kb);
#pragma unroll
                for (int g = 0; g < QG; ++g) {
                    const float dotn = c4[g] - qoff[g];
                    x[g] = valid ? (ks * dotn - kz * qsum[g]) * sm_scale : -INFINITY;
                }
            } else {
                const float dotn = kv_qk<QL>(qa, r.kq[0], ebits) - qoff[0];
                x[0] = valid ? (ks * dotn - kz * qsum[0]) * sm_scale : -INFINITY;
            }
        }
        // The reference point of a lane's running softmax is not its exact maximum but a bound KV_MARGIN (log2 units) above the row
        // that last raised it: exact maxima move in most steps of a 2048-token request (16 row lanes per wave, each a new maximum
        // with probability 1 / t) and every move rescales 34 registers in all lanes; a bound 2^8 above moves once or twice per
        // request. p = exp2(x - m) stays <= 1, the merge takes m as it is.
        if constexpr (PVM) {
            f16x4 ah;
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                if (__builtin_amdgcn_ballot_w64(x[g] > m[g]) != 0) {   // (wave-uniform m: everybody moves, to a bound above the wave's largest score)
                    const float m_new = fq_wave_max(x[g]) + KV_MARGIN;
                    const float alpha = __builtin_amdgcn_exp2f(m[g] - m_new);
                    d[g] *= alpha;
                    zacc[g] *= alpha;
#pragma unroll
                    for (int t = 0; t < 8; ++t) oacc[t][g] *= alpha;       // (D row g of every tile: head g — meaningful in lanes 0 .. 15)
                    m[g] = m_new;
                }
                const float pg = valid ? __builtin_amdgcn_exp2f(x[g] - m[g]) : 0.0f;
                d[g] += pg;
                const f16 a16 = (f16)((valid ? pg * vs : 0.0f) * 256.0f);   // 2^KV_MARGIN: a row at the reference maximum weighs ~s
                ah[g] = a16;
                zacc[g] += __builtin_fmaf((float)a16 * 0.00390625f, KV_OFF, valid ? pg * vz : 0.0f);
            }
            if (part == 0) *reinterpret_cast<f16x4*>(pv_pt + slot * 32) = ah;       // row `slot`, heads 0 .. 3 (columns 4 .. 15 stay zero)
            const uint32_t vw[4] = {r.vq[0].x, r.vq[0].y, r.vq[0].z, r.vq[0].w};
#pragma unroll
            for (int w = 0; w < 4; ++w) {      // dword w's eight values (feature order KV_PERM) -> tile 2 part + w / 2, row `slot`, columns 8 (w % 2) ..
                uint32_t u[4];
                kv_unpack8(vw[w], ebits, u);
// (the strided 16-byte writes conflict two ways; swapping the halves of rows 4..7 / 12..15 — un-swizzled by the readers' own addresses — took
                //  SQ_LDS_BANK_CONFLICT from 2.69 M to 0.59 M per launch and the launch from 33.8 to 34.2 us: LDS is not what the step waits for. Not kept: r06c64)
                *reinterpret_cast<u32x4*>(pv_vt + (2 * part + (w >> 1)) * 512 + slot * 32 + (w & 1) * 16) = u32x4{u[0], u[1], u[2], u[3]};
            }
            const f16x4 pa = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4*)(pv_pt + 8 * lane)));
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const f16x4 vb = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4*)(pv_vt + t * 512 + 8 * lane)));
                oacc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(pa, vb, oacc[t], 0, 0, 0);
            }
            return;
        }
        float pr[QG];
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            const float m_new = x[g] > m[g] ? x[g] + KV_MARGIN : m[g];
            if (__builtin_amdgcn_ballot_w64(x[g] > m[g]) != 0) {   // some lane's bound moved: everybody rescales (by 1 where it did not)
                const float alpha = x[g] > m[g] ? __builtin_amdgcn_exp2f(m[g] - m_new) : 1.0f;
                d[g] *= alpha;
                zacc[g] *= alpha;
#pragma unroll
                for (int j = 0; j < 32; ++j) acc[g][j] *= alpha;
                m[g] = m_new;
            }
            pr[g] = valid ? __builtin_amdgcn_exp2f(x[g] - m[g]) : 0.0f;
            d[g] += pr[g];
        }
        if constexpr (F16 && ALLV) {
            // gfx940+ VALU-trans-use hazard: the result of v_exp_f32 may not be read by the NEXT VALU instruction. The compiler pads its own
            // instructions but not an asm statement's (cdna_hip_programming.md 5.7), and here — no select between the exponential and the
            // v_fma_mix_f32 asm that multiplies by it any more — the first FMA of a step read a stale p: six NaNs in 4096 outputs at 2048 rows.
#pragma unroll
            for (int g = 0; g < QG; ++g) asm volatile("s_nop 1" : "+v"(pr[g]));
        }
        if constexpr (F16) {
            // (a masked row's VALUES are whatever the page holds, maybe NaN: they must not meet p = 0 in an FMA)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t vw[4] = {r.vq[w].x, r.vq[w].y, r.vq[w].z, r.vq[w].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t pair = (!UNI || valid) ? vw[e] : 0u;
#pragma unroll
                    for (int g = 0; g < QG; ++g) {
                        kv_fma_half<0>(acc[g][w * 8 + 2 * e], pair, pr[g]);
                        kv_fma_half<1>(acc[g][w * 8 + 2 * e + 1], pair, pr[g]);
                    }
                }
            }
        } else {
            float pvs[QG];
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                pvs[g] = valid ? pr[g] * vs : 0.0f;   // (UNI: a masked row's parameters are whatever the page holds)
                const float pvz = valid ? pr[g] * vz : 0.0f;
                zacc[g] += __builtin_fmaf(pvs[g], KV_OFF, pvz);
            }
            const uint32_t vw[4] = {r.vq[0].x, r.vq[0].y, r.vq[0].z, r.vq[0].w};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                uint32_t u[4];
                kv_unpack8(vw[w], ebits, u);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < QG; ++g) {
                        kv_fma_half<0>(acc[g][w * 8 + 2 * e], u[e], pvs[g]);
                        kv_fma_half<1>(acc[g][w * 8 + 2 * e + 1], u[e], pvs[g]);
                    }
            }
        }
    };
    // NB - 1 steps of rows are in flight under a step's arithmetic (a step is 2.1 KB per wave; 16 waves per CU). The steady-state loop
    // is free of conditions on purpose: the compiler counts s_waitcnt vmcnt over EVERY path to a use, and one conditional request
    // makes it wait for all but the shortest path's loads — i.e. for the prefetched rows (seen in the first build: vmcnt(3) in front
    // of the MFMA with twelve loads in flight). Head and tail (at most 2 (NB - 1) steps) run the conditional form.
    constexpr int NB = KV_DEPTH + 1;
    Rows buf[NB];
    int64_t base = (KV_ABL & 1) ? seq_len : (int64_t)vwave * RPW;
    if (base + (int64_t)2 * (NB - 1) * STRIDE < seq_len) {
#pragma unroll
        for (int i = 0; i < NB - 1; ++i) request(base + (int64_t)i * STRIDE, buf[i]);
        for (; base + (int64_t)2 * (NB - 1) * STRIDE < seq_len; base += (int64_t)NB * STRIDE) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                request(base + (int64_t)(i + NB - 1) * STRIDE, buf[(i + NB - 1) % NB]);
                step(base + (int64_t)i * STRIDE, buf[i], std::true_type{});
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < NB - 1; ++i)
            if (base + (int64_t)i * STRIDE < seq_len) request(base + (int64_t)i * STRIDE, buf[i]);
    }
    // the last (at most 2 (NB - 1)) steps: step j lives in buf[j % NB], the first NB - 1 are in flight
#pragma unroll
    for (int j = 0; j < 2 * (NB - 1); ++j) {
        if (j + NB - 1 < 2 * (NB - 1) && base + (int64_t)(j + NB - 1) * STRIDE < seq_len)
            request(base + (int64_t)(j + NB - 1) * STRIDE, buf[(j + NB - 1) % NB]);
        if (base + (int64_t)j * STRIDE < seq_len) {
            if constexpr (APPEND) {
                // (the steady-state loop above never reaches the last row: it stops two strides short of the length)
                if (own && base + (int64_t)j * STRIDE + slot == last_row) {
                    Rows& r = buf[j % NB];
                    r.kq[0] = reinterpret_cast<const uint4*>(s_new)[part];
                    r.vq[0] = reinterpret_cast<const uint4*>(s_new + 16)[part];
                    r.kpar = s_new[32], r.vpar = s_new[33];
                }
            }
            step(base + (int64_t)j * STRIDE, buf[j % NB], std::false_type{});
        }
    }
    if constexpr (PVM) {
        // one state per wave and head: m (uniform), the denominators and zero-point sums of the rows' part-0 lanes, O in lanes 0 .. 15
        float dt[QG], zt[QG];
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            dt[g] = fq_wave_sum(part == 0 ? d[g] : 0.0f);
            zt[g] = fq_wave_sum(part == 0 ? zacc[g] : 0.0f);
        }
        __syncthreads();                                         // every wave is done with its tiles: the states below take their place
        float* st_m = reinterpret_cast<float*>(kv_smem);         // [NW][4]
        float* st_d = st_m + NW * 4;                             // [NW][4]
        float (*st_o)[HD] = reinterpret_cast<float (*)[HD]>(st_d + NW * 4);   // [NW * 4][HD]
        if (lane == 0) {
#pragma unroll
            for (int g = 0; g < QG; ++g) st_m[wave * 4 + g] = m[g], st_d[wave * 4 + g] = dt[g];
        }
        if (lane < 16) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int f = 32 * (t >> 1) + 8 * (2 * (t & 1) + (lane >> 3)) + KV_PERM[lane & 7];   // column `lane` of tile t (see the tile writes)
#pragma unroll
                for (int g = 0; g < QG; ++g) st_o[wave * 4 + g][f] = oacc[t][g] * 0.00390625f - zt[g];
            }
        }
        __syncthreads();
        for (int idx = tid; idx < QG * HD; idx += NW * 64) {
            const int g = idx / HD, f = idx - g * HD;
            float mm = st_m[g];
#pragma unroll
            for (int w = 1; w < NW; ++w) mm = fmaxf(mm, st_m[w * 4 + g]);
            float oo = 0.0f, dd = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float mw = st_m[w * 4 + g];
                const float wt = mw == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(mw - mm);
                oo = __builtin_fmaf(st_o[w * 4 + g][f], wt, oo);
                dd = __builtin_fmaf(st_d[w * 4 + g], wt, dd);
            }
            if constexpr (!SPLIT) {
                const size_t oi = transpose_out ? ((size_t)b * HD + f) * QH + head + g : ((size_t)b * QH + head + g) * HD + f;
                o[oi] = dd > 0.0f ? (f16)(oo / dd) : (f16)0.0f;
            } else {      // the workgroup's state of head g for the pair's last arriver (agent-scope stores: see the hand-over of the lane-state form below)
                float* mine = kv_states(ws, (size_t)gridDim.x * QH) + (((size_t)b * QH + head + g) * gridDim.z + blockIdx.z) * (HD + 2);
                if (f == 0) {
                    __hip_atomic_store(mine, mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(mine + 1, dd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __hip_atomic_store(mine + 2 + f, oo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if constexpr (SPLIT) {
            // the same hand-over as below, for the four heads at once: states in memory (vmcnt), then one arrival per head; the last arriver of a head merges
            const int S = (int)gridDim.z;
            unsigned* s_lastv = reinterpret_cast<unsigned*>(&st_o[NW * 4][0]);      // [4], behind the states
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid < QG)
                s_lastv[tid] = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(ws) + (size_t)b * QH + head + tid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(S - 1);
            __syncthreads();
            for (int idx = tid; idx < QG * HD; idx += NW * 64) {
                const int g = idx / HD, f = idx - g * HD;
                if (!s_lastv[g]) continue;
                const float* all = kv_states(ws, (size_t)gridDim.x * QH) + ((size_t)b * QH + head + g) * S * (HD + 2);
                auto ld = [](const float* ptr) { return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
                float mz[16], dz[16], oz[16];
#pragma unroll
                for (int z = 0; z < 16; ++z) {
                    const float* st_z = all + (size_t)(z < S ? z : 0) * (HD + 2);
                    mz[z] = ld(st_z), dz[z] = ld(st_z + 1), oz[z] = ld(st_z + 2 + f);
                }
                float mm = -INFINITY;
#pragma unroll
                for (int z = 0; z < 16; ++z) mm = z < S ? fmaxf(mm, mz[z]) : mm;
                float dd = 0.0f, oo = 0.0f;
#pragma unroll
                for (int z = 0; z < 16; ++z) {
                    const float wt = (z >= S || mz[z] == -INFINITY) ? 0.0f : __builtin_amdgcn_exp2f(mz[z] - mm);
                    dd += dz[z] * wt;
                    oo += oz[z] * wt;
                }
                const size_t oi = transpose_out ? ((size_t)b * HD + f) * QH + head + g : ((size_t)b * QH + head + g) * HD + f;
                o[oi] = dd > 0.0f ? (f16)(oo / dd) : (f16)0.0f;
            }
            __syncthreads();
            if (tid < QG && s_lastv[tid])
                __hip_atomic_store(reinterpret_cast<unsigned*>(ws) + (size_t)b * QH + head + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < QG; ++g) {   // (QG > 1: the heads' states go through the same LDS one after the other)
        const int st = wave * RPW + slot;
        if (part == 0) {
            s_m[st] = m[g];
            s_d[st] = d[g];
        }
    #pragma unroll
        for (int w = 0; w < 4; ++w)
    #pragma unroll
            for (int e = 0; e < 8; ++e) s_o[st][part * 32 + w * 8 + feat(e)] = acc[g][w * 8 + e] - zacc[g];
        __syncthreads();
        // state.cuh merge: every partial state rescaled to the common maximum. Rounds 1-5 let 128 threads walk all NS states (two serial loops
        // of NS LDS round trips): 7 of the 10.8 us of a launch over 64 cached tokens (ablation builds, tools/gpu_call.sh r05c29). Now (a) the
        // maximum by a wave reduction, every state's weight computed once by thread s; (b) ALL threads sum: thread (feature f, chunk c) the
        // states of chunk c; (c) the NCH partial sums per feature.
        constexpr int THREADS = NW * 64, NCH = THREADS / HD, CH = NS / NCH;
        static_assert(NCH >= 1 && NS % NCH == 0 && NS <= THREADS, "merge geometry");
        float* s_w = s_q + QG * HD;             // [NS] weights
        float* s_red = s_w + NS;                // [NW] wave maxima
        float* s_pd = s_red + NW;               // [NCH] partial denominators
        float (*s_po)[HD + 1] = reinterpret_cast<float (*)[HD + 1]>(s_pd + NCH);   // [NCH][HD + 1] partial numerators
        const float mloc = tid < NS ? s_m[tid] : -INFINITY;
        const float wmax = fq_wave_max(mloc);
        if (lane == 0) s_red[wave] = wmax;
        __syncthreads();
        float mm = s_red[0];
    #pragma unroll
        for (int w = 1; w < NW; ++w) mm = fmaxf(mm, s_red[w]);
        if (tid < NS) {
            const float w = mloc == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(mloc - mm);
            s_w[tid] = w;
            s_d[tid] *= w;
        }
        __syncthreads();
        {
            const int f = tid % HD, c = tid / HD;
            float oo = 0.0f, dd = 0.0f;
            if (!(KV_ABL & 2)) {
    #pragma unroll 8
                for (int s = c * CH; s < (c + 1) * CH; ++s) {
                    oo = __builtin_fmaf(s_o[s][f], s_w[s], oo);
                    dd += s_d[s];
                }
            }
            s_po[c][f] = oo;
            if (f == 0) s_pd[c] = dd;
        }
        __syncthreads();
        if (tid < HD) {
            float dd = 0.0f, oo = 0.0f;
    #pragma unroll
            for (int c = 0; c < NCH; ++c) {
                dd += s_pd[c];
                oo += s_po[c][tid];
            }
            const size_t oi = transpose_out ? ((size_t)b * HD + tid) * QH + head + g : ((size_t)b * QH + head + g) * HD + tid;
            if (!SPLIT) {
                o[oi] = dd > 0.0f ? (f16)(oo / dd) : (f16)0.0f;  // an empty sequence attends to nothing: zeros, not 0/0
            } else {
                // Agent-scope stores (sc1: written through to memory — the XCDs' L2s are not coherent with each other) instead of plain
                // stores + __threadfence(): the fence is a write-back of the XCD's whole L2 per workgroup (buffer_wbl2), measured +20 us
                // on a 26 us launch.
                float* mine = kv_states(ws, (size_t)gridDim.x * QH) + (((size_t)b * QH + head + g) * gridDim.z + blockIdx.z) * (HD + 2);
                if (tid == 0) {
                    __hip_atomic_store(mine, mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(mine + 1, dd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __hip_atomic_store(mine + 2 + tid, oo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (SPLIT) {
            // The hand-over below (relaxed agent-scope atomics, s_waitcnt vmcnt(0), barrier, relaxed fetch_add) has no release / acquire pair: it is
            // correct because gfx9 counts stores in vmcnt and the sc1 (agent-scope) stores write through to memory the merging workgroup's sc1 loads
            // read. A target where stores retire on another counter (gfx10+: vscnt) needs the fences back — this file is gfx942 / gfx950 only.
    #if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
    #error "fq_kv_decode_kernel<SPLIT>: the vmcnt-based hand-over is only valid on gfx942 / gfx950"
    #endif
            const int S = (int)gridDim.z;
            unsigned* cnt = reinterpret_cast<unsigned*>(ws) + (size_t)b * QH + head + g;
            __shared__ unsigned s_last;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this workgroup's state has reached memory ...
            __syncthreads();
            if (tid == 0) s_last = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(S - 1);   // ... before it is counted
            __syncthreads();
            if (s_last) {
                if (tid < HD) {
                    const float* all = kv_states(ws, (size_t)gridDim.x * QH) + ((size_t)b * QH + head + g) * S * (HD + 2);
                    auto ld = [](const float* ptr) { return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };   // (sc1: not this XCD's L2)
                    float mz[16], dz[16], oz[16];      // every load first (S <= 16 round trips side by side, not one after the other)
    #pragma unroll
                    for (int z = 0; z < 16; ++z) {
                        const float* st_z = all + (size_t)(z < S ? z : 0) * (HD + 2);
                        mz[z] = ld(st_z), dz[z] = ld(st_z + 1), oz[z] = ld(st_z + 2 + tid);
                    }
                    float mm = -INFINITY;
    #pragma unroll
                    for (int z = 0; z < 16; ++z) mm = z < S ? fmaxf(mm, mz[z]) : mm;
                    float dd = 0.0f, oo = 0.0f;
    #pragma unroll
                    for (int z = 0; z < 16; ++z) {
                        const float w = (z >= S || mz[z] == -INFINITY) ? 0.0f : __builtin_amdgcn_exp2f(mz[z] - mm);
                        dd += dz[z] * w;
                        oo += oz[z] * w;
                    }
                    const size_t oi = transpose_out ? ((size_t)b * HD + tid) * QH + head + g : ((size_t)b * QH + head + g) * HD + tid;
                    o[oi] = dd > 0.0f ? (f16)(oo / dd) : (f16)0.0f;
                }
                if (tid == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next launch finds the counters as this one did
            }
        }

        if (QG > 1) __syncthreads();   // the next head's states overwrite s_m / s_d / s_o / s_w
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
    p.copies = 1;
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

// Split form: how many workgroups share a (request, head) pair, and the workspace they meet in. One pair per workgroup above 128
// pairs; below, as many as bring the launch to ~256 eight-wave workgroups (one per CU) — but no split shorter than 256 rows when the
// caller knows the sequence length (seq_hint > 0): a workgroup's fixed work (the query, the merge of its 128 lane states) is worth ~4 steps.
int fq_kv_decode_splits(int batch, int num_heads, int seq_hint) {
    const int64_t pairs = (int64_t)batch * num_heads;
    if (pairs <= 0 || pairs > 128) return 1;
#ifndef KV_SPLIT_WGS
#define KV_SPLIT_WGS 256
#endif
    int s = (int)(KV_SPLIT_WGS / pairs);   // one 8-wave workgroup per CU
    if (s > 16) s = 16;
    if (seq_hint > 0) {
        const int by_len = seq_hint / 256;
        if (s > by_len) s = by_len;
    }
    return s < 2 ? 1 : s;
}
// The shared-cache launch (fq_kv_batch_decode_gqa): workgroups are (request, KV head) pairs where the query heads of a KV head share one
// (head_dim 128, groups of 2 / 4) — that count decides the split; the workspace holds a counter and up to 16 states per QUERY head either way.
// Merged where it pays (measured, profiles/r06_gqa_cache.txt: a Llama-3-8B step 178 -> 161 us at 64 requests, 293 -> 261 at 128, but 65 -> 79 at
// one request and 102 -> 111 at sixteen: a quarter of the workgroups, each with four times the p . v arithmetic): from one merged workgroup per CU on.
#ifndef KV_MERGE_MIN_PAIRS
#define KV_MERGE_MIN_PAIRS 32    // (third session: with p . v of the merged launch on the matrix pipe, its split hand-over, four waves per SIMD and the one-pass query transform the merged
                                 // launch wins from 4 requests x 8 KV heads on: the step at 4 / 6 / 10 / 12 / 14 / 16 / 24 requests 70.0 / 76.0 / 90.5 / 89.4 / 94.0 / 95.0 / 107.9 -> 68.8 / 72.7 / 82.9 / 84.4 / 87.6 / 90.3 / 98.6 us
                                 // (r06c55, r06c57, r06c81, r06c82); at 2 requests level, at one a loss)
#endif
#ifndef KV_MERGE_QG
#define KV_MERGE_QG 0      // (measurement knob) 2: a group of four as TWO workgroups of two query heads (152 VGPRs: three waves per SIMD instead of two)
#endif
int fq_kv_decode_wg_heads(int batch, int num_q_heads, int q_group, int head_dim) {
    // (a group of EIGHT — Llama-2/3-70B — runs as two workgroups of four query heads: the kernel only needs its QG heads to lie inside one group)
    const bool merge = KV_MERGE_HEADS && head_dim == 128 && (q_group == 2 || q_group == 4 || q_group == 8) && (int64_t)batch * (num_q_heads / q_group) >= KV_MERGE_MIN_PAIRS;
    if (KV_MERGE_QG == 2 && merge && q_group == 4) return num_q_heads / 2;
    if (merge && q_group == 8) return num_q_heads / 4;
    return merge ? num_q_heads / q_group : num_q_heads;
}
int64_t fq_kv_decode_ws_bytes_gqa(int batch, int num_q_heads, int q_group, int head_dim) {
    const int64_t wg_pairs = (int64_t)batch * fq_kv_decode_wg_heads(batch, num_q_heads, q_group, head_dim), pairs = (int64_t)batch * num_q_heads;
    if (wg_pairs <= 0 || wg_pairs > 128) return 0;
    return ((pairs + 3) & ~(int64_t)3) * (int64_t)sizeof(unsigned) + pairs * 16 * (head_dim + 2) * (int64_t)sizeof(float);
}
int64_t fq_kv_decode_ws_bytes(int batch, int num_heads, int head_dim) {   // for ANY split count this library chooses (<= 16)
    const int64_t pairs = (int64_t)batch * num_heads;
    if (pairs <= 0 || pairs > 128) return 0;
    return ((pairs + 3) & ~(int64_t)3) * (int64_t)sizeof(unsigned) + pairs * 16 * (head_dim + 2) * (int64_t)sizeof(float);
}

int fq_launch_kv_decode(f16* o, const f16* q, void* kv_data, void* kv_param, const int* indptr, const int* indices, const int* last,
                        int num_layers, int layer_idx, int num_heads, int page_size, int head_dim, int batch, const f16* qt,
                        int transpose_out, hipStream_t stream, bool f16_cache, float* ws, int splits, int qgroup, const f16* k_new,
                        const f16* v_new, const void* t_image, int src_heads, int copies) {
    // num_heads: QUERY heads (q, o); qgroup of them share one set of rows; the cache holds (num_heads / qgroup) * copies heads: copies = 1 and
    // qgroup = 1 the reference's layout read head by head, copies = 1 and qgroup = g a cache that holds the KV heads once, copies = qgroup = g the
    // reference's layout (g identical copies per KV head) read ONE copy per group
    if (splits > 16 || qgroup < 1 || num_heads % qgroup || copies < 1 || copies > 8) return -1000;   // (the merge of the split launch reads at most 16 states: fq_kv_decode_splits never returns more)
    PagedKv p = make_kv(kv_data, kv_param, indptr, indices, last, num_layers, layer_idx, num_heads / qgroup * copies, page_size, head_dim, batch);
    p.copies = copies;
    const bool split = ws != nullptr && splits > 1;
    // k_new: the step's own row is quantised and appended by this launch (KvNew): INT4 cache, head_dim 128, page_size % 16 == 0 only
    KvNew nw = {k_new, v_new, reinterpret_cast<const uint4*>(t_image), src_heads};
    if (k_new != nullptr && (f16_cache || head_dim != 128 || page_size % 16 || v_new == nullptr || src_heads < 1 || (num_heads / qgroup * copies) % src_heads)) return -1000;
    const bool append = k_new != nullptr;
    // (round 6) a shared cache with 2 or 4 query heads per KV head at head_dim 128: ONE workgroup per (request, KV head) serves its query heads
    // from one pass over the rows (fq_kv_decode_kernel<.., QG>); every other geometry: a workgroup per query head
    const int wg_heads = fq_kv_decode_wg_heads(batch, num_heads, qgroup, head_dim);
    const int qg = num_heads / wg_heads;
    const dim3 grid((unsigned)batch, (unsigned)wg_heads, split ? (unsigned)splits : 1u);
    const bool wide = (int64_t)batch * wg_heads * (split ? splits : 1) < ((KV_PVM_OCC4 && qg == 4 && !f16_cache) ? KV_PVM_WIDE_MAX + 1 : 512);  // fewer than two workgroups per CU: 8 waves each instead of 4 (measured: 158 -> 113 us at 8 x 8192; no gain from 512 up)
#define FQ_DEC5(HD_, NW_, UNI_, F16_, SP_, QG_, AP_)                                                                   \
    {                                                                                                                 \
        constexpr size_t ns_ = (size_t)(NW_ * (64 / (HD_ / 32))), nch_ = (size_t)NW_ * 64 / HD_;                      \
        constexpr size_t lds = sizeof(float) * (ns_ * (HD_ + 1 + 2) + QG_ * HD_ + ns_ + NW_ + nch_ + nch_ * (HD_ + 1)) + (AP_ ? 256 : 0); \
        FQ_RAISE_LDS_CAP((fq_kv_decode_kernel<HD_, NW_, UNI_, F16_, SP_, QG_, AP_>), lds);                            \
        hipLaunchKernelGGL((fq_kv_decode_kernel<HD_, NW_, UNI_, F16_, SP_, QG_, AP_>), grid, dim3(NW_ * 64), lds, stream, o, q, p, qt, transpose_out, ws, qgroup, nw); \
    }
#define FQ_DEC4(HD_, NW_, UNI_, F16_, SP_, QG_)                                                                        \
    {                                                                                                                 \
        if constexpr (HD_ == 128 && UNI_ && !F16_) {                                                                  \
            if (append) FQ_DEC5(HD_, NW_, UNI_, F16_, SP_, QG_, true) else FQ_DEC5(HD_, NW_, UNI_, F16_, SP_, QG_, false) \
        } else FQ_DEC5(HD_, NW_, UNI_, F16_, SP_, QG_, false)                                                         \
    }
#define FQ_DEC3(HD_, NW_, UNI_, F16_, SP_)                                                                             \
    {                                                                                                                 \
        if constexpr (HD_ == 128) {                                                                                   \
            if (qg == 4) FQ_DEC4(HD_, NW_, UNI_, F16_, SP_, 4) else if (qg == 2) FQ_DEC4(HD_, NW_, UNI_, F16_, SP_, 2) \
            else FQ_DEC4(HD_, NW_, UNI_, F16_, SP_, 1)                                                                \
        } else FQ_DEC4(HD_, NW_, UNI_, F16_, SP_, 1)                                                                  \
    }
#define FQ_DEC2(HD_, NW_, UNI_, F16_)                                                                                  \
    {                                                                                                                 \
        if (split) FQ_DEC3(HD_, NW_, UNI_, F16_, true) else FQ_DEC3(HD_, NW_, UNI_, F16_, false)                      \
    }
#define FQ_DEC(HD_, NW_)                                                                                              \
    {                                                                                                                 \
        const bool uni = page_size % (64 / (HD_ / 32)) == 0;   /* a wave's rows never straddle a page */              \
        if (f16_cache) {                                                                                              \
            if (uni) FQ_DEC2(HD_, NW_, true, true) else FQ_DEC2(HD_, NW_, false, true)                                \
        } else {                                                                                                      \
            if (uni) FQ_DEC2(HD_, NW_, true, false) else FQ_DEC2(HD_, NW_, false, false)                              \
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
#undef FQ_DEC2
#undef FQ_DEC3
#undef FQ_DEC4
#undef FQ_DEC5
    return (int)hipGetLastError();
}
