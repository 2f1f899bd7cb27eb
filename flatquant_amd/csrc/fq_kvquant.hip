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

// Eight elements (four packed fp16 pairs) -> one dword of unsigned nibbles, the arithmetic of kv_q1 without a division and
// on packed pairs (the pieces of fq_quant8_h16, fq_common.hpp): the fp16 quotient RN16(a / s) is exact from three fp32 fmas on
// r = v_rcp_f32(s) (a and s are fp16 values); rint is the packed add of 1536 (ulp 1 in [1024, 2048): half to even), the
// zero point (an integer <= 15 with lac) is added to that sum exactly, the clamp to [0, 15] is a packed max / min against
// 1536 / 1551, and the digit is the low nibble of each half. Quotients beyond +-512 leave the exact range of the magic add
// on the side they are clamped to. Without lac the zero point (-xmin, not an integer) is added BEFORE the division
// (kv_cache.py:36-43), a packed fp16 add.
template <bool LAC>
__device__ __forceinline__ unsigned kv_q8(uint32_t xa, uint32_t xb, uint32_t xc, uint32_t xd, float r, float s, uint32_t zero2) {
    uint32_t ha, hb, hc, hd;
    float t0, t1, e0, e1;
    if (!LAC)
        asm("v_pk_add_f16 %0, %0, %4\n\tv_pk_add_f16 %1, %1, %4\n\tv_pk_add_f16 %2, %2, %4\n\tv_pk_add_f16 %3, %3, %4"
            : "+v"(xa), "+v"(xb), "+v"(xc), "+v"(xd)
            : "v"(zero2));
#define FQ_KV_PAIR(h, x)                                                                    \
    "v_fma_mix_f32 %[t0], %[" #x "], %[r], 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"          \
    "v_fma_mix_f32 %[t1], %[" #x "], %[r], 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"          \
    "v_fma_mix_f32 %[e0], -%[t0], %[s], %[" #x "] op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"     \
    "v_fma_mix_f32 %[e1], -%[t1], %[s], %[" #x "] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"     \
    "v_fma_f32 %[t0], %[e0], %[r], %[t0]\n\t"                                               \
    "v_fma_f32 %[t1], %[e1], %[r], %[t1]\n\t"                                               \
    "v_cvt_pk_f16_f32 %[" #h "], %[t0], %[t1]\n\t"
    asm(FQ_KV_PAIR(ha, xa) FQ_KV_PAIR(hb, xb) FQ_KV_PAIR(hc, xc) FQ_KV_PAIR(hd, xd)
        : [ha] "=&v"(ha), [hb] "=&v"(hb), [hc] "=&v"(hc), [hd] "=&v"(hd), [t0] "=&v"(t0), [t1] "=&v"(t1), [e0] "=&v"(e0),
          [e1] "=&v"(e1)
        : [xa] "v"(xa), [xb] "v"(xb), [xc] "v"(xc), [xd] "v"(xd), [r] "v"(r), [s] "v"(s));
#undef FQ_KV_PAIR
    const uint32_t magic2 = 0x66006600u;   // (1536.0h, 1536.0h)
    uint32_t hi2 = 0x660F660Fu;            // (1551.0h, 1551.0h)
    asm volatile("" : "+v"(hi2));
    asm("v_pk_add_f16 %[ha], %[ha], %[mg]\n\tv_pk_add_f16 %[hb], %[hb], %[mg]\n\t"
        "v_pk_add_f16 %[hc], %[hc], %[mg]\n\tv_pk_add_f16 %[hd], %[hd], %[mg]"
        : [ha] "+v"(ha), [hb] "+v"(hb), [hc] "+v"(hc), [hd] "+v"(hd)
        : [mg] "s"(magic2));
    if (LAC)
        asm("v_pk_add_f16 %0, %0, %4\n\tv_pk_add_f16 %1, %1, %4\n\tv_pk_add_f16 %2, %2, %4\n\tv_pk_add_f16 %3, %3, %4"
            : "+v"(ha), "+v"(hb), "+v"(hc), "+v"(hd)
            : "v"(zero2));
    uint32_t d, p1, p2, u1, u2;
    const uint32_t sel = 0x06040200u;      // bytes 0 and 2 of the second source, then of the first
    // lac only: without it (x + zero) / scale lies in [0, 15] by construction (x + zero <= fp16(xmax - xmin) = 15 scale
    // up to 2^-11, and >= fp16(xmin - xmin) = 0; the 1e-5 floor only makes the quotients smaller)
    if (LAC)
        asm("v_pk_max_f16 %[ha], %[ha], %[mg]\n\tv_pk_max_f16 %[hb], %[hb], %[mg]\n\t"
            "v_pk_max_f16 %[hc], %[hc], %[mg]\n\tv_pk_max_f16 %[hd], %[hd], %[mg]\n\t"
            "v_pk_min_f16 %[ha], %[ha], %[hi]\n\tv_pk_min_f16 %[hb], %[hb], %[hi]\n\t"
            "v_pk_min_f16 %[hc], %[hc], %[hi]\n\tv_pk_min_f16 %[hd], %[hd], %[hi]"
            : [ha] "+v"(ha), [hb] "+v"(hb), [hc] "+v"(hc), [hd] "+v"(hd)
            : [mg] "s"(magic2), [hi] "v"(hi2));
    asm("v_perm_b32 %[p1], %[hb], %[ha], %[sel]\n\t"        // low bytes of e0, e1, e2, e3
        "v_perm_b32 %[p2], %[hd], %[hc], %[sel]\n\t"        // low bytes of e4 .. e7
        "v_lshrrev_b32_e32 %[u1], 4, %[p1]\n\t"
        "v_lshrrev_b32_e32 %[u2], 4, %[p2]\n\t"
        "v_bfi_b32 %[p1], %[m4], %[u1], %[p1]\n\t"          // byte 0 = n0 | n1 << 4, byte 2 = n2 | n3 << 4
        "v_bfi_b32 %[p2], %[m4], %[u2], %[p2]\n\t"
        "v_perm_b32 %[d], %[p2], %[p1], %[sel]"
        : [d] "=v"(d), [p1] "=&v"(p1), [p2] "=&v"(p2), [u1] "=&v"(u1), [u2] "=&v"(u2), [ha] "+v"(ha), [hb] "+v"(hb),
          [hc] "+v"(hc), [hd] "+v"(hd)
        : [sel] "s"(sel), [m4] "v"(0x00F000F0u));
    return d;
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

// The A fragments of matrix^T for fq_kv_quant_kernel / fq_rowmm_kernel: A row fc <-> output column n, eight consecutive k per lane — a
// COLUMN piece of the row-major matrix. Gathered from global memory that is eight 2-byte loads 2 HD bytes apart per fragment, 64
// latency-bound loads per thread: ~8 us in front of every launch, 10.7 us for the 128 rows of a decode step (rocprofv3, tools/gpu_call.sh
// r05c18). Now the matrix goes through LDS: coalesced 16-byte loads of HD / 2 rows at a time (row pitch HD + 2 halfwords: consecutive
// rows start one bank apart), the column pieces are gathered from there.
template <int HD, typename T>
__device__ __forceinline__ void kv_stage_tfrag(const T* __restrict__ Tm, uint4* tfrag, unsigned short* raw, int tid) {
    constexpr int KS = HD / 16, NTL = HD / 32, PITCH = HD + 2, HALF = HD / 2, CPR = HD / 8, NLD = HALF * CPR / 256;
    static_assert(HALF * CPR % 256 == 0, "whole 16-byte loads per thread");
    uint4 pre[2][NLD];   // both halves requested at once: one global round trip in front of the launch's first MFMA, not two
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int k2 = 0; k2 < NLD; ++k2) pre[half][k2] = reinterpret_cast<const uint4*>(Tm + (size_t)(half * HALF) * HD)[tid + 256 * k2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int k2 = 0; k2 < NLD; ++k2) {
            const int i = tid + 256 * k2;
            const int k = i / CPR, c8 = i - k * CPR;
            const uint4 v = pre[half][k2];
            unsigned* dst = reinterpret_cast<unsigned*>(raw + k * PITCH + c8 * 8);   // (4-byte aligned: PITCH is even)
            dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = v.w;
        }
        __syncthreads();
        for (int item = tid + half * (KS / 2) * NTL * 64; item < (half + 1) * (KS / 2) * NTL * 64; item += 256) {
            const int f = item >> 6, ln = item & 63, fh = ln >> 5, fc = ln & 31;
            const int s = f / NTL, nt = f - s * NTL;
            const int n = nt * 32 + 16 * ((fc >> 2) & 1) + 4 * (fc >> 3) + (fc & 3);  // output column of A row fc
            const unsigned short* src = raw + ((s * 16 - half * HALF) + fh * 8) * PITCH + n;
            unsigned short e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = src[j * PITCH];
            tfrag[item] = uint4{(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                                (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
        }
        __syncthreads();
    }
}

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
        uint8_t* qdst[4];
        unsigned* pdst[4];
        const int ndst = dn.n;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
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
