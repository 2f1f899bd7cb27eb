// fq_kv_common.hpp — the pieces of the KV-cache quantiser (deploy/transformers/kv_cache.py:11-51,268) that more than one file uses: the per-row
// (scale, zero) parameters, the eight-elements-to-a-dword quantiser, the fragment image of the K transform. Moved here from fq_kvquant.hip when
// the decode-attention launch learnt to quantise and append the step's own K / V row (fq_kvcache.hip, round 6): both files must round alike.
#pragma once
#include "fq_common.hpp"

namespace fqkv {

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

}  // namespace fqkv
