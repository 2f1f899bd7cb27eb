// fq_hadamard_reg.hip — online Hadamard rotation with the FWHT done in REGISTERS (no LDS butterfly passes):
//     y = hadK [K,K] @ FWHT_P( x.view(rows, K, P) ) * scale,   P = 512 * CH, CH a power of two.
// Same contract and arithmetic as fq_hadamard.hip (hadamard_utils.py:132-141, deploy/functional/online_trans.py:
// 144-151), which stays as the fallback for every other shape.
//
// A wave transforms one length-P vector. Lane l holds, for each 16-byte chunk j < CH, the 8 consecutive elements
// j*512 + 8 l .. +7 (one coalesced 1 KB load per chunk), so element-index bits 0-2 are registers of a chunk, bits 3-8
// are the lane number and bits 9+ are the chunk number. The butterfly stages run in ascending bit order — the order of
// hadamard_utils.py:94-101 and of the fast_hadamard_transform kernel, so the fp32 results are bit-identical to the
// oracle's fwht_f32:
//   bits 0-2  in registers;
//   bits 3-8  ACROSS LANES, partner = lane ^ 2^s: quad_perm DPP (s = 0, 1), half-mirror + quad reverse (s = 2),
//             row rotate by 8 (s = 3), v_permlane16_swap / v_permlane32_swap (s = 4, 5). A lane whose bit s is set
//             needs partner - own: its own value gets the sign bit flipped first (exact), then both kinds of lane add;
//   bits 9+   in registers, between chunks.
// K == 1 (n = 512 .. 8192): a wave per row, nothing but registers between the load and the store.
// K  > 1 (n = K * 512 or K * 1024, e.g. 14336 = 28 x 512, 28672 = 28 x 1024): a 4-wave workgroup per row; wave w
//   transforms the sub-vectors w, w+4, ... and drops them as fp16 into an LDS image [k][p] with plain 16-byte
//   writes; the K x K factor then runs on the matrix cores exactly as in fq_hadamard.hip (A = 32 positions x 16 k,
//   B = hadK^T fragments), its A fragments read k-strided out of the row-major image by ds_read_b64_tr_b16 (the
//   hardware transpose read: lane i of a 16-lane group receives column i of the 4 x 16 block its group addresses).
#include "fq_common.hpp"

namespace {

__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float flip(float v, unsigned m) {
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) ^ m);
}

// per-lane constants of the six cross-lane stages: sign-bit masks (0x80000000 where bit s of the lane number is set)
// for the stages that flip-and-add, +-1.0f for the two that use an fma
struct XMask {
    unsigned m[2];
    float sgn[2];
    __device__ __forceinline__ explicit XMask(int lane) {
        m[0] = (lane & 1) ? 0x80000000u : 0u;
        m[1] = (lane & 2) ? 0x80000000u : 0u;
        sgn[0] = (lane & 16) ? -1.0f : 1.0f;
        sgn[1] = (lane & 32) ? -1.0f : 1.0f;
    }
};

// the six cross-lane butterfly stages on one register (14 VALU). "s_nop 1": a VALU result needs two wait states
// before a DPP / permlane read of it.
template <int XS = 6>  // XS: how many of the stages (lane ^ 1, 2, 4, 8, 16, 32) to run: vectors of 8 * 2^XS elements
__device__ __forceinline__ float xlane(float v, const XMask& k) {
    v = flip(v, k.m[0]) + dppf<0xB1>(v);  // lane ^ 1: quad_perm [1,0,3,2]; lanes with the bit set need partner - own
    v = flip(v, k.m[1]) + dppf<0x4E>(v);  // lane ^ 2: quad_perm [2,3,0,1]
    if (XS == 3) {  // lane ^ 4 only
        float r;
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                     "v_sub_f32_dpp %0, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa"
                     : "=&v"(r) : "v"(v));
        return r;
    }
    {   // lane ^ 4 and lane ^ 8: the partner sits one / two banks (groups of 4 lanes) away -> row shifts, one masked add
        // for the lower lanes of each pair (own + partner) and one masked sub for the upper ones (partner - own)
        float r;
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                     "v_sub_f32_dpp %0, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa"
                     : "=&v"(r) : "v"(v));
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %1, %1 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
                     "v_sub_f32_dpp %0, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xc"
                     : "=&v"(v) : "v"(r));
    }
    if (XS == 4) return v;
    {   // lane ^ 16: after the swap, a = value of the lower lane of the pair, b = of the upper one, in BOTH lanes' view
        // (even rows keep a = own and receive b = partner; odd rows receive a = partner and keep b = own)
        float a = v, b = v;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        v = __builtin_fmaf(b, k.sgn[0], a);  // a +- b, one rounding
    }
    if (XS == 5) return v;
    {   // lane ^ 32: same with the wave's halves
        float a = v, b = v;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        v = __builtin_fmaf(b, k.sgn[1], a);
    }
    return v;
}

template <int W>  // radix 2^W butterfly over v[0 .. 2^W), stages in ascending bit order
__device__ __forceinline__ void bfly8(float (&v)[8]) {
#pragma unroll
    for (int b = 0; b < W; ++b) {
        const int s = 1 << b;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (!(i & s)) {
                const float a = v[i], c = v[i + s];
                v[i] = a + c;
                v[i + s] = a - c;
            }
    }
}

// FWHT of a length 512*CH vector held as v[chunk][8] (see the header for the element mapping)
template <int CH, int XS = 6>
__device__ __forceinline__ void fwht_wave(float (&v)[CH][8], const XMask& k) {
#pragma unroll
    for (int j = 0; j < CH; ++j) bfly8<3>(v[j]);
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = xlane<XS>(v[j][e], k);
#pragma unroll
    for (int s = 1; s < CH; s <<= 1)
#pragma unroll
        for (int j = 0; j < CH; ++j)
            if (!(j & s)) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = v[j][e], c = v[j + s][e];
                    v[j][e] = a + c;
                    v[j + s][e] = a - c;
                }
            }
}

template <int CH, bool SILU = false>
__device__ __forceinline__ void load_vec(const f16* __restrict__ p, const f16* __restrict__ p2, int lane, float (&v)[CH][8]) {
    u32x4 r[CH], r2[SILU ? CH : 1];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        r[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + j * 512 + lane * 8));
        if (SILU) r2[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p2 + j * 512 + lane * 8));
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        f16x8 hv = __builtin_bit_cast(f16x8, r[j]);
        if (SILU) hv = fq_silu_mul8(hv, __builtin_bit_cast(f16x8, r2[SILU ? j : 0]));
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = (float)hv[e];
    }
}
template <int CH>
__device__ __forceinline__ void to_f16(const float (&v)[CH][8], float scale, f16x8 (&o)[CH]) {
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) o[j][e] = fq_mul_to_f16(v[j][e], scale);
}

// Fused deploy.nn.Quantizer (deploy/nn/quantization.py:13-36 on the fp16 Hadamard output, lac clip factors): per-row
// extrema of the fp16 results, fp16 scale, fp16 division, round-half-even, clamp, pack. `o` holds a lane's fp16 results.
struct HadQuant {
    float sig_max, sig_min;
    uint8_t* q;   // [rows, n/2]
    f16* scale;   // [rows]
    const f16* up;  // SILU kernels: x is `gate`, the transform's input is fp16(up * fp16(silu(gate))) (fq_silu_mul8)
    float* y32;     // (round 4) != nullptr: the scaled fp32 butterflies are stored as they are — no rounding to fp16 — instead of y
                    // (fast_hadamard_transform on an up-cast input: deploy/nn/online_trans.py:55-59, force_fp32=True)
};
// 8 fp16 results -> one dword of nibbles: packed pairs, exact fp16 quotient without a division (fq_quant8_h16, fq_common.hpp)
__device__ __forceinline__ uint32_t quant8_h(f16x8 v, FqH16Recip rc, bool clamp) {
    const u32x4 xv = __builtin_bit_cast(u32x4, v);
    return clamp ? fq_quant8_h16<true>(xv[0], xv[1], xv[2], xv[3], rc) : fq_quant8_h16<false>(xv[0], xv[1], xv[2], xv[3], rc);
}
__device__ __forceinline__ void minmax8(f16x8 v, float& mx, float& mn) {
    // extrema on packed fp16 pairs (the values are fp16: exact), then the two halves
    f16x2 a = {v[0], v[1]}, b = a;
#pragma unroll
    for (int e = 2; e < 8; e += 2) {
        const f16x2 pr = {v[e], v[e + 1]};
        a = __builtin_elementwise_max(a, pr);
        b = __builtin_elementwise_min(b, pr);
    }
    mx = fq_max3(mx, (float)a[0], (float)a[1]);
    mn = fq_min3(mn, (float)b[0], (float)b[1]);
}

// ---- K == 1: one wave per row (n >= 512), or SUB = 512 / n rows per wave (n = 64, 128, 256: XS cross-lane stages) ----
template <int CH, bool QUANT, bool SILU = false, int SUB = 1>
__global__ __launch_bounds__(256) void fq_had_pow2_kernel(const f16* __restrict__ x, f16* __restrict__ y, int64_t rows,
                                                          float scale, HadQuant hq) {
    static_assert(SUB == 1 || (CH == 1 && !QUANT), "short vectors: one chunk per lane, no fused quantiser");
    constexpr int n = 512 * CH / SUB;
    constexpr int XS = SUB == 1 ? 6 : SUB == 2 ? 5 : SUB == 4 ? 4 : 3;
    const int lane = threadIdx.x & 63;
    const XMask k(lane);
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    if (SUB > 1) {  // consecutive rows are contiguous: a wave's 512 elements are rows SUB * g .. SUB * g + SUB - 1
        const int64_t groups = (rows + SUB - 1) / SUB;
        for (int64_t g = wave_id; g < groups; g += n_waves) {
            const bool ok = g * SUB + lane / (64 / SUB) < rows;
            float v[1][8];
            f16x8 hv = {0};
            if (ok) hv = __builtin_bit_cast(f16x8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(x + g * 512 + lane * 8)));
#pragma unroll
            for (int e = 0; e < 8; ++e) v[0][e] = (float)hv[e];
            fwht_wave<1, XS>(v, k);
            if (!QUANT && hq.y32 != nullptr) {
                if (ok) {
                    f32x4* dst = reinterpret_cast<f32x4*>(hq.y32 + g * 512 + lane * 8);
                    dst[0] = f32x4{v[0][0] * scale, v[0][1] * scale, v[0][2] * scale, v[0][3] * scale};
                    dst[1] = f32x4{v[0][4] * scale, v[0][5] * scale, v[0][6] * scale, v[0][7] * scale};
                }
                continue;
            }
            f16x8 o[1];
            to_f16<1>(v, scale, o);
            if (ok) *reinterpret_cast<uint4*>(y + g * 512 + lane * 8) = __builtin_bit_cast(uint4, o[0]);
        }
        return;
    }
    for (int64_t row = wave_id; row < rows; row += n_waves) {
        float v[CH][8];
        load_vec<CH, SILU>(x + row * n, SILU ? hq.up + row * n : nullptr, lane, v);
        fwht_wave<CH>(v, k);
        if (!QUANT && !SILU && hq.y32 != nullptr) {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                f32x4* dst = reinterpret_cast<f32x4*>(hq.y32 + row * n + j * 512 + lane * 8);
                dst[0] = f32x4{v[j][0] * scale, v[j][1] * scale, v[j][2] * scale, v[j][3] * scale};
                dst[1] = f32x4{v[j][4] * scale, v[j][5] * scale, v[j][6] * scale, v[j][7] * scale};
            }
            continue;
        }
        f16x8 o[CH];
        to_f16<CH>(v, scale, o);
        if (!QUANT) {
#pragma unroll
            for (int j = 0; j < CH; ++j)
                *reinterpret_cast<uint4*>(y + row * n + j * 512 + lane * 8) = __builtin_bit_cast(uint4, o[j]);
        } else {
            float mx = -INFINITY, mn = INFINITY;
#pragma unroll
            for (int j = 0; j < CH; ++j) minmax8(o[j], mx, mn);
            mx = fq_wave_max(mx);
            mn = fq_wave_min(mn);
            const float sc = fq_token_scale<FQ_QUANT_F16>(mx, mn, hq.sig_max, hq.sig_min, FQ_SIG_F16);  // deploy.nn.Quantizer arithmetic
            if (lane == 0) hq.scale[row] = (f16)sc;
            const float rinv = fq_fast_inv(sc);
            const bool clampq = fq_h16_needs_clamp(mx, mn, rinv);
            const FqH16Recip rc = fq_h16_recip(sc);
            uint32_t* qp = reinterpret_cast<uint32_t*>(hq.q + row * (n / 2));
#pragma unroll
            for (int j = 0; j < CH; ++j) qp[j * 64 + lane] = quant8_h(o[j], rc, clampq);  // 8 nibbles = elements j*512 + 8 lane ..
        }
    }
}

// ---- K > 1: a 4-wave workgroup per row, K x K factor on the matrix cores ----
struct KmixGeom {
    int K, KP, KT;  // K, K padded to 16, 32-wide output tiles over k'
};
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int CH, bool QUANT, bool SILU = false, int SUB = 1>
__global__ __launch_bounds__(256) void fq_had_kmix_kernel(const f16* __restrict__ x, f16* __restrict__ y, int64_t rows,
                                                          KmixGeom g, const f16* __restrict__ hadK, float scale,
                                                          HadQuant hq) {
    static_assert(SUB == 1 || (CH == 1 && !QUANT && !SILU), "short sub-vectors: one chunk per lane, plain output");
    constexpr int P = 512 * CH / SUB;  // SUB > 1: P = 256, 128, 64 and a wave transforms SUB consecutive sub-vectors at once
    constexpr int XS = SUB == 1 ? 6 : SUB == 2 ? 5 : SUB == 4 ? 4 : 3;
    constexpr int PITCH = P + 8;  // fp16 elements per image row: +16 bytes skews consecutive k rows across the banks
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f16* V = reinterpret_cast<f16*>(smem);                                   // [KP][PITCH]
    uint4* kfrag = reinterpret_cast<uint4*>(smem + (size_t)g.KP * PITCH * 2);  // [KP/16][KT][64]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = g.K;
    const XMask km(lane);

    // once per workgroup: B fragments of hadK^T (frag (s, kt): B[k = 16 s + 8 h + j][k' = 32 kt + c] = hadK[k'][k]),
    // zero rows K .. KP of the image (their B entries are zero, but 0 * garbage must not be NaN)
    {
        const int nfr = (g.KP / 16) * g.KT * 64;
        for (int item = tid; item < nfr; item += 256) {
            const int f = item >> 6, ln = item & 63, fh = ln >> 5, fc = ln & 31;
            const int s = f / g.KT, kt = f - s * g.KT;
            const int kp = kt * 32 + fc;
            f16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = s * 16 + fh * 8 + j;
                v[j] = (kk < K && kp < K) ? hadK[kp * K + kk] : (f16)0.0f;
            }
            kfrag[item] = __builtin_bit_cast(uint4, v);
        }
        for (int i = K * PITCH + tid * 8; i < g.KP * PITCH; i += 256 * 8)
            *reinterpret_cast<uint4*>(V + i) = make_uint4(0, 0, 0, 0);
    }

    // transpose-read addressing of this lane (see the header): 16-lane group grp = 2 h + (c >> 4), index i in it
    const int grp = lane >> 4, i16 = lane & 15;
    const int q = i16 & 3;  // the 4-column chunk this lane PROVIDES
    const int pcol = 16 * (q & 1) + 8 * (grp & 1) + 4 * (q >> 1);
    const int tr_base = (8 * (grp >> 1) + (i16 >> 2)) * PITCH + pcol;  // element offset; + (16 s + 4 rd) PITCH + 32 pt

    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const f16* xr = x + row * (int64_t)K * P;
        __syncthreads();  // the previous row's fragment reads are done
        const f16* ur = SILU ? hq.up + row * (int64_t)K * P : nullptr;
        u32x4 raw[CH];  // the NEXT sub-vector's chunks are requested before the current one is transformed
        u32x4 raw2[SILU ? CH : 1];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            raw[j] = (SUB == 1 && wave < K) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(xr + (int64_t)wave * P + j * 512 + lane * 8))
                              : u32x4{0, 0, 0, 0};
            if (SILU)
                raw2[j] = wave < K ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ur + (int64_t)wave * P + j * 512 + lane * 8))
                                   : u32x4{0, 0, 0, 0};
        }
        if (SUB > 1) {
            // sub-vectors are contiguous, so group kg (sub-vectors SUB kg ..) is one coalesced 1 KB load like a P = 512
            // vector; lanes of sub-vectors beyond K (last group) stay out of the load and the image
            const int sub = lane / (64 / SUB), sl = lane % (64 / SUB);
            for (int kg = wave; kg * SUB < K; kg += 4) {
                const int k = kg * SUB + sub;
                float v[1][8];
                f16x8 hv = {0};
                if (k < K) hv = __builtin_bit_cast(f16x8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(xr + (int64_t)kg * 512 + lane * 8)));
#pragma unroll
                for (int e = 0; e < 8; ++e) v[0][e] = (float)hv[e];
                fwht_wave<1, XS>(v, km);
                f16x8 o[1];
                to_f16<1>(v, scale, o);
                if (k < K) *reinterpret_cast<uint4*>(V + k * PITCH + sl * 8) = __builtin_bit_cast(uint4, o[0]);
            }
        } else
        for (int k = wave; k < K; k += 4) {
            float v[CH][8];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                f16x8 hv = __builtin_bit_cast(f16x8, raw[j]);
                if (SILU) hv = fq_silu_mul8(hv, __builtin_bit_cast(f16x8, raw2[SILU ? j : 0]));
#pragma unroll
                for (int e = 0; e < 8; ++e) v[j][e] = (float)hv[e];
            }
            if (k + 4 < K) {
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    raw[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(xr + (int64_t)(k + 4) * P + j * 512 + lane * 8));
                    if (SILU)
                        raw2[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ur + (int64_t)(k + 4) * P + j * 512 + lane * 8));
                }
            }
            fwht_wave<CH>(v, km);
            f16x8 o[CH];
            to_f16<CH>(v, scale, o);
#pragma unroll
            for (int j = 0; j < CH; ++j)
                *reinterpret_cast<uint4*>(V + k * PITCH + j * 512 + lane * 8) = __builtin_bit_cast(uint4, o[j]);
        }
        __syncthreads();
        // ---- out^T[p][k'] = sum_k V[k][p] hadK[k'][k]; lane (h, .) ends with p = 32 pt + 16 h + reg of row k' ----
        f16* yp = y + row * (int64_t)K * P;
        const int ksteps = g.KP >> 4;
        constexpr int MAXT = (P / 32) * 2 / 4;  // tiles per wave for KT <= 2 (checked by the launcher for QUANT)
        const int ntiles = (P / 32) * g.KT;
        f16x8 res[QUANT ? MAXT : 1][2];
        float mx = -INFINITY, mn = INFINITY;
        auto tile = [&](int t, f16x8& v0, f16x8& v1) {  // one 32 (p) x 32 (k') output tile -> 16 fp16 per lane
            const int pt = t / g.KT, kt = t - pt * g.KT;
            f32x16 acc = {0};
            for (int s = 0; s < ksteps; ++s) {
                const f16* ap = V + tr_base + (16 * s) * PITCH + 32 * pt;
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap + 4 * PITCH));
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                const s16x8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                acc = mfma32(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, kfrag[(s * g.KT + kt) * 64 + lane]), acc);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v0[e] = (f16)acc[e];
                v1[e] = (f16)acc[8 + e];
            }
        };
        if (!QUANT) {
            for (int t = wave; t < ntiles; t += 4) {
                f16x8 v0, v1;
                tile(t, v0, v1);
                const int pt = t / g.KT, kp = (t - pt * g.KT) * 32 + c;
                if (kp < K) {
                    uint4* op = reinterpret_cast<uint4*>(yp + (int64_t)kp * P + pt * 32 + h * 16);
                    op[0] = __builtin_bit_cast(uint4, v0);
                    op[1] = __builtin_bit_cast(uint4, v1);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < MAXT; ++u) {  // unrolled: res[] is indexed statically (registers)
                const int t = wave + 4 * u;
                if (t < ntiles) {
                    tile(t, res[u][0], res[u][1]);
                    const int pt = t / g.KT, kp = (t - pt * g.KT) * 32 + c;
                    if (kp < K) {
                        minmax8(res[u][0], mx, mn);
                        minmax8(res[u][1], mx, mn);
                    }
                }
            }
        }
        if (QUANT) {
            float* red = reinterpret_cast<float*>(kfrag + (g.KP / 16) * g.KT * 64);  // 8 floats behind the fragments
            mx = fq_wave_max(mx);
            mn = fq_wave_min(mn);
            if (lane == 0) {
                red[wave] = mx;
                red[4 + wave] = mn;
            }
            __syncthreads();  // also: every wave is done reading V -> it becomes the packed-output stage
            mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            mn = fminf(fminf(red[4], red[5]), fminf(red[6], red[7]));
            const float sc = fq_token_scale<FQ_QUANT_F16>(mx, mn, hq.sig_max, hq.sig_min, FQ_SIG_F16);  // deploy.nn.Quantizer arithmetic
            const float rinv = fq_fast_inv(sc);
            const bool clampq = fq_h16_needs_clamp(mx, mn, rinv);
            const FqH16Recip rc = fq_h16_recip(sc);
            unsigned char* obuf = smem;  // [K][P/2] bytes
#pragma unroll
            for (int u = 0; u < MAXT; ++u) {
                const int t = wave + 4 * u;
                const int pt = t / g.KT, kt = t - pt * g.KT;
                const int kp = kt * 32 + c;
                if (t < ntiles && kp < K)
                    *reinterpret_cast<uint2*>(obuf + ((int64_t)kp * P + pt * 32 + h * 16) / 2) =
                        make_uint2(quant8_h(res[u][0], rc, clampq), quant8_h(res[u][1], rc, clampq));
            }
            __syncthreads();
            if (tid == 0) hq.scale[row] = (f16)sc;
            uint4* qp = reinterpret_cast<uint4*>(hq.q + row * ((int64_t)K * P / 2));
            for (int i = tid; i < K * P / 32; i += 256) qp[i] = reinterpret_cast<const uint4*>(obuf)[i];
        }
    }
}

template <int CH, bool QUANT, bool SILU, int SUB = 1>
int launch_pow2(const f16* x, f16* y, int64_t rows, float scale, const HadQuant& hq, int n_cu, hipStream_t stream) {
    int64_t blocks = ((rows + SUB - 1) / SUB + 3) / 4;
    const int64_t cap = (int64_t)n_cu * (CH <= 8 ? 4 : 2);  // 16 / 8 waves per CU
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((fq_had_pow2_kernel<CH, QUANT, SILU, SUB>), dim3((unsigned)blocks), dim3(256), 0, stream, x, y, rows, scale, hq);
    return (int)hipGetLastError();
}

template <int CH, bool QUANT, bool SILU, int SUB = 1>
int launch_kmix(const f16* x, f16* y, int64_t rows, int K, const f16* hadK, float scale, const HadQuant& hq, int n_cu,
                hipStream_t stream) {
    KmixGeom g;
    g.K = K;
    g.KP = (K + 15) / 16 * 16;
    g.KT = (K + 31) / 32;
    if (QUANT && g.KT > 2) return -1000;  // the fused form keeps a wave's tiles in registers (sized for KT <= 2)
    const size_t lds = (size_t)g.KP * (512 * CH / SUB + 8) * 2 + (size_t)(g.KP / 16) * g.KT * 1024 + 64;
    if (lds > 160 * 1024) return -1000;
    auto kern = fq_had_kmix_kernel<CH, QUANT, SILU, SUB>;
    FQ_RAISE_LDS_CAP(kern, 160 * 1024);
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    int64_t blocks = (int64_t)n_cu * per_cu;
    if (blocks > rows) blocks = rows;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, x, y, rows, g, hadK, scale, hq);
    return (int)hipGetLastError();
}

template <bool QUANT, bool SILU = false>
int dispatch_reg(const f16* x, f16* y, int64_t rows, int n, int K, const f16* hadK, float scale, const HadQuant& hq,
                 int n_cu, hipStream_t stream) {
    if (K < 1 || n % K) return -1000;
    const int P = n / K;
    if (K == 1) {
        switch (P) {
            case 512: return launch_pow2<1, QUANT, SILU>(x, y, rows, scale, hq, n_cu, stream);
            case 1024: return launch_pow2<2, QUANT, SILU>(x, y, rows, scale, hq, n_cu, stream);
            case 2048: return launch_pow2<4, QUANT, SILU>(x, y, rows, scale, hq, n_cu, stream);
            case 4096: return launch_pow2<8, QUANT, SILU>(x, y, rows, scale, hq, n_cu, stream);
            case 8192: return launch_pow2<16, QUANT, SILU>(x, y, rows, scale, hq, n_cu, stream);
            default: break;
        }
        if (!QUANT && !SILU) {  // short vectors (e.g. head_dim 128): several rows per wave
            if (P == 256) return launch_pow2<1, false, false, 2>(x, y, rows, scale, hq, n_cu, stream);
            if (P == 128) return launch_pow2<1, false, false, 4>(x, y, rows, scale, hq, n_cu, stream);
            if (P == 64) return launch_pow2<1, false, false, 8>(x, y, rows, scale, hq, n_cu, stream);
        }
        return -1000;
    }
    if (K > 192) return -1000;
    if (P == 512) return launch_kmix<1, QUANT, SILU>(x, y, rows, K, hadK, scale, hq, n_cu, stream);
    if (P == 1024) return launch_kmix<2, QUANT, SILU>(x, y, rows, K, hadK, scale, hq, n_cu, stream);
    if (!QUANT && !SILU) {  // 11008 = 172 x 64, 5120 = 40 x 128, 7168 = 28 x 256, 18944 = 148 x 128 ...
        if (P == 256) return launch_kmix<1, false, false, 2>(x, y, rows, K, hadK, scale, hq, n_cu, stream);
        if (P == 128) return launch_kmix<1, false, false, 4>(x, y, rows, K, hadK, scale, hq, n_cu, stream);
        if (P == 64) return launch_kmix<1, false, false, 8>(x, y, rows, K, hadK, scale, hq, n_cu, stream);
    }
    return -1000;
}

}  // namespace

// Returns -1000 for shapes this file does not cover (the caller falls back to fq_hadamard.hip).
int fq_launch_hadamard_reg(const f16* x, f16* y, int64_t rows, int n, int K, const f16* hadK, float scale, int n_cu,
                           hipStream_t stream) {
    return dispatch_reg<false>(x, y, rows, n, K, hadK, scale, HadQuant{1.0f, 1.0f, nullptr, nullptr, nullptr, nullptr}, n_cu, stream);
}

// The power-of-two transform alone with an fp32 result: y32 [vecs, P] = FWHT_P(x [vecs, P]) * scale, P in {64 .. 8192}.
int fq_launch_fwht_f32(const f16* x, float* y32, int64_t vecs, int P, float scale, int n_cu, hipStream_t stream) {
    if (y32 == nullptr) return -1000;
    return dispatch_reg<false>(x, nullptr, vecs, P, 1, nullptr, scale, HadQuant{1.0f, 1.0f, nullptr, nullptr, nullptr, y32}, n_cu, stream);
}

// Hadamard + deploy.nn.Quantizer in one launch (no fp16 round trip through HBM). -1000: shape not covered.
int fq_launch_hadamard_quant(const f16* x, int64_t rows, int n, int K, const f16* hadK, float scale, float sig_max,
                             float sig_min, uint8_t* q, f16* scale_out, int n_cu, hipStream_t stream) {
    return dispatch_reg<true>(x, nullptr, rows, n, K, hadK, scale, HadQuant{sig_max, sig_min, q, scale_out, nullptr, nullptr}, n_cu, stream);
}

// x_up * silu(x_gate) formed in registers in front of the same Hadamard + Quantizer launch.
int fq_launch_silu_hadamard_quant(const f16* gate, const f16* up, int64_t rows, int n, int K, const f16* hadK, float scale,
                                  float sig_max, float sig_min, uint8_t* q, f16* scale_out, int n_cu, hipStream_t stream) {
    return dispatch_reg<true, true>(gate, nullptr, rows, n, K, hadK, scale, HadQuant{sig_max, sig_min, q, scale_out, up, nullptr},
                                    n_cu, stream);
}
