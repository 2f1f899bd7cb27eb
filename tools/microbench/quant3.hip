// Round 4: a third formulation of the packed-INT4 quantiser, checked and timed before it goes into the
// kernels. "fraction in the low half":
//   u = fma(y, inv, C),  C = 200.5 + 2^-16 (bits 0x43488001)   — one rounding, ulp(u) = 2^-16 for u in [128, 256)
//   bits(u) = 0x4340_0000 + floor((y inv + 8.5 + 2^-16) 2^16):  high half = 0x4340 + (rint(y inv) + 8), low half = the fraction
//   of y inv + 8.5 (+ 2^-16) in units of 2^-16.  A low half >= 2 proves rint(fl(y / s)) == high - 0x4348 (see fq_common.hpp);
//   digits gather by v_mad_u32_u16 (op_sel picks the high half), the test is a v_min3_u16 chain over the low halves.
// 23 VALU per 8 elements (19 with v_pk_fma_f32) against 33 of fq_quant8_two.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/microbench/quant3 tools/microbench/quant3.hip && gpurun -- tools/microbench/quant3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "../../flatquant_amd/csrc/fq_common.hpp"

// ---- candidates -------------------------------------------------------------------------------------------------
// (the product versions live in fq_common.hpp as fq_quant8_lo*: this file is the experiment they came from)
template <bool CLAMP, bool PK>
__device__ __forceinline__ uint32_t q3(f32x2 y01, f32x2 y23, f32x2 y45, f32x2 y67, float inv, unsigned long long& amb) {
    return fq_quant8_lo<CLAMP, PK>(y01, y23, y45, y67, inv, amb);
}

__device__ __forceinline__ uint32_t ref8(const float* y, float s) {
    return fq_pack8(fq_qexact(y[0], s), fq_qexact(y[1], s), fq_qexact(y[2], s), fq_qexact(y[3], s), fq_qexact(y[4], s),
                    fq_qexact(y[5], s), fq_qexact(y[6], s), fq_qexact(y[7], s));
}

// ---- correctness: every thread quantises 8 values; mismatches outside flagged dwords must be 0 --------------------
template <bool CLAMP, bool PK>
__global__ void check(const float* y, const float* scale, int n8, unsigned long long* stat) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = y[(size_t)i * 8 + j];
    const float s = scale[i];
    const float inv = fq_fast_inv(s);
    unsigned long long amb = 0;
    const uint32_t got = q3<CLAMP, PK>(f32x2{v[0], v[1]}, f32x2{v[2], v[3]}, f32x2{v[4], v[5]}, f32x2{v[6], v[7]}, inv, amb);
    const uint32_t want = ref8(v, s);
    const bool flagged = (amb >> (threadIdx.x & 63)) & 1ull;
    if (flagged) atomicAdd(&stat[0], 1ull);
    else if (got != want) {
        atomicAdd(&stat[1], 1ull);
        if (atomicAdd(&stat[2], 1ull) < 4) printf("MISMATCH i=%d s=%g got %08x want %08x y0=%g\n", i, s, got, want, v[0]);
    }
}

// ---- rate: MFMA phase + quantiser phase per iteration, like tools/microbench/phase_overlap.hip -----------------------
template <int KIND, bool MFMA, bool QUANT>   // KIND 0 fq_quant8_two<false>, 1 lo<false,false>, 2 lo<false,true>, 3 two<true>, 4 lo<true,false>, 5 lo<true,true>
__global__ __launch_bounds__(1024) void rate(unsigned long long* out, float seed, int iters) {
    extern __shared__ unsigned char lds_[];
    f32x16 Y[2];   // (two tiles quantised twice per iteration: 64 elements per lane inside the 128-register budget of 4 waves per SIMD)
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) Y[c][r] = seed * (float)(r - 7) + 0.01f * (float)(threadIdx.x & 63) + (float)c;
    f16x8 fa, fb;
    for (int j = 0; j < 8; ++j) { fa[j] = (f16)(seed + j); fb[j] = (f16)(seed - j); }
    f32x16 acc[4] = {};
    const int wave = threadIdx.x >> 6;
    float inv = 1.0f / (seed * 3.0f);
    const float ilo = fq_inv_lo(inv), ihi = fq_inv_hi(inv);
    uint32_t sink = 0;
    __syncthreads();
    for (int g = wave >> 2; g > 0; --g) __builtin_amdgcn_s_sleep(12);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MFMA) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[c], 0, 0, 0);
        }
        if (QUANT) {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int c = c4 & 1;
                asm volatile("" : "+v"(Y[c]));
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int o = hh * 8;
                    unsigned long long m = 0;
                    uint32_t d;
                    if (KIND == 0) d = fq_quant8_two<false>(Y[c][o], Y[c][o + 1], Y[c][o + 2], Y[c][o + 3], Y[c][o + 4], Y[c][o + 5], Y[c][o + 6], Y[c][o + 7], ilo, ihi, m);
                    else if (KIND == 3) d = fq_quant8_two<true>(Y[c][o], Y[c][o + 1], Y[c][o + 2], Y[c][o + 3], Y[c][o + 4], Y[c][o + 5], Y[c][o + 6], Y[c][o + 7], ilo, ihi, m);
                    else {
                        const f32x2 a = {Y[c][o], Y[c][o + 1]}, b = {Y[c][o + 2], Y[c][o + 3]}, e = {Y[c][o + 4], Y[c][o + 5]}, f = {Y[c][o + 6], Y[c][o + 7]};
                        if (KIND == 1) d = q3<false, false>(a, b, e, f, inv, m);
                        else if (KIND == 2) d = q3<false, true>(a, b, e, f, inv, m);
                        else if (KIND == 4) d = q3<true, false>(a, b, e, f, inv, m);
                        else d = q3<true, true>(a, b, e, f, inv, m);
                    }
                    sink ^= d;
                    if (m) sink += 1;   // (the rare path's branch, as in the kernels)
                }
            }
        }
    }
    asm volatile("" : "+v"(sink), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));   // (the clock is read AFTER the work)
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)sink;
    for (int c = 0; c < 4; ++c) s += acc[c][0];
    if (s == 12345.678f) out[1] = 1;
    if ((threadIdx.x & 63) == 0) atomicMax(&out[0], t1 - t0);
}

template <int KIND, bool MFMA, bool QUANT>
static double run_rate(unsigned long long* d, int threads) {
    const int iters = 64;
    auto kern = rate<KIND, MFMA, QUANT>;
    // 512 threads: 120 KB of dynamic LDS keeps a second workgroup off the CU -> exactly 2 waves per SIMD
    const size_t lds = threads == 512 ? 120 * 1024 : 0;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    unsigned long long h = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(d, 0, 16);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), lds, 0, d, 1.0f, iters);
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    }
    return (double)h / iters;
}

template <int KIND>
static void report(const char* name, unsigned long long* d) {
    for (int threads : {512, 1024}) {
        const double mf = run_rate<KIND, true, false>(d, threads), q = run_rate<KIND, false, true>(d, threads), both = run_rate<KIND, true, true>(d, threads);
        printf("%-34s %2d waves/SIMD: mfma-only %7.1f  quant-only %7.1f  both %7.1f ticks/iter (64 elements/lane + 32 MFMA per wave)\n", name,
               threads / 256, mf, q, both);
    }
}

__global__ void cvt_probe(float* out) {
    const float vals[12] = {0.5f, 1.5f, 2.5f, 3.5f, 254.5f, 255.5f, -0.5f, 300.0f, 0.49f, 0.51f, 1.49f, 2.51f};
    if (threadIdx.x < 12) {
        uint32_t d = 0;
        const float v = vals[threadIdx.x];
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(d) : "v"(v));
        out[threadIdx.x] = (float)d;
    }
}

int main() {
    // ---- rounding of v_cvt_pk_u8_f32 (for the record)
    {
        float* d;
        hipMalloc(&d, 64);
        cvt_probe<<<1, 64>>>(d);
        float h[12];
        hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
        printf("v_cvt_pk_u8_f32 of 0.5 1.5 2.5 3.5 254.5 255.5 -0.5 300 0.49 0.51 1.49 2.51:");
        for (int i = 0; i < 12; ++i) printf(" %g", h[i]);
        printf("\n");
    }
    // ---- correctness
    const int n8 = 1 << 22;
    std::vector<float> y((size_t)n8 * 8), s(n8);
    srand(1234);
    auto urand = []() { return (float)rand() / (float)RAND_MAX; };
    for (int i = 0; i < n8; ++i) {
        const int mode = i & 7;
        float sc = std::ldexp(0.5f + urand(), (rand() % 24) - 16);   // scales over 24 binades
        if (mode == 7) sc = std::ldexp(1.0f, (rand() % 24) - 16);
        s[i] = sc;
        for (int j = 0; j < 8; ++j) {
            float t;
            if (mode < 3) t = (urand() * 2.0f - 1.0f) * 7.4f;                                   // plain
            else if (mode < 6) {                                                                  // near half-integers
                const int k = (rand() % 16) - 8;
                const float eps = std::ldexp((urand() * 2.0f - 1.0f), -(10 + rand() % 16));
                t = ((float)k + 0.5f) * (1.0f + eps);
            } else t = (urand() * 2.0f - 1.0f) * 12.0f;                                          // beyond the clamp (CLAMP variants only)
            y[(size_t)i * 8 + j] = t * sc;
        }
    }
    float *dy, *ds;
    unsigned long long* dst;
    hipMalloc(&dy, y.size() * 4);
    hipMalloc(&ds, s.size() * 4);
    hipMalloc(&dst, 64);
    hipMemcpy(dy, y.data(), y.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ds, s.data(), s.size() * 4, hipMemcpyHostToDevice);
    auto run_check = [&](const char* name, auto kern) {
        hipMemset(dst, 0, 64);
        hipLaunchKernelGGL(kern, dim3(n8 / 256), dim3(256), 0, 0, dy, ds, n8, dst);
        unsigned long long h[3];
        hipMemcpy(h, dst, 24, hipMemcpyDeviceToHost);
        printf("check %-22s flagged dwords %llu of %d (%.2e), unflagged mismatches %llu\n", name, h[0], n8, (double)h[0] / n8, h[1]);
    };
    run_check("lo<clamp,single>", check<true, false>);
    run_check("lo<clamp,pk>", check<true, true>);   // (round 4: NOT bit-exact as written — the pk form is kept for the rate comparison only)
    // the no-clamp forms are only defined for quotients inside [-8.49, 7.49]: restrict the data to modes 0..5 by zeroing mode 6/7 rows
    for (int i = 0; i < n8; ++i)
        if ((i & 7) >= 6)
            for (int j = 0; j < 8; ++j) y[(size_t)i * 8 + j] = (urand() * 2.0f - 1.0f) * 7.4f * s[i];
    // and keep the near-half-integer rows inside the range (k = -8 -> -8.5 (1 + eps) may fall below -8.49)
    for (int i = 0; i < n8; ++i)
        for (int j = 0; j < 8; ++j) {
            float& v = y[(size_t)i * 8 + j];
            const float t = v / s[i];
            if (t > 7.45f || t < -8.45f) v = 0.25f * s[i];
        }
    hipMemcpy(dy, y.data(), y.size() * 4, hipMemcpyHostToDevice);
    run_check("lo<noclamp,single>", check<false, false>);
    run_check("lo<noclamp,pk>", check<false, true>);

    // ---- rate
    unsigned long long* d;
    hipMalloc(&d, 16);
    report<0>("fq_quant8_two<noclamp> (round 3)", d);
    report<1>("fq_quant8_lo<noclamp,single>", d);
    report<2>("fq_quant8_lo<noclamp,pk_fma>", d);
    report<3>("fq_quant8_two<clamp> (round 3)", d);
    report<4>("fq_quant8_lo<clamp,single>", d);
    report<5>("fq_quant8_lo<clamp,pk_fma>", d);
    return 0;
}
