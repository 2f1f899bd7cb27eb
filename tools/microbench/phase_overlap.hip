// Do the VALU phases of some waves hide behind the MFMA phases of OTHER waves of the same SIMD? (gfx950)
// A 16-wave workgroup per CU (4 waves per SIMD, as fq_kron64_kernel), every wave loops over
//   phase A: 32 x v_mfma_f32_32x32x16_f16 (4 accumulator chains)      = one token's two GEMMs
//   phase B: NV VALU instructions of one kind                          = one token's statistics + quantiser
// with the waves of a SIMD started a quarter iteration apart. Reported: cycles per iteration per SIMD against
// the MFMA-only and VALU-only loops. KIND 0: v_fma_f32, 1: v_pk_fma_f32 (half as many), 2: v_lshl_add_u32,
// 3: mix of fma + max3 + lshl_add (the quant8_two stream).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int NV, bool MFMA>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float seed, int iters) {
    float a[8], b[8];
    f32x2 p[8], q[8];
    unsigned ia[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = seed + i + threadIdx.x; b[i] = seed * i; p[i] = f32x2{a[i], b[i]}; q[i] = f32x2{b[i], a[i]};
        ia[i] = threadIdx.x * 7 + i;
    }
    f16x8 fa, fb;
    for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)(seed + j); fb[j] = (_Float16)(seed - j); }
    f32x16 acc[4] = {};
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    for (int g = wave >> 2; g > 0; --g) __builtin_amdgcn_s_sleep(12);   // ~ a quarter of an iteration per group
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MFMA) {
            __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[c], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
#pragma unroll
        for (int u = 0; u < NV / 8; ++u) {
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
            if (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                REP8(X)
#undef X
            } else if (KIND == 1) {
                if (u & 1) continue;  // half as many packed instructions for the same element count
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if (KIND == 2) {
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]));
                REP8(X)
#undef X
            } else {
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[0]) : "v"(b[0]));
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[1]) : "v"(b[1]));
                asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(ia[0]) : "v"(ia[1]));
                asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(ia[2]) : "v"(ia[3]));
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[2]) : "v"(b[2]));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[3]) : "v"(b[3]), "v"(b[4]));
                asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[4]) : "v"(b[5]));
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[5]) : "v"(b[6]), "v"(b[7]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)ia[i];
    for (int c = 0; c < 4; ++c) s += acc[c][0];
    if (s == 12345.678f) out[1] = 1;
    if ((threadIdx.x & 63) == 0) atomicMax(&out[0], t1 - t0);
}

template <int KIND, int NV, bool MFMA>
static double run(unsigned long long* d) {
    const int iters = 64;
    hipMemset(d, 0, 16);
    k<KIND, NV, MFMA><<<256, 1024>>>(d, 1.0f, iters);
    hipMemset(d, 0, 16);
    k<KIND, NV, MFMA><<<256, 1024>>>(d, 1.0f, iters);
    unsigned long long h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    return (double)h / iters;   // s_memtime ticks (100 MHz on gfx950? printed raw) per iteration of the slowest wave
}

template <int KIND, int NV>
static void report(const char* name, unsigned long long* d, double mf) {
    const double both = run<KIND, NV, true>(d), valu = run<KIND, NV, false>(d);
    printf("%-28s NV=%4d  mfma-only %8.1f  valu-only %8.1f  both %8.1f  (sum %8.1f, max %8.1f) ticks/iter, 4 waves/SIMD\n", name,
           NV, mf, valu, both, mf + valu, mf > valu ? mf : valu);
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 16);
    const double mf = run<0, 0, true>(d);
    report<0, 256>("v_fma_f32", d, mf);
    report<0, 512>("v_fma_f32", d, mf);
    report<1, 512>("v_pk_fma_f32 (NV/2 instr)", d, mf);
    report<2, 512>("v_lshl_add_u32", d, mf);
    report<3, 256>("mix fma/lshl_add/max3/cvt", d, mf);
    report<3, 512>("mix fma/lshl_add/max3/cvt", d, mf);
    return 0;
}
