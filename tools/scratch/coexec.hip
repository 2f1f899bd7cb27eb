// Does VALU work of one wave overlap MFMA work of another wave on the same SIMD? (gfx950 microbenchmark)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// role: 0 = MFMA loop, 1 = VALU loop, 2 = idle; chosen per block: mode bit mask per parity
__global__ __launch_bounds__(256) void k(float* out, int iters, int role_even, int role_odd) {
    const int role = (blockIdx.x & 1) ? role_odd : role_even;
    const int lane = threadIdx.x;
    if (role == 0) {
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(lane * 0.01f + j); b[j] = (_Float16)(j * 0.5f - lane * 0.02f); }
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
            }
        }
        out[blockIdx.x * 256 + lane] = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (role == 1) {
        float v[8];
        for (int j = 0; j < 8; ++j) v[j] = lane * 0.001f + j;
        const float m = 1.0001f, ad = 0.5f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], m, ad);   // 128 VALU per iteration, 8 chains
            }
        }
        float s = 0;
        for (int j = 0; j < 8; ++j) s += v[j];
        out[blockIdx.x * 256 + lane] = s;
    }
}
static float run(float* d, int blocks, int iters, int re, int ro) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, 256>>>(d, iters, re, ro); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<blocks, 256>>>(d, iters, re, ro);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}
int main() {
    float* d; hipMalloc(&d, 2048 * 256 * 4);
    const int it = 2000;
    // 512 blocks of 4 waves: 2 blocks per CU -> 2 waves per SIMD (one of each parity if the dispatcher alternates)
    printf("MFMA both parities        : %8.1f us (16 MFMA/iter/wave, 2 waves/SIMD)\n", run(d, 512, it, 0, 0));
    printf("VALU both parities        : %8.1f us (128 VALU/iter/wave, 2 waves/SIMD)\n", run(d, 512, it, 1, 1));
    printf("MFMA even, idle odd       : %8.1f us\n", run(d, 512, it, 0, 2));
    printf("VALU even, idle odd       : %8.1f us\n", run(d, 512, it, 1, 2));
    printf("MFMA even + VALU odd      : %8.1f us  (perfect overlap = max of the two lines above)\n", run(d, 512, it, 0, 1));
    printf("1024 blocks MFMA/VALU mix : %8.1f us (4 waves/SIMD)\n", run(d, 1024, it, 0, 1));
    printf("1024 blocks all MFMA      : %8.1f us\n", run(d, 1024, it, 0, 0));
    printf("1024 blocks all VALU      : %8.1f us\n", run(d, 1024, it, 1, 1));
    return 0;
}
