// ds_read_b64_tr_b16 + v_mfma_f32_16x16x16_f16 layout probe (gfx950): hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
// LDS tile [16 rows][16 cols] fp16, element (r, c) = r * 16 + c; every lane reads 8 bytes at 8 * lane through the transpose read.
// Expected (cdna_hip_programming.md T10): lane l = 16 G + n gets (row 4G + j, col n) for j = 0..3: the B fragment of a 16x16x16 MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out, float* out2) {
    __shared__ __attribute__((aligned(16))) f16 tile[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) tile[i] = (f16)(float)i;
    __syncthreads();
    unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)tile + 8u * lane;
    f16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (float)v[j];
    // MFMA: A[m][k] = (m == 1 ? 1 : 0) for all k  (row 1 of A is ones) -> D[1][n] = sum_k B[k][n]; B = the tile (k = row, n = col)
    f16x4 a;
    for (int j = 0; j < 4; ++j) a[j] = (lane % 16 == 1) ? (f16)1.0f : (f16)0.0f;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(a, v, c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) out2[lane * 4 + j] = c[j];
}
int main() {
    float *d, *d2, h[256], h2[256];
    hipMalloc(&d, sizeof(h)); hipMalloc(&d2, sizeof(h2));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, d2);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(h2, d2, sizeof(h2), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (h[l * 4 + j] != (float)((4 * (l / 16) + j) * 16 + (l % 16))) ++bad;
    printf("tr read: %d mismatches against (row 4G + j, col n); lane 0: %g %g %g %g, lane 17: %g %g %g %g\n", bad, h[0], h[1], h[2], h[3], h[68], h[69], h[70], h[71]);
    // D[1][n] = sum over rows r of (16 r + n) = 16 * 120 + 16 n = 1920 + 16 n; lane (G, n) holds D[4G + r][n]: expect lanes 0..15, element r = 1
    int bad2 = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const float want = (l / 16 == 0 && r == 1) ? 1920.0f + 16.0f * (l % 16) : 0.0f;
        if (h2[l * 4 + r] != want) ++bad2;
    }
    printf("mfma 16x16x16: %d mismatches against D[4G + r][n] in lane (G, n); lane 3: %g %g %g %g\n", bad2, h2[12], h2[13], h2[14], h2[15]);
    return 0;
}
