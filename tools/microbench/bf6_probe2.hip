#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const int* a, const int* b, float* d, int fmt) {
    const int lane = threadIdx.x;
    i32x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = a[lane * 8 + i]; bv[i] = b[lane * 8 + i]; }
    f32x16 acc = {0};
    // cbsz = A format, blgp = B format: 3 = bf6 (e3m2); scales: 127 = 2^0
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 3, 3, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int i = 0; i < 16; ++i) d[lane * 16 + i] = acc[i];
}
static unsigned bf6(int v) {
    static const unsigned c[9] = {0x00, 0x0C, 0x10, 0x12, 0x14, 0x15, 0x16, 0x17, 0x18};
    return v < 0 ? (0x20 | c[-v]) : c[v];
}
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void thr(float* out, int iters) {
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 2654435761u + i; b[i] = threadIdx.x * 40503u + 7 * i; }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    typedef int i32x16 __attribute__((ext_vector_type(16)));
    i32x16 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // bf6 x bf6, K = 64
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 3, 3, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c1, 3, 3, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, a, c2, 3, 3, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, b, c3, 3, 3, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else if (MODE == 1) {  // fp8 x fp8, K = 64
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c1, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, a, c2, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, b, c3, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else if (MODE >= 3 && MODE <= 8) {  // 16x16x128 / other formats: 3 bf6 16x16x128, 4 fp4 32x32x64, 5 fp4 16x16x128, 6 fp6 (e2m3) 32x32x64, 7 fp8 16x16x128, 8 fp6 16x16x128
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            static_assert(sizeof(f32x4) == 16, "");
            f32x4 &e0 = *reinterpret_cast<f32x4*>(&c0), &e1 = *reinterpret_cast<f32x4*>(&c1), &e2 = *reinterpret_cast<f32x4*>(&c2), &e3 = *reinterpret_cast<f32x4*>(&c3);
#define M16(F) e0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, e0, F, F, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); \
               e1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, a, e1, F, F, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); \
               e2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, a, e2, F, F, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); \
               e3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, b, e3, F, F, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#define M32(F) c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, F, F, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); \
               c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c1, F, F, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); \
               c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, a, c2, F, F, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); \
               c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, b, c3, F, F, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            if (MODE == 3) { M16(3) } else if (MODE == 4) { M32(4) } else if (MODE == 5) { M16(4) } else if (MODE == 6) { M32(2) } else if (MODE == 7) { M16(0) } else { M16(2) }
        } else {  // i8, K = 32
            i32x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
            d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b4, a4, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, a4, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b4, b4, d3, 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i] + (float)(d0[i] + d1[i] + d2[i] + d3[i]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
static void run_thr(const char* name, double ops_per_mfma) {
    float* o; hipMalloc(&o, 1024 * 256 * 4);
    const int iters = 20000, blocks = 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    thr<MODE><<<blocks, 256>>>(o, 2000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    thr<MODE><<<blocks, 256>>>(o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double total = (double)blocks * 4 * iters * 4 * ops_per_mfma;
    printf("%s: %.2f ms, %.2f Pop/s\n", name, ms, total / ms / 1e12);
}
int main() {
    run_thr<0>("bf6 32x32x64 (scale)", 2.0 * 32 * 32 * 64);
    run_thr<1>("fp8 32x32x64 (scale)", 2.0 * 32 * 32 * 64);
    run_thr<2>("i8  32x32x32", 2.0 * 32 * 32 * 32);
    run_thr<3>("bf6 16x16x128 (scale)", 2.0 * 16 * 16 * 128);
    run_thr<8>("fp6 16x16x128 (scale)", 2.0 * 16 * 16 * 128);
    run_thr<6>("fp6 32x32x64 (scale)", 2.0 * 32 * 32 * 64);
    run_thr<4>("fp4 32x32x64 (scale)", 2.0 * 32 * 32 * 64);
    run_thr<5>("fp4 16x16x128 (scale)", 2.0 * 16 * 16 * 128);
    run_thr<7>("fp8 16x16x128 (scale)", 2.0 * 16 * 16 * 128);

    int A[32][64], B[64][32];
    for (int i = 0; i < 32; ++i) for (int k2 = 0; k2 < 64; ++k2) A[i][k2] = ((i + 3 * k2) % 16) - 8;
    for (int k2 = 0; k2 < 64; ++k2) for (int j = 0; j < 32; ++j) B[k2][j] = ((5 * k2 + 7 * j + 3) % 16) - 8;
    std::vector<int> ha(64 * 8, 0), hb(64 * 8, 0);
    for (int l = 0; l < 64; ++l) {
        // hypothesis: lane l holds row/col l%32, k = 32*(l/32) + e, e = 0..31, 6-bit fields packed little-endian
        for (int e = 0; e < 32; ++e) {
            unsigned ca = bf6(A[l % 32][32 * (l / 32) + e]), cb = bf6(B[32 * (l / 32) + e][l % 32]);
            int bit = e * 6;
            for (int t = 0; t < 6; ++t) {
                if (ca >> t & 1) ha[l * 8 + (bit + t) / 32] |= 1u << ((bit + t) % 32);
                if (cb >> t & 1) hb[l * 8 + (bit + t) / 32] |= 1u << ((bit + t) % 32);
            }
        }
    }
    int *da, *db; float* dd;
    hipMalloc(&da, 64 * 8 * 4); hipMalloc(&db, 64 * 8 * 4); hipMalloc(&dd, 64 * 16 * 4);
    hipMemcpy(da, ha.data(), 64 * 8 * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 64 * 8 * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dd, 3);
    std::vector<float> hd(64 * 16);
    hipMemcpy(hd.data(), dd, 64 * 16 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        long ref = 0; for (int k2 = 0; k2 < 64; ++k2) ref += A[row][k2] * B[k2][col];
        if ((long)hd[l * 16 + r] != ref) { if (bad < 8) printf("lane %d r %d got %f want %ld\n", l, r, hd[l * 16 + r], ref); ++bad; }
    }
    printf("mismatches %d / 1024\n", bad);
    return 0;
}
