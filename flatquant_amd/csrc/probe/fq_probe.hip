// fq_probe.hip — MEASUREMENT / TEST INFRASTRUCTURE, built into flatquant_amd/lib/libfqprobe.so (NOT part of libfqhip.so or of
// include/fqhip.h; declared in include/fqprobe.h): the no-arithmetic streaming kernel bench.py quotes as the practical HBM
// floor next to the 8 TB/s spec peak, and the one-instruction MFMA probe the oracle's accumulation model was calibrated with
// (tools/mfma_probe*.py). Nothing in the product library calls or links this file.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

// D[32,32] = A[32,16] . B[16,32] + C with one v_mfma_f32_32x32x16_f16. All row-major, A/B fp16, C/D fp32.
__global__ void fq_probe_mfma_kernel(const f16* __restrict__ A, const f16* __restrict__ B, const float* __restrict__ C,
                                     float* __restrict__ D) {
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    f16x8 a, b;
    f32x16 acc;
    for (int j = 0; j < 8; ++j) {
        a[j] = A[c * 16 + h * 8 + j];        // A[i = c][k = 8h + j]
        b[j] = B[(h * 8 + j) * 32 + c];      // B[k = 8h + j][col = c]
    }
    for (int r = 0; r < 16; ++r) acc[r] = C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + c];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + c] = acc[r];
}

// Streams exactly the bytes of the d = 4096 fused kernel — read 8 KB, write 2 KB + 2 B per token — with perfectly coalesced
// 16-byte accesses, 8 loads in flight per lane and no arithmetic beyond an OR-fold.
__global__ __launch_bounds__(256) void fq_probe_stream_kernel(const u32x4* __restrict__ x, int64_t rows,
                                                              u32x4* __restrict__ q, f16* __restrict__ s) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t t = wave; t < rows; t += nw) {
        const u32x4* p = x + t * 512 + lane;  // 512 x 16 B = 8 KB per token
        u32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_nontemporal_load(p + i * 64);
        u32x4 a = (v[0] | v[1]) ^ (v[2] | v[3]), b = (v[4] | v[5]) ^ (v[6] | v[7]);
        q[t * 128 + lane] = a;
        q[t * 128 + 64 + lane] = b;
        if (lane == 0) s[t] = (f16)1.0f;
    }
}

}  // namespace

extern "C" {

int fq_probe_mfma_32x32x16_f16(const void* A, const void* B, const void* C, void* D, void* stream) {
    if (!A || !B || !C || !D) return -1;
    hipLaunchKernelGGL(fq_probe_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const f16*)A, (const f16*)B,
                       (const float*)C, (float*)D);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int fq_probe_stream_4096(const void* x, int64_t rows, void* q, void* s, int waves_per_simd, void* stream) {
    if (!x || !q || !s || rows <= 0 || waves_per_simd < 1 || waves_per_simd > 8) return -1;
    int dev = 0, n_cu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (n_cu <= 0) n_cu = 256;
    int64_t blocks = (int64_t)n_cu * waves_per_simd;
    if (blocks > (rows + 3) / 4) blocks = (rows + 3) / 4;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fq_probe_stream_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x, rows,
                       (u32x4*)q, (f16*)s);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // extern "C"
