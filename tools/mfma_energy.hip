// Sustained fp16 / bf16 matrix-pipe rate UNDER THE PACKAGE POWER CAP, by instruction shape (round 5, the energy census' follow-up):
// the dense Kronecker kernels run at 1400 W and 1.7 GHz, and their "no MFMA" ablation puts 0.8-0.95 pJ per FLOP on the matrix pipe +
// operand delivery (profiles/r05_energy_census.txt) — what does the pipe ALONE sustain at the cap, and does the instruction shape matter?
// v_mfma_f32_32x32x16_f16 moves 16 accumulator registers in and out per 32768 FLOP, v_mfma_f32_16x16x32_f16 4 per 16384 FLOP (half
// the accumulator traffic per FLOP, the same operand traffic). Registers only, no memory: every wave loops over independent chains.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_energy.bin tools/mfma_energy.hip && tools/mfma_energy.bin [seconds]
// Prints, per variant: launches, seconds, PFLOP/s; sample rocm-smi next to it (tools/gpu_call.sh r05c4 does) for W and MHz.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rnd(unsigned s) {   // a fixed pseudo-random value in (-1/8, 1/8): realistic mantissa toggling
    s = s * 747796405u + 2891336453u;
    s = ((s >> ((s >> 28) + 4)) ^ s) * 277803737u;
    s = (s >> 22) ^ s;
    return ((float)(s & 0xffff) / 65536.0f - 0.5f) * 0.25f;
}

template <int V, bool ZERO>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    f16x8 a, b;
    bf16x8 ab, bb;
    for (int j = 0; j < 8; ++j) {
        const float x = ZERO ? 0.0f : rnd(t * 16 + j), y = ZERO ? 0.0f : rnd(t * 16 + 8 + j);
        a[j] = (_Float16)x, b[j] = (_Float16)y, ab[j] = (__bf16)x, bb[j] = (__bf16)y;
    }
    float acc_out = 0.0f;
    if (V == 0 || V == 2) {            // 32x32x16: four independent chains of 16 registers
        f32x16 c[4];
        for (int k = 0; k < 4; ++k) c[k] = f32x16{0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                c[k] = V == 0 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[k], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c[k], 0, 0, 0);
        }
        for (int k = 0; k < 4; ++k) acc_out += c[k][0] + c[k][15];
    } else {                           // 16x16x32: eight independent chains of 4 registers (the same FLOPs per loop trip)
        f32x4 c[8];
        for (int k = 0; k < 8; ++k) c[k] = f32x4{0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                c[k] = V == 1 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[k], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c[k], 0, 0, 0);
        }
        for (int k = 0; k < 8; ++k) acc_out += c[k][0] + c[k][3];
    }
    if (acc_out == 12345.678f) out[t] = acc_out;   // (keeps the chains alive)
}

template <int V, bool ZERO>
void run(const char* name, float* out, double seconds, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 20000;   // 256 CUs x (waves_per_simd) workgroups of 4 waves
    const double flop_per_launch = (double)blocks * 4 * iters * 4 * 32768.0;   // (8 x 16384 = 4 x 32768 per trip)
    hipLaunchKernelGGL((mfma_loop<V, ZERO>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    double el = 0;
    do {
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((mfma_loop<V, ZERO>), dim3(blocks), dim3(256), 0, 0, out, iters);
        hipDeviceSynchronize();
        n += 4;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < seconds);
    printf("%-44s %2d waves/SIMD  %6ld launches in %5.2f s  ->  %6.3f PFLOP/s\n", name, waves_per_simd, n, el, flop_per_launch * n / el / 1e15);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double s = argc > 1 ? atof(argv[1]) : 3.0;
    float* out;
    hipMalloc(&out, 256 * 16 * 256 * sizeof(float));
    for (int w : {2, 4}) {
        run<0, false>("v_mfma_f32_32x32x16_f16  (random operands)", out, s, w);
        run<1, false>("v_mfma_f32_16x16x32_f16  (random operands)", out, s, w);
    }
    run<2, false>("v_mfma_f32_32x32x16_bf16 (random operands)", out, s, 4);
    run<3, false>("v_mfma_f32_16x16x32_bf16 (random operands)", out, s, 4);
    run<0, true>("v_mfma_f32_32x32x16_f16  (all-zero operands)", out, s, 4);
    return 0;
}
