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
typedef int i32x8 __attribute__((ext_vector_type(8)));

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
        // inline asm with the accumulator TIED (D = C, one register quad per chain): through the builtin the allocator rotated the eight
        // accumulators through overlapping quads — a[4:7] = mfma(.., a[2:5]) — and the overlap serialises the chains (the first build
        // of this probe measured the 16x16x32 shape at half its rate for that reason)
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (V == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c[k]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c[k]) : "v"(ab), "v"(bb));
            }
        }
        for (int k = 0; k < 8; ++k) acc_out += c[k][0] + c[k][3];
    }
    if (acc_out == 12345.678f) out[t] = acc_out;   // (keeps the chains alive)
}

// v_mfma_scale_f32_32x32x64_f8f6f4 with BF6 (E3M2) operands and unit block scales — the instruction of the Linear4bit GEMM
// (flatquant_amd/csrc/fq_gemm_bf6.hip): 131072 FLOP per instruction. MODE 0: operands = random 6-bit codes; 1: codes of random
// integers in [-8, 7] (what the GEMM feeds it: INT4 digits are exact in BF6); 2: all zero.
template <int MODE>
__global__ __launch_bounds__(256) void mfma6_loop(float* out, int iters) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    const unsigned char tab[16] = {0, 12, 16, 18, 20, 21, 22, 23, 56, 55, 54, 53, 52, 50, 48, 44};   // BF6 codes of 0..7, -8..-1
    unsigned aw[6] = {0, 0, 0, 0, 0, 0}, bw[6] = {0, 0, 0, 0, 0, 0};
    for (int e = 0; e < 32; ++e) {
        unsigned ra = (unsigned)(rnd(t * 64 + e) * 262144.0f) , rb = (unsigned)(rnd(t * 64 + 32 + e) * 262144.0f);
        unsigned ca = MODE == 0 ? (ra & 63u) : MODE == 1 ? tab[ra & 15u] : 0u, cb = MODE == 0 ? (rb & 63u) : MODE == 1 ? tab[rb & 15u] : 0u;
        const int bit = e * 6, w = bit >> 5, sh = bit & 31;
        aw[w] |= ca << sh;
        bw[w] |= cb << sh;
        if (sh > 26) {
            aw[w + 1] |= ca >> (32 - sh);
            bw[w + 1] |= cb >> (32 - sh);
        }
    }
    const i32x8 a = {(int)aw[0], (int)aw[1], (int)aw[2], (int)aw[3], (int)aw[4], (int)aw[5], 0, 0};
    const i32x8 b = {(int)bw[0], (int)bw[1], (int)bw[2], (int)bw[3], (int)bw[4], (int)bw[5], 0, 0};
    f32x16 c[4];
    for (int k = 0; k < 4; ++k) c[k] = f32x16{0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[k], 3, 3, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    float acc_out = 0.0f;
    for (int k = 0; k < 4; ++k) acc_out += c[k][0] + c[k][15];
    if (acc_out == 12345.678f) out[t] = acc_out;
}

template <int MODE>
void run6(const char* name, float* out, double seconds, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 10000;
    const double flop_per_launch = (double)blocks * 4 * iters * 4 * 131072.0;
    hipLaunchKernelGGL((mfma6_loop<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    double el = 0;
    do {
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((mfma6_loop<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipDeviceSynchronize();
        n += 4;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < seconds);
    printf("%-44s %2d waves/SIMD  %6ld launches in %5.2f s  ->  %6.3f Pop/s\n", name, waves_per_simd, n, el, flop_per_launch * n / el / 1e15);
    fflush(stdout);
}

template <int V, bool ZERO>
void run(const char* name, float* out, double seconds, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 20000;   // 256 CUs x (waves_per_simd) workgroups of 4 waves
    const double flop_per_launch = (double)blocks * 4 * iters * 4 * 32768.0;   // (8 x 16384 = 4 x 32768 per trip)
    hipLaunchKernelGGL((mfma_loop<V, ZERO>), dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    double el = 0;
    do {
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((mfma_loop<V, ZERO>), dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipDeviceSynchronize();
        n += 4;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < seconds);
    printf("%-44s %2d waves/SIMD  %6ld launches in %5.2f s  ->  %6.3f PFLOP/s\n", name, waves_per_simd, n, el, flop_per_launch * n / el / 1e15);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double s = argc > 1 ? atof(argv[1]) : 3.0;
    float* out;
    (void)hipMalloc(&out, 256 * 16 * 256 * sizeof(float));
    if (argc > 2) {   // tools/mfma_energy.bin <seconds> fp6 : the Linear4bit GEMM's instruction only
        run6<0>("v_mfma_scale_f32_32x32x64 BF6 (random codes)", out, s, 2);
        run6<1>("v_mfma_scale_f32_32x32x64 BF6 (INT4 digits)", out, s, 2);
        run6<1>("v_mfma_scale_f32_32x32x64 BF6 (INT4 digits)", out, s, 4);
        run6<2>("v_mfma_scale_f32_32x32x64 BF6 (all zero)", out, s, 2);
        return 0;
    }
    for (int w : {2, 4}) {
        run<0, false>("v_mfma_f32_32x32x16_f16  (random operands)", out, s, w);
        run<1, false>("v_mfma_f32_16x16x32_f16  (random operands)", out, s, w);
    }
    run<2, false>("v_mfma_f32_32x32x16_bf16 (random operands)", out, s, 4);
    run<3, false>("v_mfma_f32_16x16x32_bf16 (random operands)", out, s, 4);
    run<0, true>("v_mfma_f32_32x32x16_f16  (all-zero operands)", out, s, 4);
    return 0;
}
