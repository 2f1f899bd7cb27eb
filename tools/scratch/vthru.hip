// Issue cost of the VALU / MFMA instructions the quantiser uses, on gfx950: cycles per instruction per SIMD with
// 1, 2 and 4 waves resident on the SIMD (s_memtime around 100 x 64 independent instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float seed) {
    float a[8], b[8];
    f32x2 p[8], q[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; b[i] = seed * i; p[i] = f32x2{a[i], b[i]}; q[i] = f32x2{b[i], a[i]}; }
    f16x8 fa, fb;
    for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)(seed + j); fb[j] = (_Float16)(seed - j); }
    f32x16 acc[4] = {};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 100; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                REP8(X)
#undef X
            } else if (OP == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if (OP == 2) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if (OP == 3) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if (OP == 4) {
#define X(i) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (OP == 5) {
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (OP == 6) {
#define X(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]));
                REP8(X)
#undef X
            } else if (OP == 7) {
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                REP8(X)
#undef X
            } else if (OP == 8) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (OP == 9) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                REP8(X)
#undef X
            } else if (OP == 10) {  // 8 MFMAs on 4 accumulators
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[c], 0, 0, 0);
            } else if (OP == 11) {  // 8 MFMAs interleaved with 8 pk_fma
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[c], 0, 0, 0);
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[r * 4 + c]) : "v"(q[r * 4 + c]));
                    }
            } else if (OP >= 13 && OP <= 17) {  // 8 MFMAs, each followed by fillers of one kind
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[c], 0, 0, 0);
#pragma unroll
                        for (int z = 0; z < (OP == 13 ? 8 : 4); ++z) {
                            const int i = (r * 4 + c + z) & 7;
                            if (OP == 13) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[z & 7]));
                            if (OP == 14) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[z & 7]));
                            if (OP == 15) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(a[i]) : "v"(b[z]), "v"(b[z + 1]));
                            if (OP == 16) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[z]), "v"(b[z + 1]));
                            if (OP == 17) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[z]));
                        }
                    }
            } else if (OP == 12) {  // 8 MFMAs, each followed by 4 pk_fma
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[c], 0, 0, 0);
#pragma unroll
                        for (int z = 0; z < 4; ++z)
                            asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[(r * 4 + c + z) & 7]) : "v"(q[z]));
                    }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    for (int c = 0; c < 4; ++c) s += acc[c][0];
    if (s == 12345.678f) out[1] = 1;
    if ((threadIdx.x & 63) == 0) atomicMax(&out[0], t1 - t0);
}
template <int OP>
static void run(const char* name, unsigned long long* d) {
    printf("%-34s", name);
    for (int threads : {256, 512, 1024}) {
        hipMemset(d, 0, 16);
        k<OP><<<256, threads>>>(d, 1.0f);
        hipMemset(d, 0, 16);
        k<OP><<<256, threads>>>(d, 1.0f);
        unsigned long long h = 0;
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        const double per = (double)h / (100.0 * 64.0);      // slowest wave, ticks per instruction (of that wave)
        printf("  %d w/SIMD: %6.2f ticks/inst/wave = %5.2f per SIMD", threads / 256, per, per / (threads / 256));
    }
    printf("\n");
}
int main() {
    unsigned long long* d;
    hipMalloc(&d, 16);
    run<0>("v_fma_f32", d);
    run<9>("v_mul_f32", d);
    run<1>("v_pk_fma_f32", d);
    run<2>("v_pk_mul_f32", d);
    run<3>("v_pk_add_f32", d);
    run<4>("v_max3_f32 (abs mods)", d);
    run<5>("v_med3_f32", d);
    run<6>("v_rndne_f32", d);
    run<7>("v_cvt_pk_f16_f32", d);
    run<8>("v_perm_b32", d);
    run<10>("mfma_32x32x16_f16 (4 acc)", d);
    run<11>("mfma + 1 pk_fma each (per pair)", d);
    run<12>("mfma + 4 pk_fma each (per 5)", d);
    run<13>("mfma + 8 v_fma_f32 each", d);
    run<14>("mfma + 4 v_fma_f32 each", d);
    run<15>("mfma + 4 v_max3_f32 each", d);
    run<16>("mfma + 4 v_med3_f32 each", d);
    run<17>("mfma + 4 v_cvt_pk_f16_f32 each", d);
    return 0;
}
