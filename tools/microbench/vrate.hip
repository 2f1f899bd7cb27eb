// Issue cost of single VALU instructions on gfx950 (MI355X): s_memtime ticks per instruction per SIMD with 1, 2 and 4 waves
// resident on the SIMD, 8 independent chains per wave (second session of round 3: which of the quantiser's candidate
// instructions are full rate — v_fma_mix_f32 and the packed fp16 ops in particular).
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/vrate tools/microbench/vrate.hip && gpurun -- tools/microbench/vrate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define OPS(F)                                                                                              \
    F(0, "v_fma_f32", "v_fma_f32 %0, %0, %1, %0")                                                           \
    F(1, "v_fma_f32 (sgpr src2)", "v_fma_f32 %0, %0, %1, %3")                                               \
    F(2, "v_fma_mix_f32 lo", "v_fma_mix_f32 %0, %0, %1, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]")                \
    F(3, "v_fma_mix_f32 hi", "v_fma_mix_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]")                \
    F(4, "v_fma_mixlo_f16", "v_fma_mixlo_f16 %0, %0, %1, %2 op_sel_hi:[0,0,0]")                               \
    F(5, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 4, %1")                                                  \
    F(6, "v_cvt_f16_f32", "v_cvt_f16_f32 %0, %0")                                                           \
    F(7, "v_cvt_f32_f16", "v_cvt_f32_f16 %0, %0")                                                           \
    F(8, "v_cvt_f32_f16 sdwa hi", "v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1") \
    F(9, "v_cvt_pk_f16_f32", "v_cvt_pk_f16_f32 %0, %0, %1")                                                 \
    F(10, "v_cvt_pkrtz_f16_f32", "v_cvt_pkrtz_f16_f32 %0, %0, %1")                                          \
    F(11, "v_pk_max_f16", "v_pk_max_f16 %0, %0, %1")                                                        \
    F(12, "v_pk_add_f16", "v_pk_add_f16 %0, %0, %1")                                                        \
    F(13, "v_pk_mul_f16", "v_pk_mul_f16 %0, %0, %1")                                                        \
    F(14, "v_pk_fma_f16", "v_pk_fma_f16 %0, %0, %1, %2")                                                    \
    F(15, "v_max3_f32", "v_max3_f32 %0, %0, %1, %2")                                                        \
    F(16, "v_max_f32", "v_max_f32 %0, %0, %1")                                                              \
    F(17, "v_med3_f32", "v_med3_f32 %0, %0, %1, %2")                                                        \
    F(18, "v_bfi_b32", "v_bfi_b32 %0, %0, %1, %2")                                                          \
    F(19, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2")                                                    \
    F(20, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2")                                                        \
    F(21, "v_add_u32", "v_add_u32 %0, %0, %1")                                                              \
    F(22, "v_xor_b32", "v_xor_b32 %0, %0, %1")                                                              \
    F(23, "v_sad_u32", "v_sad_u32 %0, %1, %2, %0")                                                          \
    F(24, "v_mul_f32", "v_mul_f32 %0, %0, %1")                                                              \
    F(25, "v_add_f32", "v_add_f32 %0, %0, %1")                                                              \
    F(26, "v_rndne_f32", "v_rndne_f32 %0, %0")                                                              \
    F(27, "v_cvt_i32_f32", "v_cvt_i32_f32 %0, %0")                                                          \
    F(28, "v_pk_fma_f32", "v_pk_fma_f32 %4, %4, %5, %4")                                                    \
    F(29, "v_pk_mul_f32", "v_pk_mul_f32 %4, %4, %5")                                                        \
    F(30, "v_pk_add_f32", "v_pk_add_f32 %4, %4, %5")                                                        \
    F(31, "v_mov_b32", "v_mov_b32 %0, %1")                                                                  \
    F(34, "v_dot2_f32_f16", "v_dot2_f32_f16 %0, %1, %2, %0")                                                \
    F(35, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2")                                                  \
    F(36, "v_cmp_ne_u32 (sgpr dst)", "v_cmp_ne_u32_e64 s[20:21], %0, %1")                                   \
    F(37, "v_mad_u32_u16 op_sel hi", "v_mad_u32_u16 %0, %1, 16, %0 op_sel:[1,0,0,0]")                       \
    F(38, "v_mad_u32_u16 sgpr", "v_mad_u32_u16 %0, %1, %3, %0 op_sel:[1,0,0,0]")                            \
    F(39, "v_min3_u16", "v_min3_u16 %0, %0, %1, %2")                                                        \
    F(40, "v_med3_i32", "v_med3_i32 %0, %0, %1, %2")                                                        \
    F(41, "v_min_u16", "v_min_u16_e32 %0, %0, %1")                                                          \
    F(42, "v_cmp_gt_u16 (sgpr dst)", "v_cmp_gt_u16_e64 s[20:21], 2, %0")                                    \
    F(43, "v_alignbit_b32", "v_alignbit_b32 %0, %1, %0, 4")                                                 \
    F(44, "v_bfe_u32", "v_bfe_u32 %0, %0, 16, 4")                                                           \
    F(45, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 4, %1")                                                   \
    F(46, "v_bitop3_b32", "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x36")                                        \
    F(47, "v_mad_u32_u24 k", "v_mad_u32_u24 %0, %1, 16, %0")

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float seed) {
    float a[8], b[8], c[8];
    f32x2 p[8], q[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = seed + i + threadIdx.x;
        b[i] = seed * i + 0.5f;
        c[i] = seed - i;
        p[i] = f32x2{a[i], b[i]};
        q[i] = f32x2{b[i], c[i]};
    }
    const float sg = seed * 3.0f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 100; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#define F(N, NAME, TXT) if (OP == N) {                                                                                     \
    asm volatile(TXT : "+v"(a[0]) : "v"(b[0]), "v"(c[0]), "s"(sg), "v"(p[0]), "v"(q[0]) : "s20", "s21");             \
    asm volatile(TXT : "+v"(a[1]) : "v"(b[1]), "v"(c[1]), "s"(sg), "v"(p[1]), "v"(q[1]) : "s20", "s21");             \
    asm volatile(TXT : "+v"(a[2]) : "v"(b[2]), "v"(c[2]), "s"(sg), "v"(p[2]), "v"(q[2]) : "s20", "s21");             \
    asm volatile(TXT : "+v"(a[3]) : "v"(b[3]), "v"(c[3]), "s"(sg), "v"(p[3]), "v"(q[3]) : "s20", "s21");             \
    asm volatile(TXT : "+v"(a[4]) : "v"(b[4]), "v"(c[4]), "s"(sg), "v"(p[4]), "v"(q[4]) : "s20", "s21");             \
    asm volatile(TXT : "+v"(a[5]) : "v"(b[5]), "v"(c[5]), "s"(sg), "v"(p[5]), "v"(q[5]) : "s20", "s21");             \
    asm volatile(TXT : "+v"(a[6]) : "v"(b[6]), "v"(c[6]), "s"(sg), "v"(p[6]), "v"(q[6]) : "s20", "s21");             \
    asm volatile(TXT : "+v"(a[7]) : "v"(b[7]), "v"(c[7]), "s"(sg), "v"(p[7]), "v"(q[7]) : "s20", "s21");             \
}
            OPS(F)
#undef F
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[1] = 1;
    if ((threadIdx.x & 63) == 0) atomicMax(&out[0], t1 - t0);
}
template <int OP>
static void run(const char* name, unsigned long long* d) {
    printf("%-28s", name);
    for (int threads : {256, 512, 1024}) {
        unsigned long long h = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(d, 0, 16);
            k<OP><<<256, threads>>>(d, 1.0f);
            hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        }
        const double per = (double)h / (100.0 * 64.0);
        printf("  %dw: %5.2f/inst/SIMD", threads / 256, per / (threads / 256));
    }
    printf("\n");
}
int main() {
    unsigned long long* d;
    hipMalloc(&d, 16);
#define F(N, NAME, TXT) run<N>(NAME, d);
    OPS(F)
#undef F
    return 0;
}
