// Round 4: exhaustive check of the two-operation exact fp16 quotient of fq_quant8_h16 (fq_common.hpp):
//   rhi = RN32(1 / s), rlo = RN32(fma(-s, rhi, 1) rhi);  t = fma(x, rhi, RN32(x rlo));  RN16(t) == x /h s  (native _Float16 division,
//   itself checked against the correctly rounded quotient for all 2^32 pairs by tools/microbench/h16div.hip)
// for EVERY finite fp16 x and EVERY positive finite fp16 s (65536 x 31743 pairs), through the same v_fma_mix_f32 instructions.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/microbench/h16div2 tools/microbench/h16div2.hip && gpurun -- tools/microbench/h16div2
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../flatquant_amd/csrc/fq_common.hpp"

__global__ void check(unsigned long long* bad, unsigned* first) {
    const unsigned sbits = blockIdx.x + 1;          // 1 .. 0x7BFF: every positive finite fp16 scale (subnormals included)
    const f16 s = __builtin_bit_cast(f16, (unsigned short)sbits);
    const FqH16Recip rc = fq_h16_recip((float)s);
    unsigned long long nbad = 0;
    for (unsigned xb = threadIdx.x * 2; xb < 65536; xb += blockDim.x * 2) {
        const uint32_t xpair = xb | ((xb + 1) << 16);
        float t0, t1, l0, l1;
        uint32_t h;
        asm("v_fma_mix_f32 %[l0], %[x], %[rlo], 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[l1], %[x], %[rlo], 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[t0], %[x], %[rhi], %[l0] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[t1], %[x], %[rhi], %[l1] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_cvt_pk_f16_f32 %[h], %[t0], %[t1]"
            : [h] "=&v"(h), [t0] "=&v"(t0), [t1] "=&v"(t1), [l0] "=&v"(l0), [l1] "=&v"(l1)
            : [x] "v"(xpair), [rhi] "v"(rc.hi), [rlo] "v"(rc.lo));
        for (int e = 0; e < 2; ++e) {
            const unsigned short xs = (unsigned short)(xb + e);
            if ((xs & 0x7C00) == 0x7C00) continue;   // inf / nan inputs are not part of the contract
            const f16 x = __builtin_bit_cast(f16, xs);
            const f16 want = x / s;
            const unsigned short got = (unsigned short)(h >> (16 * e));
            unsigned short w = __builtin_bit_cast(unsigned short, want);
            // a zero quotient: the sign of zero does not reach the digits (rint(+-0) = 0)
            if ((w & 0x7FFF) == 0 && (got & 0x7FFF) == 0) continue;
            if (got != w) {
                ++nbad;
                if (atomicAdd(first, 1u) < 8) printf("x=%04x s=%04x got %04x want %04x\n", xs, sbits, got, w);
            }
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    unsigned long long* bad;
    unsigned* first;
    hipMalloc(&bad, 8);
    hipMalloc(&first, 4);
    hipMemset(bad, 0, 8);
    hipMemset(first, 0, 4);
    check<<<0x7BFF, 256>>>(bad, first);
    unsigned long long h = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("two-operation exact fp16 quotient: %llu mismatches over 65536 x 31743 (x, s) pairs (finite x, positive finite s)\n", h);
    return h != 0;
}
