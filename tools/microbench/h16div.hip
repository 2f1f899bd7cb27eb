// Exhaustive check on gfx950: is hipcc's native _Float16 division (cvt, rcp_f32, mul, cvt, div_fixup_f16 or its refined
// form) identical to the correctly rounded quotient fp16(fp32(a) / fp32(b)) (IEEE fp32 division, then RNE) for ALL
// pairs of finite fp16 values?  (Decides whether the fp16-arithmetic Quantizer path may use it.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16;
__global__ void k(unsigned long long* bad, unsigned* first) {
    const unsigned bb = blockIdx.x * blockDim.x + threadIdx.x;  // bit pattern of b
    if (bb >= 65536) return;
    const f16 b = __builtin_bit_cast(f16, (unsigned short)bb);
    if (!(b == b) || (bb & 0x7fff) == 0x7c00 || (bb & 0x7fff) == 0) return;  // NaN / Inf / zero divisors
    unsigned long long n = 0;
    for (unsigned ab = 0; ab < 65536; ++ab) {
        const f16 a = __builtin_bit_cast(f16, (unsigned short)ab);
        if (!(a == a) || (ab & 0x7fff) == 0x7c00) continue;
        const f16 q1 = a / b;
        float fq = (float)a / (float)b;
        asm volatile("" : "+v"(fq));
        const f16 q2 = (f16)fq;
        const unsigned short u1 = __builtin_bit_cast(unsigned short, q1), u2 = __builtin_bit_cast(unsigned short, q2);
        if (u1 != u2 && !((u1 & 0x7fff) == 0 && (u2 & 0x7fff) == 0)) {
            if (n == 0) atomicCAS(first, 0u, (ab << 16) | bb);
            ++n;
        }
    }
    if (n) atomicAdd(bad, n);
}
int main() {
    unsigned long long* d; unsigned* f;
    hipMalloc(&d, 8); hipMalloc(&f, 4); hipMemset(d, 0, 8); hipMemset(f, 0, 4);
    k<<<256, 256>>>(d, f);
    unsigned long long h = 0; unsigned hf = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, f, 4, hipMemcpyDeviceToHost);
    printf("native f16 division vs correctly rounded: %llu mismatching pairs (first a=0x%04x b=0x%04x)\n", h, hf >> 16, hf & 0xffff);
    return 0;
}
