#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../flatquant_amd/csrc/fq_common.hpp"
__global__ void k(float* out) {
    int l = threadIdx.x;
    float v = (float)((l * 37 + 5) % 64);   // permutation of 0..63
    float v0 = v;
    out[l] = fq_dpp<0xB1>(v0);
    out[64 + l] = fq_dpp<0x4E>(v0);
    out[128 + l] = fq_dpp<0x141>(v0);
    out[192 + l] = fq_dpp<0x140>(v0);
    unsigned u = __builtin_bit_cast(unsigned, v0);
    unsigned w1 = u; asm volatile("" : "+v"(w1));
    auto r = __builtin_amdgcn_permlane16_swap(u, w1, false, false);
    out[256 + l] = __builtin_bit_cast(float, r[0]);
    out[320 + l] = __builtin_bit_cast(float, r[1]);
    unsigned w2 = u; asm volatile("" : "+v"(w2));
    auto r2 = __builtin_amdgcn_permlane32_swap(u, w2, false, false);
    out[384 + l] = __builtin_bit_cast(float, r2[0]);
    out[448 + l] = __builtin_bit_cast(float, r2[1]);
    out[512 + l] = fq_wave_max(v0);
    out[576 + l] = fq_wave_min(v0);
}
int main() {
    float* d; hipMalloc(&d, 640 * 4);
    k<<<1, 64>>>(d);
    float h[640]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"src", "quad[1,0,3,2]", "quad[2,3,0,1]", "half_mirror", "row_mirror", "pl16 r0", "pl16 r1", "pl32 r0", "pl32 r1", "wave_max", "wave_min"};
    printf("%-14s", names[0]); for (int l = 0; l < 64; ++l) printf("%3d", (l * 37) % 64); printf("\n");
    for (int r = 0; r < 10; ++r) { printf("%-14s", names[r + 1]); for (int l = 0; l < 64; ++l) printf("%3d", (int)h[r * 64 + l]); printf("\n"); }
    return 0;
}
