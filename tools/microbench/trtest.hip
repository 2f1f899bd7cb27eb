#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const short* in, short* out) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + l * 4));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short h[4096], o[256];
    for (int i = 0; i < 4096; ++i) h[i] = (short)i;
    short *d, *e;
    hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, e);
    hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
    return 0;
}
