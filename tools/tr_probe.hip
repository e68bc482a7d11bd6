// probe of ds_read_b64_tr_b16: every lane reads at its own address (lane*8 bytes: elements 4*lane .. 4*lane+3 of a u16
// array holding its own index); prints which elements each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned* out, int mode) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)lds;
    if (mode == 0) addr += lane * 8;
    else addr += ((lane & 15) >> 2) * 32 + (lane & 3) * 8 + (lane >> 4) * 128;  // [4 rows][16 cols] per 16-lane group
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[lane * 2] = v[0];
    out[lane * 2 + 1] = v[1];
}
int main() {
    unsigned* d;
    hipMalloc(&d, 64 * 2 * 4);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        unsigned h[128];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l)
            printf("lane %2d: %4u %4u %4u %4u\n", l, h[2 * l] & 0xffff, h[2 * l] >> 16, h[2 * l + 1] & 0xffff, h[2 * l + 1] >> 16);
    }
    return 0;
}
