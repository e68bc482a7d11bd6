// probe: does the immediate offset of `buffer_load_dwordx4 ... offen offset:N lds` move BOTH the memory address and the LDS
// address?  Source = u32 index array; M0 = LDS base; one load with offset:1024; dump where the data landed and what it is.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const unsigned* src, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned long long sa = (unsigned long long)src;
    u32x4 rs = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sa), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sa >> 32)) & 0xffffu, 0xffffffffu, 0x00020000u};
    const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds;
    const unsigned voff = threadIdx.x * 16u;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:1024 lds\n\ts_waitcnt vmcnt(0)" : : "s"(lds_addr), "v"(voff), "s"(rs) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
    unsigned h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i;
    unsigned *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 8192); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
    unsigned r[2048]; hipMemcpy(r, o, 8192, hipMemcpyDeviceToHost);
    int first = -1, last = -1; for (int i = 0; i < 2048; ++i) if (r[i] != 0xdeadbeefu) { if (first < 0) first = i; last = i; }
    printf("written dwords [%d, %d]; lds[%d] = %u (source dword index), lds[%d] = %u\n", first, last, first, r[first], last, r[last]);
    return 0;
}
