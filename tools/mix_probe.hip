// probe: x - fp16(x) through v_fma_mix_f32 (half operand * -1.0 + x) equals v_cvt_f32_f16 + v_sub_f32 bit for bit
#include <hip/hip_runtime.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float res_lo(unsigned h, float x) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x)); return r; }
__device__ __forceinline__ float res_hi(unsigned h, float x) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x)); return r; }
__global__ void k(const float* in, unsigned* o2, float* o3) {
    float a = in[threadIdx.x], b = in[threadIdx.x + 64];
    h2 h = __builtin_convertvector(f2{a, b}, h2);
    unsigned hu = __builtin_bit_cast(unsigned, h);
    float ra = res_lo(hu, a), rb = res_hi(hu, b);
    h2 m = __builtin_convertvector(f2{ra, rb}, h2);
    o2[threadIdx.x] = hu;
    o2[threadIdx.x + 64] = __builtin_bit_cast(unsigned, m);
    o3[threadIdx.x] = ra; o3[threadIdx.x + 64] = rb;
    o3[threadIdx.x + 128] = a - (float)h[0]; o3[threadIdx.x + 192] = b - (float)h[1];
}
int main() {
    float hin[128]; for (int i = 0; i < 128; ++i) hin[i] = 0.1f * i - 3.3333f + 1e-3f * i * i;
    float *din, *d3; unsigned* d2; hipMalloc(&din, 512); hipMalloc(&d2, 512); hipMalloc(&d3, 1024);
    hipMemcpy(din, hin, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, d2, d3);
    float h3[256]; hipMemcpy(h3, d3, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 64; ++i) { if (h3[i] != h3[128 + i] || h3[64 + i] != h3[192 + i]) ++bad; }
    printf("mismatches %d (sample %g %g | %g %g)\n", bad, h3[5], h3[133], h3[69], h3[197]);
    return 0;
}
