// pk_opsel_mfma_probe.hip — stand-alone probe for the gfx950 hazard of DESIGN.md §2 / profiles/r03_coresidency.md: does
// `v_pk_fma_f32 ... op_sel:[0,1,0]` (low lane fed by src1's HIGH dword) lose its low-lane result when ANOTHER WORKGROUP's wave on
// the same SIMD issues v_mfma_f32_32x32x16_f16?  Two workgroups per CU (72 KiB of LDS each, ~250 VGPRs per wave: exactly one
// victim wave and one aggressor wave per SIMD): workgroups < grid/2 are victims (packed-FMA chains checked bit for bit against
// scalar FMA chains), the others aggressors (mode 0: idle spin, 1: MFMAs from registers, 2: MFMAs fed by LDS reads, 3: + L2
// loads, 4: + a barrier per 16 MFMAs — the cooperative chain's mix).
//   hipcc --offload-arch=gfx950 -O2 tools/pk_opsel_mfma_probe.hip -o tools/_bin/pk_opsel_mfma_probe && tools/_bin/pk_opsel_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// FORM of the victim's packed op (acc = (lo, hi), A = (w0, w1), Y = (y0, y1)):
//   0  v_pk_fma_f32 acc, A, Y, acc                          lo = w0*y0 + lo          hi = w1*y1 + hi      (no swizzle)
//   1  ... op_sel_hi:[1,0,1]                                lo = w0*y0 + lo          hi = w1*y0 + hi      (high lane takes Y's LOW dword)
//   2  ... op_sel:[0,1,0]                                   lo = w0*y1 + lo          hi = w1*y1 + hi      (LOW lane takes Y's HIGH dword)
//   3  ... op_sel:[1,0,0]                                   lo = w1*y0 + lo          hi = w1*y1 + hi      (low lane takes A's high dword)
//   4  ... op_sel:[0,0,1]                                   lo = w0*y0 + hi_old      hi = w1*y1 + hi      (low lane takes acc's high dword)
//   5  v_pk_mul_f32 t, A, Y op_sel:[0,1]; v_pk_add_f32 acc   lo += w0*y1              hi += w1*y1          (packed MUL, low lane <- src1 high)
//   6  v_pk_add_f32 acc, acc, P op_sel:[0,1] op_sel_hi:[1,0] lo += p1                 hi += p0             (the swap-add of r2l_stratified_z)
template <int FORM>
__global__ __launch_bounds__(256, 2) void probe(unsigned* bad, const float* src, float* sink, int iters, int mode) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[72 * 1024];
    const int lane = threadIdx.x & 63;
    const bool victim = blockIdx.x < gridDim.x / 2;
    reinterpret_cast<u32x4*>(lds)[threadIdx.x] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    __syncthreads();
    if (!victim) {
        f32x16 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[t][c] = 0.f;
        f16x8 a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.001f * (lane + k)); b[k] = (_Float16)(0.002f * (lane - k)); }
        const u32x4* s4 = reinterpret_cast<const u32x4*>(src) + lane;
        for (int i = 0; i < iters; ++i) {
            if (mode == 0) { __builtin_amdgcn_s_sleep(64); continue; }
            if (mode >= 3) a = __builtin_bit_cast(f16x8, __builtin_nontemporal_load(s4 + 64 * (i & 1023)));
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (mode >= 2) b = __builtin_bit_cast(f16x8, *(volatile u32x4*)(reinterpret_cast<u32x4*>(lds) + ((lane + 64 * t) & 255)));
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[t], 0, 0, 0);
            }
            if (mode >= 4) __syncthreads();
        }
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) s += acc[t][3];
        if (s == 12345.f) sink[0] = s;
        return;
    }
    float w0 = 0.37f + 0.001f * lane, w1 = -0.21f + 0.002f * lane;
    f32x2 y = {1.0f + 0.01f * lane, -0.5f + 0.02f * lane};
    unsigned nbad_lo = 0, nbad_hi = 0;
    for (int i = 0; i < iters * 4; ++i) {
        f32x2 acc = {0.f, 0.f};
        float r0 = 0.f, r1 = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            // operand pair completed by v_movs right in front of the packed op, as hipcc does; v250 / v251 also force ~250 VGPRs
#define MOVS "v_mov_b32 v251, %[w1]\n\tv_mov_b32 v250, %[w0]\n\t"
#define OPS : [acc] "+v"(acc) : [w0] "v"(w0), [w1] "v"(w1), [y] "v"(y) : "v250", "v251", "v248", "v249"
            if (FORM == 0) asm volatile(MOVS "v_pk_fma_f32 %[acc], v[250:251], %[y], %[acc]" OPS);
            if (FORM == 1) asm volatile(MOVS "v_pk_fma_f32 %[acc], v[250:251], %[y], %[acc] op_sel_hi:[1,0,1]" OPS);
            if (FORM == 2) asm volatile(MOVS "v_pk_fma_f32 %[acc], v[250:251], %[y], %[acc] op_sel:[0,1,0]" OPS);
            if (FORM == 3) asm volatile(MOVS "v_pk_fma_f32 %[acc], v[250:251], %[y], %[acc] op_sel:[1,0,0]" OPS);
            if (FORM == 4) asm volatile(MOVS "v_pk_fma_f32 %[acc], v[250:251], %[y], %[acc] op_sel:[0,0,1]" OPS);
            if (FORM == 5) asm volatile(MOVS "v_pk_mul_f32 v[248:249], v[250:251], %[y] op_sel:[0,1]\n\tv_pk_add_f32 %[acc], %[acc], v[248:249]" OPS);
            if (FORM == 6) asm volatile(MOVS "v_pk_mul_f32 v[248:249], v[250:251], %[y]\n\tv_pk_add_f32 %[acc], %[acc], v[248:249] op_sel:[0,1] op_sel_hi:[1,0]" OPS);
#undef MOVS
#undef OPS
            const float h_old = r1;
            if (FORM == 0) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r0) : "v"(w0), "v"(y[0])); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r1) : "v"(w1), "v"(y[1])); }
            if (FORM == 1) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r0) : "v"(w0), "v"(y[0])); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r1) : "v"(w1), "v"(y[0])); }
            if (FORM == 2) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r0) : "v"(w0), "v"(y[1])); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r1) : "v"(w1), "v"(y[1])); }
            if (FORM == 3) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r0) : "v"(w1), "v"(y[0])); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r1) : "v"(w1), "v"(y[1])); }
            if (FORM == 4) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(w0), "v"(y[0]), "v"(h_old)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r1) : "v"(w1), "v"(y[1])); }
            if (FORM == 5) { float p0, p1; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(w0), "v"(y[1])); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(w1), "v"(y[1]));
                             asm volatile("v_add_f32 %0, %0, %1" : "+v"(r0) : "v"(p0)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(r1) : "v"(p1)); }
            if (FORM == 6) { float p0, p1; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(w0), "v"(y[0])); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(w1), "v"(y[1]));
                             asm volatile("v_add_f32 %0, %0, %1" : "+v"(r0) : "v"(p1)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(r1) : "v"(p0)); }
            w0 += 0.0001f; w1 -= 0.0002f;
        }
        if (__float_as_uint(acc[0]) != __float_as_uint(r0)) ++nbad_lo;
        if (__float_as_uint(acc[1]) != __float_as_uint(r1)) ++nbad_hi;
        y[0] += 0.001f; y[1] -= 0.001f;
    }
    if (nbad_lo) atomicAdd(bad + 2 * (lane >> 4), nbad_lo);
    if (nbad_hi) atomicAdd(bad + 2 * (lane >> 4) + 1, nbad_hi);
}

template <int FORM>
static void run(const char* form, unsigned* bad, float* src, float* sink, int iters) {
    const char* names[] = {"idle spin", "MFMAs from registers", "MFMAs fed by LDS reads"};
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(bad, 0, 32);
        hipLaunchKernelGGL(probe<FORM>, dim3(512), dim3(256), 0, 0, bad, src, sink, iters, mode);
        hipError_t e = hipDeviceSynchronize();
        unsigned h[8];
        hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost);
        printf("%-44s | neighbour: %-22s %s | wrong chains per 16-lane quarter (low|high lane): ", form, names[mode],
               e == hipSuccess ? "" : hipGetErrorString(e));
        for (int q = 0; q < 4; ++q) printf("%u|%u ", h[2 * q], h[2 * q + 1]);
        printf("\n");
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 5000;
    unsigned* bad;
    float *src, *sink;
    hipMalloc(&bad, 32);
    hipMalloc(&src, 4096 * 1024);
    hipMalloc(&sink, 64);
    hipMemset(src, 0, 4096 * 1024);
    run<0>("0 v_pk_fma_f32 (no op_sel)", bad, src, sink, iters);
    run<1>("1 v_pk_fma_f32 op_sel_hi:[1,0,1]", bad, src, sink, iters);
    run<2>("2 v_pk_fma_f32 op_sel:[0,1,0]", bad, src, sink, iters);
    run<3>("3 v_pk_fma_f32 op_sel:[1,0,0]", bad, src, sink, iters);
    run<4>("4 v_pk_fma_f32 op_sel:[0,0,1]", bad, src, sink, iters);
    run<5>("5 v_pk_mul_f32 op_sel:[0,1] + v_pk_add_f32", bad, src, sink, iters);
    run<6>("6 v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]", bad, src, sink, iters);
    return 0;
}
