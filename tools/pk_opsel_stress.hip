// pk_opsel_stress.hip — stand-alone probe for the fault seen in the cooperative training forward (DESIGN.md §2,
// csrc/r2l_coopf.h FC_SOLO_LDS_BYTES): does v_pk_fma_f32 with op_sel:[0,1,0] (src1's HIGH dword feeding the LOW lane of
// the packed FMA), reading an operand pair whose low half a v_mov_b32 wrote one instruction earlier, ever return a wrong
// result when other waves share the SIMD — plain VALU waves, or waves that issue MFMAs back to back?  Each wave checks
// the packed chain against scalar v_fma_f32 chains on the same operands, bit for bit, and counts mismatching lanes.
//   hipcc --offload-arch=gfx950 -O2 tools/pk_opsel_stress.hip -o tools/_bin/pk_opsel_stress && tools/_bin/pk_opsel_stress
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// block = 512 threads: waves 0..3 and 4..7 land on the same four SIMDs
__global__ __launch_bounds__(512) void stress(unsigned* bad, float* sink, int iters, int mfma_mask, int lds_bytes_touch) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lds_bytes_touch > 0) lds[threadIdx.x] = 0.f;
    if ((mfma_mask >> wave) & 1) {  // this wave keeps the matrix pipe busy
        f32x16 acc = {};
        f16x8 a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.001f * (lane + k)); b[k] = (_Float16)(0.002f * (lane - k)); }
        if (lds_bytes_touch < 0) {
            // ... the way the cooperative chain does: A operands streaming in from L2 (16 B per lane, several loads in
            // flight), B operands from LDS, MFMAs on both
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4* src = reinterpret_cast<const u32x4*>(sink) + lane;
            u32x4* l = reinterpret_cast<u32x4*>(lds) + (threadIdx.x & 255);
            *l = u32x4{1u, 2u, 3u, 4u};
            for (int i = 0; i < iters; ++i) {
                u32x4 w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = __builtin_nontemporal_load(src + 64 * ((i * 4 + u) & 1023));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const u32x4 bb = *(volatile u32x4*)(l + 0);
                    a = __builtin_bit_cast(f16x8, w[u]);
                    b = __builtin_bit_cast(f16x8, bb);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
                }
            }
            if (acc[0] == 12345.f) sink[0] = acc[3];
            return;
        }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
        if (acc[0] == 12345.f) sink[0] = acc[3];
        return;
    }
    float w0a = 0.37f + 0.001f * lane, w1a = -0.21f + 0.002f * lane, w0b = 0.11f - 0.003f * lane, w1b = 0.05f + 0.004f * lane;
    f32x2 y = {1.0f + 0.01f * lane, -0.5f + 0.02f * lane};
    unsigned nbad = 0;
    for (int i = 0; i < iters; ++i) {
        f32x2 acc = {0.f, 0.f};
        float r0 = 0.f, r1 = 0.f, d = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // the training forward's tail: the (c0 | c1) weight pair is completed by a v_mov_b32 into its LOW half one
            // instruction ahead of the packed FMA; the multiplicand pair is used low half first, then high half
            asm volatile(
                "v_mov_b32 v200, %[w0a]\n\t"
                "v_mov_b32 v201, %[w1a]\n\t"
                "v_fma_f32 %[d], %[w0b], %[y0], %[d]\n\t"
                "v_pk_fma_f32 %[acc], v[200:201], %[y], %[acc] op_sel_hi:[1,0,1]\n\t"
                "v_mov_b32 v201, %[w1b]\n\t"
                "v_mov_b32 v200, %[w0b]\n\t"
                "v_fmac_f32 %[d], %[w1b], %[y1]\n\t"
                "v_pk_fma_f32 %[acc], v[200:201], %[y], %[acc] op_sel:[0,1,0]\n\t"
                : [acc] "+v"(acc), [d] "+v"(d)
                : [w0a] "v"(w0a), [w1a] "v"(w1a), [w0b] "v"(w0b), [w1b] "v"(w1b), [y] "v"(y), [y0] "v"(y[0]), [y1] "v"(y[1])
                : "v200", "v201");
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r0) : "v"(w0a), "v"(y[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r1) : "v"(w1a), "v"(y[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r0) : "v"(w0b), "v"(y[1]));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r1) : "v"(w1b), "v"(y[1]));
            w0a += 0.0001f; w1b -= 0.0002f;  // fresh operands every round: a stale read shows
        }
        if (__float_as_uint(acc[0]) != __float_as_uint(r0)) nbad += 1u;
        if (__float_as_uint(acc[1]) != __float_as_uint(r1)) nbad += 1u << 16;
        y[0] += 0.001f; y[1] -= 0.001f;
        if (d == 12345.f) sink[0] = d;
    }
    if (nbad) atomicAdd(bad + (lane >> 4), nbad);  // per 16-lane quarter: low-lane mismatches in bits 0-15, high-lane in 16-31
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned* bad;
    float* sink;
    hipMalloc(&bad, 16);
    hipMalloc(&sink, 4096 * 512 * 4);
    hipMemset(sink, 0, 4096 * 512 * 4);
    struct { const char* name; int grid, mask, lds; } cfg[] = {
        {"1 block/CU, all VALU waves (2 per SIMD)", 256, 0x00, 0},
        {"1 block/CU, waves 4-7 issue MFMAs (1 VALU + 1 MFMA wave per SIMD)", 256, 0xF0, 0},
        {"4 blocks/CU, all VALU waves (8 per SIMD)", 1024, 0x00, 0},
        {"4 blocks/CU, half the waves issue MFMAs", 1024, 0xF0, 0},
        {"2 blocks/CU held apart by 72 KiB LDS each, half MFMA", 512, 0xF0, 72 * 1024},
        {"1 block/CU, waves 4-7: L2 loads + LDS reads + MFMAs (the chain's mix)", 256, 0xF0, -1},
        {"2 blocks/CU, waves 4-7: L2 loads + LDS reads + MFMAs", 512, 0xF0, -1},
        {"2 blocks/CU, waves 1,3,5,7: L2 loads + LDS reads + MFMAs", 512, 0xAA, -1},
    };
    for (auto& c : cfg) {
        hipMemset(bad, 0, 16);
        hipFuncSetAttribute((const void*)stress, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(stress, dim3(c.grid), dim3(512), c.lds > 0 ? c.lds : 4096, 0, bad, sink, iters, c.mask, c.lds);
        hipError_t e = hipDeviceSynchronize();
        unsigned h[4];
        hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
        printf("%-72s %s  mismatching checks per 16-lane quarter (low lane | high lane): ", c.name, hipGetErrorString(e));
        for (int q = 0; q < 4; ++q) printf("%u|%u ", h[q] & 0xffffu, h[q] >> 16);
        printf("\n");
    }
    return 0;
}
