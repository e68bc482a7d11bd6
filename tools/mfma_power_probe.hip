// mfma_power_probe.hip — what does a dense 16-bit MFMA stream sustain on this chip, and does it depend on the operand type and
// on the operand DATA?  The library's chains deliver ~1.1 - 1.3 PF/s of fp16 products against the 2.5 PF/s datasheet figure and
// DESIGN.md argues that the chip's power cap, not instruction issue, sets that.  This probe takes everything else away: every
// SIMD of every CU runs one wave that issues nothing but v_mfma_f32_32x32x16_{f16,bf16} on register operands (8 independent
// accumulator tiles, no memory, no LDS, no VALU in the loop) for ~25 ms per case, and reports the sustained product rate for
//   type  : fp16 | bf16
//   data  : zero operands | a few set bits | random mantissas of "hi"-like values (|x| ~ 1) | "mid"-like values (|x| ~ 2^-12)
// If the rate drops with operand entropy at equal instruction streams, the bound is energy; if bf16 sustains more than fp16 on
// the same data class, a bf16 product is cheaper than an fp16 one.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_power_probe.hip -o tools/_bin/mfma_power_probe && tools/_bin/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ROT: every tile has its own A operand and B alternates between two (operand buses toggle like in a GEMM); else one (a, b)
// pair feeds every MFMA
template <bool BF16, bool ROT>
__global__ __launch_bounds__(256, 1) void stream_kernel(const u32x4* __restrict__ ops, float* sink, int iters) {
    // operands: 16 bytes per lane and quad, prepared on the host (bit patterns of the chosen type / data class)
    u32x4 qa[8], qb[2];
#pragma unroll
    for (int t = 0; t < 8; ++t) qa[t] = ops[(ROT ? t : 0) * 256 + threadIdx.x];
    qb[0] = ops[8 * 256 + threadIdx.x];
    qb[1] = ops[(ROT ? 9 : 8) * 256 + threadIdx.x];
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[t][c] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const u32x4 ua = qa[t], ub = qb[r & 1];
                if (BF16)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), acc[t], 0, 0, 0);
                else
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ua), __builtin_bit_cast(f16x8, ub), acc[t], 0, 0, 0);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s += acc[t][5];
    if (s == 12345.678f) sink[0] = s;
}

// the same stream with the A operands re-read from LDS at the render kernel's ratio: 2 x ds_read_b128 (hi, mid) per 3 MFMAs
// (PROBE_LDS=1 with PROBE_ONE): what the LDS -> register traffic of the weights costs in sustained rate
__global__ __launch_bounds__(256, 1) void stream_lds_kernel(const u32x4* __restrict__ ops, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[6 * 1024];  // 96 KiB of operand data (random fp16 bit patterns)
    for (int i = threadIdx.x; i < 6 * 1024; i += 256) lds[i] = ops[(i % 8) * 256 + (i * 7 + threadIdx.x) % 256];
    __syncthreads();
    const u32x4 ub_h = ops[8 * 256 + threadIdx.x], ub_m = ops[9 * 256 + threadIdx.x];
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[t][c] = 0.f;
    const int lane = threadIdx.x & 63;
    // software pipeline: the 16 A quads of iteration i + 1 are read while the 24 MFMAs of iteration i run (ping-pong sets)
    u32x4 a0[16], a1[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) a0[q] = lds[q * 64 + lane];
#define LDS_STEP(CUR, NXT, BASE)                                                                                              \
    _Pragma("unroll") for (int t = 0; t < 8; ++t) {                                                                           \
        NXT[2 * t] = lds[(BASE) + t * 128 + lane];                                                         \
        NXT[2 * t + 1] = lds[(BASE) + t * 128 + 64 + lane];                                                \
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, CUR[2 * t + 1]), __builtin_bit_cast(f16x8, ub_h), acc[t], 0, 0, 0); \
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, CUR[2 * t]), __builtin_bit_cast(f16x8, ub_m), acc[t], 0, 0, 0);     \
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, CUR[2 * t]), __builtin_bit_cast(f16x8, ub_h), acc[t], 0, 0, 0);     \
    }
    for (int i = 0; i < iters; i += 2) {
        const int b1 = ((i + 1) & 3) * 1536, b2 = ((i + 2) & 3) * 1536;
        LDS_STEP(a0, a1, b1)
        LDS_STEP(a1, a0, b2)
    }
#undef LDS_STEP
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s += acc[t][5];
    if (s == 12345.678f) sink[0] = s;
}

static unsigned short f2h(float f) {  // round-to-nearest fp16 bits (normal range only; enough for the probe)
    _Float16 h = (_Float16)f;
    unsigned short u;
    memcpy(&u, &h, 2);
    return u;
}
static unsigned short f2b(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

int main() {
    const int threads = 256, wgs = 256, n16 = threads * 8 * 10;  // ten 16-byte quads per lane
    unsigned short* host = (unsigned short*)malloc(n16 * 2);
    u32x4* dev;
    float* sink;
    hipMalloc(&dev, n16 * 2);
    hipMalloc(&sink, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* classes[4] = {"zeros", "one bit per value (1.0 / 2^-12)", "random, |x| ~ 1 (hi-like)", "random, |x| ~ 2^-12 (mid-like)"};
    const int long_run = getenv("PROBE_LONG") ? atoi(getenv("PROBE_LONG")) : 1;  // x the ~25 ms base duration
    if (getenv("PROBE_ONE")) {  // PROBE_ONE="<bf16 0|1> <class 0..3> <rotating 0|1> <mantissa bits kept>": one case, for tools/mfma_power_trace.py
        int bf = 0, cls = 2, rot = 1, keep = 10;
        sscanf(getenv("PROBE_ONE"), "%d %d %d %d", &bf, &cls, &rot, &keep);
        srand(7);
        const unsigned short mask = (unsigned short)(0xffffu << (10 - keep));
        for (int i = 0; i < n16; ++i) {
            float v = 0.f;
            const float sgn = (rand() & 1) ? 1.f : -1.f, r = 0.5f + 0.5f * (float)rand() / (float)RAND_MAX;
            if (cls == 1) v = ((i / (256 * 8)) >= 8) ? 1.0f : 0.000244140625f;
            if (cls == 2) v = sgn * 2.f * r;
            if (cls == 3) v = sgn * r * 0.000244140625f;
            host[i] = bf ? f2b(v) : (unsigned short)(f2h(v) & mask);
        }
        hipMemcpy(dev, host, n16 * 2, hipMemcpyHostToDevice);
        const int iters = 60000 * long_run;
        const bool lds_fed = getenv("PROBE_LDS") != nullptr;
        const int it2 = lds_fed ? iters * 32 / 24 : iters;  // (24 MFMAs per iteration instead of 32)
        hipEventRecord(e0);
        if (lds_fed) hipLaunchKernelGGL(stream_lds_kernel, dim3(wgs), dim3(threads), 0, 0, dev, sink, it2);
        else if (bf && rot) hipLaunchKernelGGL((stream_kernel<true, true>), dim3(wgs), dim3(threads), 0, 0, dev, sink, iters);
        else if (bf) hipLaunchKernelGGL((stream_kernel<true, false>), dim3(wgs), dim3(threads), 0, 0, dev, sink, iters);
        else if (rot) hipLaunchKernelGGL((stream_kernel<false, true>), dim3(wgs), dim3(threads), 0, 0, dev, sink, iters);
        else hipLaunchKernelGGL((stream_kernel<false, false>), dim3(wgs), dim3(threads), 0, 0, dev, sink, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double nm = lds_fed ? (double)it2 * 24.0 : (double)iters * 32.0;
        printf("%.2f ms %.3f PF/s\n", ms, (double)wgs * 4 * nm * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e15);
        return 0;
    }
    printf("%-6s %-9s %-36s %10s %12s\n", "type", "operands", "operand data", "ms", "PF/s");
    for (int rot = 0; rot < 2; ++rot)
    for (int bf = 0; bf < 2; ++bf)
        for (int cls = 0; cls < 4; ++cls) {
            if (rot && cls < 2) continue;
            srand(7);
            for (int i = 0; i < n16; ++i) {
                float v = 0.f;
                const float sgn = (rand() & 1) ? 1.f : -1.f, r = 0.5f + 0.5f * (float)rand() / (float)RAND_MAX;
                if (cls == 1) v = ((i / (256 * 8)) >= 8) ? 1.0f : 0.000244140625f;
                if (cls == 2) v = sgn * 2.f * r;
                if (cls == 3) v = sgn * r * 0.000244140625f;
                host[i] = bf ? f2b(v) : f2h(v);
            }
            hipMemcpy(dev, host, n16 * 2, hipMemcpyHostToDevice);
            const int iters = 60000 * long_run;  // x 32 MFMAs of 32 cycles: ~61 M cycles = ~25 - 30 ms per unit
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (bf && rot) hipLaunchKernelGGL((stream_kernel<true, true>), dim3(wgs), dim3(threads), 0, 0, dev, sink, iters);
                else if (bf) hipLaunchKernelGGL((stream_kernel<true, false>), dim3(wgs), dim3(threads), 0, 0, dev, sink, iters);
                else if (rot) hipLaunchKernelGGL((stream_kernel<false, true>), dim3(wgs), dim3(threads), 0, 0, dev, sink, iters);
                else hipLaunchKernelGGL((stream_kernel<false, false>), dim3(wgs), dim3(threads), 0, 0, dev, sink, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)wgs * 4 * iters * 32.0 * 2.0 * 32 * 32 * 16;
            printf("%-6s %-9s %-36s %10.2f %12.3f\n", bf ? "bf16" : "fp16", rot ? "rotating" : "fixed", classes[cls], ms, flop / (ms * 1e-3) / 1e15);
        }
    // does the energy of a product depend on how many mantissa bits of an operand are populated?  fp16, rotating operands,
    // random values; the low (10 - k) mantissa bits of B (or of A and B) cleared
    printf("\nfp16, rotating random operands, low mantissa bits cleared:\n%-28s %10s %12s\n", "mantissa bits kept", "ms", "PF/s");
    for (int both = 0; both < 2; ++both)
        for (int k = 10; k >= 0; k -= 2) {
            srand(7);
            const unsigned short mask = (unsigned short)(0xffffu << (10 - k));
            for (int i = 0; i < n16; ++i) {
                const float sgn = (rand() & 1) ? 1.f : -1.f, r = 0.5f + 0.5f * (float)rand() / (float)RAND_MAX;
                unsigned short h = f2h(sgn * 2.f * r);
                const bool is_b = (i / (256 * 8)) >= 8;
                if (is_b || both) h &= mask;
                host[i] = h;
            }
            hipMemcpy(dev, host, n16 * 2, hipMemcpyHostToDevice);
            const int iters = 60000 * long_run;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL((stream_kernel<false, true>), dim3(wgs), dim3(threads), 0, 0, dev, sink, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)wgs * 4 * iters * 32.0 * 2.0 * 32 * 32 * 16;
            printf("%s: %2d bits %14s %10.2f %12.3f\n", both ? "A and B" : "B only ", k, "", ms, flop / (ms * 1e-3) / 1e15);
        }
    return 0;
}
