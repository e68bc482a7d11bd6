// cu_exchange_probe.hip — what does it cost two CUs to hand each other 16 KB per "layer" through L2?
//
// VERDICT r3 #3 proposes pairing CUs on one 32-ray tile of the 4096-ray training step (each CU streams and multiplies HALF of a
// layer's output columns: 12.5 MB per chain instead of 25.1), which needs the two CUs to exchange their half of the layer's
// output — 32 rays x 128 features x (fp16 hi + mid) = 16 KiB per direction — after EVERY layer (87 per chain).  A layer takes
// ~3.0 us today (weight-stream bound, ~45 B/clk per CU); with half the stream ~1.5 us.  This probe measures the exchange alone,
// under the load pattern of the real thing: 256 workgroups (one per CU: 100 KiB of LDS each), paired, every pair ping-ponging
// at the same time.  Per round a workgroup writes its 16 KiB payload (256 threads x 64 B), releases a flag (agent scope), spins
// on its partner's flag (acquire), reads the partner's 16 KiB.  Both directions run concurrently, so one round = one one-way
// hand-over latency (payload write -> visible + flag -> payload read).
//   pairing "xcd":  partner = wg ^ 8   (workgroup ids go round-robin over the 8 XCDs: same XCD, same L2)
//   pairing "next": partner = wg ^ 1   (neighbouring XCDs: through the fabric / MALL)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/cu_exchange_probe tools/cu_exchange_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// mode 1: no agent-scope fences at all.  Within one XCD the L2 is the point of coherence, so it is enough that nothing is served
// from a CU's L1: payload and flag are stored plainly (the L1 is write-through; vmcnt(0) = acknowledged by L2) and loaded with
// sc1 (miss the L1).  The compiler's agent-scope release / acquire (mode 0) instead write the XCD's whole L2 back / invalidate
// it (buffer_wbl2 sc1, buffer_inv sc1): the price of coherence ACROSS the XCDs' L2s, which a same-XCD pair does not need.
__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ u32x4 ld4_sc1(const u32x4* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__global__ __launch_bounds__(256, 1) void exchange_l2_kernel(u32x4* boxes, unsigned* flags, int rounds, int partner_xor, int payload_units,
                                                             unsigned long long* cycles, unsigned* sink) {
    extern __shared__ unsigned char pad[];
    const int wg = blockIdx.x, other = wg ^ partner_xor, t = threadIdx.x;
    u32x4* mine = boxes + (size_t)wg * 2 * 1024;
    const u32x4* theirs = boxes + (size_t)other * 2 * 1024;
    unsigned acc = 0, bad = 0;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        const int par = r & 1;
        for (int u = 0; u < payload_units; ++u) mine[par * 1024 + u * 256 + t] = u32x4{(unsigned)r, (unsigned)t, acc, (unsigned)u};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's payload is in L2
        __syncthreads();                                   // ... everybody's
        if (t == 0) {
            flags[wg * 32] = (unsigned)r;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int spin = 0;  // (bounded: a stale flag must end as an error count, not as a hung GPU)
            while (ld_sc1(flags + other * 32) < (unsigned)r && ++spin < 400000) {}
            if (spin >= 400000) atomicAdd(sink + 1, 1000000u);
        }
        __syncthreads();
        for (int u = 0; u < payload_units; ++u) {
            const u32x4 v = ld4_sc1(theirs + par * 1024 + u * 256 + t);
            bad |= (v[0] != (unsigned)r) | (v[1] != (unsigned)t);  // the payload really is this round's
            acc += v[0] + v[2];
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (t == 0) cycles[wg] = t1 - t0;
    if (bad) atomicAdd(sink + 1, 1u);
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256, 1) void exchange_kernel(u32x4* boxes, unsigned* flags, int rounds, int partner_xor, int payload_units,
                                                          unsigned long long* cycles, unsigned* sink) {
    extern __shared__ unsigned char pad[];  // keeps one workgroup per CU
    const int wg = blockIdx.x, other = wg ^ partner_xor, t = threadIdx.x;
    u32x4* mine = boxes + (size_t)wg * 2 * 1024;     // two 16 KiB boxes per workgroup (double buffered: round parity)
    const u32x4* theirs = boxes + (size_t)other * 2 * 1024;
    unsigned acc = 0;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        const int par = r & 1;
        for (int u = 0; u < payload_units; ++u)  // payload_units x 4 KiB (256 threads x 16 B)
            mine[par * 1024 + u * 256 + t] = u32x4{(unsigned)r, (unsigned)t, acc, (unsigned)u};
        __threadfence();  // payload visible at agent scope ...
        __syncthreads();  // ... from every thread, before the flag
        if (t == 0) __hip_atomic_store(flags + wg * 32, (unsigned)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (t == 0)
            while (__hip_atomic_load(flags + other * 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) {}
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every thread: drop stale L1 lines before reading the payload
        for (int u = 0; u < payload_units; ++u) {
            const u32x4 v = theirs[par * 1024 + u * 256 + t];
            acc += v[0] + v[2];
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (t == 0) cycles[wg] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    int n_cu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) n_cu = prop.multiProcessorCount;
    int wc_khz = 100000;
    (void)hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
    u32x4* boxes; unsigned* flags; unsigned long long* cycles; unsigned* sink;
    hipMalloc(&boxes, (size_t)n_cu * 2 * 16384);
    hipMalloc(&flags, (size_t)n_cu * 32 * 4);
    hipMalloc(&cycles, (size_t)n_cu * 8);
    hipMalloc(&sink, 8);
    hipMemset(sink, 0, 8);
    printf("cu_exchange_probe: %d workgroups (one per CU), %d rounds, wall clock %d kHz\n", n_cu, rounds, wc_khz);
    const struct { const char* name; int x; int mode; } pairings[] = {
        {"agent-scope fences, same XCD (wg ^ 8)", 8, 0}, {"agent-scope fences, next XCD (wg ^ 1)", 1, 0},
        {"L2-coherent (sc1 loads), same XCD (wg ^ 8)", 8, 1}};
    for (const auto& p : pairings)
        for (int units : {0, 1, 2, 4}) {  // flag only, 4, 8, 16 KiB per direction and round
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(flags, 0, (size_t)n_cu * 32 * 4);
                if (p.mode == 0)
                    hipLaunchKernelGGL(exchange_kernel, dim3(n_cu), dim3(256), 100 * 1024, 0, boxes, flags, rounds, p.x, units, cycles, sink);
                else
                    hipLaunchKernelGGL(exchange_l2_kernel, dim3(n_cu), dim3(256), 100 * 1024, 0, boxes, flags, rounds, p.x, units, cycles, sink);
                if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
            }
            std::vector<unsigned long long> c(n_cu);
            hipMemcpy(c.data(), cycles, (size_t)n_cu * 8, hipMemcpyDeviceToHost);
            double mx = 0, sum = 0;
            for (auto v : c) { sum += (double)v; if ((double)v > mx) mx = (double)v; }
            const double us_mean = sum / n_cu / rounds / (wc_khz * 1e-3), us_max = mx / rounds / (wc_khz * 1e-3);
            printf("  %-44s payload %2d KiB/direction: %.3f us per round (mean over workgroups), %.3f us (slowest)\n", p.name,
                   units * 4, us_mean, us_max);
        }
    unsigned h[2] = {0, 0};
    hipMemcpy(h, sink, 8, hipMemcpyDeviceToHost);
    printf("  stale payloads seen by the L2-coherent mode: %u (must be 0)\n", h[1]);
    return 0;
}
