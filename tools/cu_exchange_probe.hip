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
    hipMalloc(&sink, 4);
    printf("cu_exchange_probe: %d workgroups (one per CU), %d rounds, wall clock %d kHz\n", n_cu, rounds, wc_khz);
    const struct { const char* name; int x; } pairings[] = {{"same XCD (wg ^ 8)", 8}, {"neighbouring XCDs (wg ^ 1)", 1}};
    for (const auto& p : pairings)
        for (int units : {0, 1, 4}) {  // flag only, 4 KiB, 16 KiB per direction and round
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(flags, 0, (size_t)n_cu * 32 * 4);
                hipLaunchKernelGGL(exchange_kernel, dim3(n_cu), dim3(256), 100 * 1024, 0, boxes, flags, rounds, p.x, units, cycles, sink);
                if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
            }
            std::vector<unsigned long long> c(n_cu);
            hipMemcpy(c.data(), cycles, (size_t)n_cu * 8, hipMemcpyDeviceToHost);
            double mx = 0, sum = 0;
            for (auto v : c) { sum += (double)v; if ((double)v > mx) mx = (double)v; }
            const double us_mean = sum / n_cu / rounds / (wc_khz * 1e-3), us_max = mx / rounds / (wc_khz * 1e-3);
            printf("  %-28s payload %2d KiB/direction: %.3f us per round (mean over workgroups), %.3f us (slowest)\n", p.name,
                   units * 4, us_mean, us_max);
        }
    return 0;
}
