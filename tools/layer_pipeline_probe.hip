// layer_pipeline_probe.hip — VERDICT r4 #1(b): what does a LAYER-STATIONARY pipeline of the cooperative fp16x2 chains sustain?
//
// Today's 4096-ray training step is bound by the weight stream: 128 workgroups (one 32-ray tile each) pull all 25.1 MB of packed
// (hi, mid) weights per chain from L2 at ~45 B/clk per CU — 3.0 us per tile and layer over 1.56 us of MFMA, 128 of 256 CUs idle.
// The alternative probed here keeps the WEIGHTS still and moves the activations: a CU pair owns ONE 256 x 256 layer (each CU 128
// output columns, (hi, mid) fp16 = 128 KiB — held in the REGISTER FILE, 128 VGPRs per lane with one wave per SIMD: LDS cannot hold
// it beside the tile buffers, and re-reading A operands from LDS for every tile would make LDS bandwidth the bound), the 32-ray
// tiles flow through the layers: per tile and stage a CU waits for its two producers, pulls the tile's B operand (16 stage
// pieces x (hi, mid) x 1 KiB = 32 KiB) from L2 into LDS, runs 48 MFMAs per wave (3 fp16 products x 16 k-blocks on its 32-column
// tile), splits its 32 x 32 outputs into (hi, mid) and publishes them as 4 KiB of the NEXT layer's stage pieces, then raises a
// flag.  Hand-over inside ONE XCD through its L2, with the idiom tools/cu_exchange_probe.hip measured (plain stores + vmcnt(0),
// sc1 loads: no agent-scope fences).  Stages are placed by HW_REG_XCC_ID read in the kernel: every XCD runs its own 10-layer
// pipeline (20 CUs of its 32), all eight at once, so the chip is loaded as the real thing would load it.
//
// Measured: tiles per second of a pipeline = stage time; and the same with the hand-over removed (every CU on private data:
// what MFMA + LDS alone take), and with the MFMAs removed (what the hand-over alone takes).  A serial run (one launch per
// stage, same code) is the bit-for-bit reference of the pipelined output: a stale read would show.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/layer_pipeline_probe tools/layer_pipeline_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MAX_XCD 8
#define TILE_BYTES 32768        // B operand of one 32-ray tile: 16 k-blocks x (hi, mid) x 64 lanes x 16 B
#define SPIN_MAX 3000000        // bounded waits: a lost flag ends as an error count, never as a hung GPU

struct Args {
    const u32x4* weights;  // [stage][cu][wave][kb 16][split 2][lane 64] 16 B: this wave's A operands of its 32-column tile
    u32x4* bufs;           // [xcd][stage + 1][ring R][TILE_BYTES / 16]: stage s reads bufs[s], writes bufs[s + 1]
    unsigned* flags;       // [xcd][stage + 1][4][32]: produced[0], produced[1], consumed[0], consumed[1] (one 128-B line each)
    unsigned* xcd_count;   // [MAX_XCD] role tickets
    unsigned long long* clocks;  // [xcd][stage][cu][2] wall clock at start / end
    unsigned* err;         // [0] timeouts, [1] xcds with too few workgroups
    int stages, tiles, ring;
    int comm;     // 1: real hand-over; 0: private data (no waits, input re-read from the CU's own slot)
    int mfma;     // 1: real MFMAs; 0: skipped
    int serial_stage;  // >= 0: this launch runs only that stage (flags already satisfied): the serial reference
    int first_xcd_only;
};

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void ld4_sc1_issue(u32x4& v, const u32x4* p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
}
__device__ __forceinline__ void st_flag(unsigned* p, unsigned v) {
    *(volatile unsigned*)p = v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ bool wait_ge(const unsigned* f0, const unsigned* f1, unsigned want, unsigned* err) {
    int spin = 0;
    while (true) {
        const unsigned a = ld_sc1(f0), b = ld_sc1(f1);
        if (a >= want && b >= want) return true;
        if (++spin > SPIN_MAX || ld_sc1(err) != 0u) { atomicAdd(err, 1u); return false; }
    }
}

template <bool MFMA>
__global__ __launch_bounds__(256, 1) void pipeline_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 2 x 32 KiB tile buffers (+ padding: one workgroup per CU)
    __shared__ int role_s;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xf;
        const unsigned ticket = atomicAdd(a.xcd_count + xcc, 1u);
        role_s = (ticket < (unsigned)(2 * a.stages) && (!a.first_xcd_only || xcc == 0)) ? (int)(xcc * 64 + ticket) : -1;
    }
    __syncthreads();
    const int role = role_s;
    if (role < 0) return;
    const int xcd = role >> 6, stage = (role & 63) >> 1, cu = role & 1;
    if (a.serial_stage >= 0 && stage != a.serial_stage) return;
    const size_t tile_units = TILE_BYTES / 16;
    // (no hand-over: every CU reads the XCD's stage-0 input ring — random data, as the real operands are; power follows the data)
    u32x4* inb = a.bufs + ((size_t)(xcd * (a.stages + 1) + (a.comm ? stage : 0)) * a.ring) * tile_units;
    u32x4* outb = a.bufs + ((size_t)(xcd * (a.stages + 1) + stage + 1) * a.ring) * tile_units;
    unsigned* fl_in = a.flags + (size_t)(xcd * (a.stages + 1) + stage) * 128;       // produced by stage - 1, consumed by us
    unsigned* fl_out = a.flags + (size_t)(xcd * (a.stages + 1) + stage + 1) * 128;  // produced by us, consumed by stage + 1
    const bool wait_in = a.comm && a.serial_stage < 0 && stage > 0;
    const bool wait_out = a.comm && a.serial_stage < 0 && stage + 1 < a.stages;

    // A operands: this wave's 32 output columns x 256 inputs x (hi, mid), resident in registers for the whole launch
    f16x8 Ah[16], Am[16];
    {
        const u32x4* w = a.weights + (((size_t)(stage * 2 + cu) * 4 + wave) * 32) * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            Ah[kb] = __builtin_bit_cast(f16x8, w[(kb * 2 + 0) * 64]);
            Am[kb] = __builtin_bit_cast(f16x8, w[(kb * 2 + 1) * 64]);
        }
    }
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    u32x4 pre[8];
    // Tile `tile` uses ring slot tile % ring of every stage, and the input ring holds `ring` distinct tiles: what a slot ends up
    // holding depends on tile % ring only — the serial reference (whole stages one after the other) and the pipeline agree.
    // (A wait that times out does not leave the loop — the barriers below stay uniform —; every later wait returns at once.)
    bool have = false;
    for (int tile = 0; tile < a.tiles; ++tile) {
        const u32x4* src = inb + (size_t)(tile % a.ring) * tile_units;
        if (!have) {
            if (wait_in) (void)wait_ge(fl_in + 0, fl_in + 32, (unsigned)tile + 1u, a.err);
#pragma unroll
            for (int j = 0; j < 8; ++j) ld4_sc1_issue(pre[j], src + j * 256 + t);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        u32x4* lb = reinterpret_cast<u32x4*>(lds + (tile & 1) * TILE_BYTES);
#pragma unroll
        for (int j = 0; j < 8; ++j) lb[j * 256 + t] = pre[j];
        __syncthreads();  // tile complete in LDS; every lane's global loads of it have landed
        if (t == 0 && wait_in) st_flag(fl_in + 64 + 32 * cu, (unsigned)tile + 1u);  // consumed: the ring slot may be overwritten
        // non-blocking look at the next tile: if both producers are done, its 32 KiB are in flight under this tile's MFMAs
        have = false;
        if (tile + 1 < a.tiles) {
            bool ready = true;
            if (wait_in) ready = ld_sc1(fl_in + 0) >= (unsigned)tile + 2u && ld_sc1(fl_in + 32) >= (unsigned)tile + 2u;
            if (ready) {
                const u32x4* nsrc = inb + (size_t)((tile + 1) % a.ring) * tile_units;
#pragma unroll
                for (int j = 0; j < 8; ++j) ld4_sc1_issue(pre[j], nsrc + j * 256 + t);
                have = true;
            }
        }
        // the wave's 32 x 32 outputs: 16 k-blocks x 3 fp16 products (mid*hi + hi*mid + hi*hi), B operands from LDS
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const u32x4* bl = reinterpret_cast<const u32x4*>(lds + (tile & 1) * TILE_BYTES) + lane;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const f16x8 bh = __builtin_bit_cast(f16x8, bl[(kb * 2 + 0) * 64]);
            const f16x8 bm = __builtin_bit_cast(f16x8, bl[(kb * 2 + 1) * 64]);
            if (MFMA) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Am[kb], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[kb], bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[kb], bh, acc, 0, 0, 0);
            } else {
                acc[kb] += (float)bh[0] + (float)bm[1] + (float)Ah[kb][0] + (float)Am[kb][1];
            }
        }
        // epilogue: ReLU, (hi, mid) split; fragment registers 8r .. 8r+7 are, lane for lane, the B operand of k-block 2T + r of the
        // next layer (T = this wave's column tile 4 cu + wave): 2 k-blocks x (hi, mid) = 4 x 16 B per lane
        u32x4 o[4];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            f16x8 hi, mid;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = fmaxf(acc[8 * r + e], 0.f);
                hi[e] = (_Float16)x;
                mid[e] = (_Float16)(x - (float)hi[e]);
            }
            o[2 * r] = __builtin_bit_cast(u32x4, hi);
            o[2 * r + 1] = __builtin_bit_cast(u32x4, mid);
        }
        if (wait_out && tile >= a.ring) (void)wait_ge(fl_out + 64, fl_out + 96, (unsigned)(tile - a.ring) + 1u, a.err);
        u32x4* dst = outb + (size_t)(tile % a.ring) * tile_units + lane;
        const int T = 4 * cu + wave;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            dst[((2 * T + r) * 2 + 0) * 64] = o[2 * r];
            dst[((2 * T + r) * 2 + 1) * 64] = o[2 * r + 1];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this lane's stores are in L2 (and the look-ahead loads have landed)
        __syncthreads();
        if (t == 0 && a.comm) st_flag(fl_out + 32 * cu, (unsigned)tile + 1u);  // produced
    }
    const unsigned long long t1 = wall_clock64();
    if (t == 0) {
        unsigned long long* c = a.clocks + ((size_t)(xcd * a.stages + stage) * 2 + cu) * 2;
        c[0] = t0; c[1] = t1;
    }
}

static unsigned short f16_bits(float v) {
    _Float16 h = (_Float16)v;
    unsigned short b;
    memcpy(&b, &h, 2);
    return b;
}

int main(int argc, char** argv) {
    const int stages = argc > 1 ? atoi(argv[1]) : 10;
    const int tiles = argc > 2 ? atoi(argv[2]) : 128;
    const int ring = argc > 3 ? atoi(argv[3]) : 4;
    int n_cu = 256, wc_khz = 100000;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) n_cu = prop.multiProcessorCount;
    (void)hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
    if (stages < 1 || 2 * stages > 32 || tiles < 1 || ring < 2) { printf("usage: probe [stages <= 16] [tiles] [ring >= 2]\n"); return 2; }
    printf("layer_pipeline_probe: %d CUs, %d stages (layers) x 2 CUs per XCD, %d tiles of 32 rays per pipeline, ring of %d tiles, wall clock %d kHz\n",
           n_cu, stages, tiles, ring, wc_khz);
    // weights: He-scaled uniform values as (hi, mid) fp16 pairs in A-operand order; input tiles: |x| <= 1 as (hi, mid)
    const size_t w_units = (size_t)stages * 2 * 4 * 32 * 64;
    std::vector<unsigned short> hw(w_units * 8);
    unsigned rng = 12345u;
    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return (float)(rng >> 8) * (1.0f / 16777216.0f); };
    for (size_t u = 0; u < w_units; u += 2 * 64)  // (kb, split) pairs: 64 lanes hi then 64 lanes mid of the same values
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e) {
                const float w = (rnd() * 2.f - 1.f) * 0.153f;
                _Float16 hi = (_Float16)w;
                hw[(u + l) * 8 + e] = f16_bits(w);
                hw[(u + 64 + l) * 8 + e] = f16_bits(w - (float)hi);
            }
    const size_t tile_units = TILE_BYTES / 16;
    const size_t buf_units = (size_t)MAX_XCD * (stages + 1) * ring * tile_units;
    std::vector<unsigned short> hin((size_t)ring * tile_units * 8);
    for (size_t u = 0; u < (size_t)ring * tile_units; u += 2 * 64)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e) {
                const float x = rnd();
                _Float16 hi = (_Float16)x;
                hin[(u + l) * 8 + e] = f16_bits(x);
                hin[(u + 64 + l) * 8 + e] = f16_bits(x - (float)hi);
            }
    Args a{};
    u32x4* d_w; u32x4* d_b; unsigned* d_f; unsigned* d_cnt; unsigned long long* d_clk; unsigned* d_err;
    hipMalloc(&d_w, w_units * 16);
    hipMalloc(&d_b, buf_units * 16);
    hipMalloc(&d_f, (size_t)MAX_XCD * (stages + 1) * 128 * 4);
    hipMalloc(&d_cnt, MAX_XCD * 4);
    hipMalloc(&d_clk, (size_t)MAX_XCD * stages * 2 * 2 * 8);
    hipMalloc(&d_err, 8);
    hipMemcpy(d_w, hw.data(), w_units * 16, hipMemcpyHostToDevice);
    a.weights = d_w; a.bufs = d_b; a.flags = d_f; a.xcd_count = d_cnt; a.clocks = d_clk; a.err = d_err;
    a.stages = stages; a.tiles = tiles; a.ring = ring;
    const size_t out_off = (size_t)stages * ring * tile_units;  // units from an XCD's buffer base to its last stage's output ring
    auto reset = [&]() {
        hipMemset(d_b, 0, buf_units * 16);
        for (int x = 0; x < MAX_XCD; ++x)  // stage 0 of every XCD reads the same input ring
            hipMemcpy(d_b + (size_t)x * (stages + 1) * ring * tile_units, hin.data(), (size_t)ring * tile_units * 16, hipMemcpyHostToDevice);
        hipMemset(d_f, 0, (size_t)MAX_XCD * (stages + 1) * 128 * 4);
        hipMemset(d_cnt, 0, MAX_XCD * 4);
        hipMemset(d_clk, 0, (size_t)MAX_XCD * stages * 2 * 2 * 8);
        hipMemset(d_err, 0, 8);
    };
    auto launch = [&](int comm, int mfma, int serial_stage, int first_only) {
        a.comm = comm; a.mfma = mfma; a.serial_stage = serial_stage; a.first_xcd_only = first_only;
        hipMemset(d_cnt, 0, MAX_XCD * 4);
        if (mfma) hipLaunchKernelGGL(pipeline_kernel<true>, dim3(n_cu), dim3(256), 96 * 1024, 0, a);
        else hipLaunchKernelGGL(pipeline_kernel<false>, dim3(n_cu), dim3(256), 96 * 1024, 0, a);
        return hipDeviceSynchronize() == hipSuccess;
    };
    std::vector<unsigned long long> clk((size_t)MAX_XCD * stages * 4);
    auto report = [&](const char* what, bool pipelined) {
        hipMemcpy(clk.data(), d_clk, clk.size() * 8, hipMemcpyDeviceToHost);
        unsigned err[2] = {0, 0};
        hipMemcpy(err, d_err, 8, hipMemcpyDeviceToHost);
        std::vector<unsigned> cnt(MAX_XCD);
        hipMemcpy(cnt.data(), d_cnt, MAX_XCD * 4, hipMemcpyDeviceToHost);
        double worst = 0, sum = 0;
        int n = 0;
        for (int x = 0; x < MAX_XCD; ++x) {
            unsigned long long lo = ~0ull, hi = 0;
            double per_stage = 0;
            int live = 0;
            for (int s = 0; s < stages; ++s)
                for (int c = 0; c < 2; ++c) {
                    const unsigned long long* q = &clk[((size_t)(x * stages + s) * 2 + c) * 2];
                    if (q[1] == 0) continue;
                    ++live;
                    if (q[0] < lo) lo = q[0];
                    if (q[1] > hi) hi = q[1];
                    per_stage += (double)(q[1] - q[0]);
                }
            if (!live) continue;
            // pipelined: first start -> last end covers tiles + stages - 1 stage times; otherwise: a CU's own loop time / tiles
            const double us = pipelined ? (double)(hi - lo) / (wc_khz * 1e-3) / (tiles + stages - 1)
                                        : per_stage / live / (wc_khz * 1e-3) / tiles;
            if (us > worst) worst = us;
            sum += us; ++n;
        }
        printf("  %-58s %.3f us per tile and stage (mean over %d XCD pipelines), %.3f (slowest); timeouts %u; workgroups per XCD:", what,
               n ? sum / n : 0.0, n, worst, err[0]);
        for (int x = 0; x < MAX_XCD; ++x) printf(" %u", cnt[x]);
        printf("\n");
        return err[0] == 0;
    };
    // 1. compute alone: private data, no waits (MFMA + LDS + the L2 reads / writes of the CU's own slot)
    reset(); launch(0, 1, -1, 0); launch(0, 1, -1, 0); report("MFMA + LDS, no hand-over (private data)", false);
    // 2. hand-over alone: the pipeline without MFMAs
    reset(); launch(1, 0, -1, 0); report("hand-over only (MFMAs skipped), 8 XCD pipelines at once", true);
    // 3. the pipeline
    std::vector<unsigned> ref((size_t)ring * tile_units * 4), got((size_t)ring * tile_units * 4);
    reset();
    for (int s = 0; s < stages; ++s) launch(1, 1, s, 1);  // serial reference on XCD 0 (one launch per stage)
    hipMemcpy(ref.data(), d_b + out_off, ref.size() * 4, hipMemcpyDeviceToHost);
    for (int rep = 0; rep < 3; ++rep) {
        reset();
        launch(1, 1, -1, 0);
        const bool ok = report(rep == 0 ? "PIPELINE: MFMA + hand-over, 8 XCD pipelines at once" : "  (again)", true);
        int bad = 0;
        for (int x = 0; x < MAX_XCD; ++x) {
            hipMemcpy(got.data(), d_b + (size_t)x * (stages + 1) * ring * tile_units + out_off, got.size() * 4, hipMemcpyDeviceToHost);
            if (memcmp(got.data(), ref.data(), got.size() * 4) != 0) ++bad;
        }
        printf("    last stage's output ring of the 8 pipelines vs the serial reference: %d differ (must be 0)%s\n", bad, ok ? "" : "  [timeouts]");
    }
    reset(); launch(1, 1, -1, 1); report("PIPELINE on ONE XCD only (the other 7 idle)", true);
    // what it would mean for a chain: 87 layers, 128 tiles (4096 rays)
    printf("  (a chain of 87 stages over 128 tiles = 214 stage times; today's cooperative chain: ~0.27 ms = 3.0 us x 87 layers per tile)\n");
    return 0;
}
