// r2l_fwd2.hip — the R2L student forward on the fp16 matrix pipe with two-way operand splits (see r2l_f2.h): every fp32
// product as three fp16 MFMA products, ~2^-21 relative.  Structure, stage order and gatherers are r2l_fwd3.hip's; stages are
// 16 KiB ([split 2][tile 8][lane 64][8 fp16]).  Default of every one-wave-per-tile forward launch (render / evaluation and the training forward with its stash);
// R2L_NO_FWD2=1: bf16x3 only.
#include "r2l_f2.h"
#include "r2l_coopf.h"
#ifndef F2_PARK_X0
// X_0 tiles parked in LDS across the body loop.  0 (default): all of X_0 left to the register allocator, which spills 76 of its
// 128 registers to scratch once per tile — 0.85 GB written and read back per 9-frame launch, none of it time-limiting.  3 (what
// the LDS left by the six stage buffers takes): 29 spills, WRITE_SIZE 849 -> 345 MB per launch (rocprofv3 --pmc), but the same-box
// A/B says 38.18 / 38.08 ms per launch against 38.05 / 37.99 with 0 (profiles/r04_render_x0_park_ab.txt): -0.3 %, so off.
#define F2_PARK_X0 0
#endif

// =================================================================================================================
// pack: flat fp32 parameters -> stage stream for the activation scale the status words ask for (r2l_f2.h: range control).
// The pack kernel only READS the status words (every workgroup derives the same scale from them); the one-thread commit
// kernel behind it writes the scale, closes the amax epoch and clears the guard — every pack: new weights, new chance.
// =================================================================================================================
__global__ void r2l_pack_fwd2_kernel(const float* __restrict__ params, unsigned short* __restrict__ out, int n_block,
                                     const unsigned* __restrict__ status) {
    const F2Next nx = f2_next_scale(status);
    f2_pack_fwd_elements(params, out, n_block, nx.inv, false, (int64_t)blockIdx.x * blockDim.x + threadIdx.x,
                         (int64_t)gridDim.x * blockDim.x);
}
__global__ void r2l_fwd2_commit_kernel(unsigned* status) {
    if (threadIdx.x == 0) f2_commit_scale(status, f2_next_scale(status), false);
}

// =================================================================================================================
// forward
// =================================================================================================================
struct F2Args {
    const float* rays_o;
    const float* rays_d;
    const float* t_rand;
    const float* ztab;
    float c2w[12];
    const float* c2w_dev;  // several frames per launch: [K][12] on the device (r2l_common.h r2l_pose_of), else nullptr
    int H, Wimg;
    float focal;
    const unsigned char* stream;  // fwd2 stage stream
    unsigned* status;             // range-guard word behind the stream: != 0 -> this launch is left to the bf16x3 kernel
    const float* params;
    int n_block;
    float* rgb;
    float* save_x;  // training: the stash slots of X_0 .. X_{n-1} / relu(hidden) as fp16 stage pieces (r2l_f2.h), slot n of
    float* save_t;  //           save_x = X_n + X_0 row-major fp32; or nullptr
    int64_t N;
    unsigned stash_mid;  // != 0: also stash the operands' mid halves, this many bytes behind the hi pieces (r2l_f2.h)
};

// MID (training stash only): the operands' mid halves are stashed too (exact weight gradients, r2l_f2.h); a template
// parameter because even a uniform branch around the store inside a stage costs the default kernel ~150 more spilled VGPRs
template <bool POSE, bool SAVE, bool MID = false>
__global__ __launch_bounds__(256, 1) void r2l_fwd2_kernel(const F2Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[F2_NBUF][F2_STAGE_BYTES];
#if F2_PARK_X0 > 0
    // X_0 (outer residual: needed again only at the tail) does not fit beside x, t and the A operands in 512 registers: hipcc
    // spilled ~76 of its 128 registers per lane to scratch (0.85 GB written and read back per 9-frame launch).
    // The LDS the six stage buffers leave free (56 KiB) takes F2_PARK_X0 = 3 of its 8 tiles: [wave][quad][lane][16 B],
    // conflict-free 16-byte accesses; the rest stays where the register allocator puts it.
    __shared__ __attribute__((aligned(16))) f32x4 x0park[4][F2_PARK_X0 * 4][64];
#endif

    // Tail weights (Linear(256, 3): 3 KiB) and biases, staged in LDS once per workgroup (published by the prologue's barrier).
    // Read from global at the end of a tile — 96 16-byte loads per lane, and with no registers to spare hipcc issued them three
    // at a time behind `s_waitcnt vmcnt(0)`: 32 serialized L2 round trips per tile, each of which ALSO waits for the twenty weight
    // DMA loads the staging pipeline has in flight (round 4: ISA of the tail; same-box A/B profiles/r04_tail_lds_ab.txt).
    __shared__ __attribute__((aligned(16))) float tail_w[3 * R2L_W + 4];

    // an earlier launch with these weights left fp16's range: the bf16x3 kernel behind this one does the work
    if (__builtin_nontemporal_load(a.status) != 0u) return;
    for (int i = threadIdx.x; i < 3 * R2L_W + 3; i += 256) tail_w[i] = a.params[f2_off_tail_w(a.n_block) + i];  // (W then b: contiguous)
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // every wave of the workgroup takes part in the weight staging and the barriers: a wave whose tile lies past the end
    // recomputes the last live tile (identical values to identical addresses), so nothing in the chain is conditional
    const int64_t n_tiles = (a.N + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile > n_tiles - 1) tile = n_tiles - 1;
    const int64_t ray = tile * R2L_TILE_RAYS + (lane & 31);
    const bool valid = ray < a.N;
    const int64_t rc = valid ? ray : a.N - 1;

    // ---- rays -----------------------------------------------------------------------------------------------------------
    float o[3], d[3];
    if constexpr (!POSE) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = a.rays_o[rc * 3 + k];
            d[k] = a.rays_d[rc * 3 + k];
        }
    } else {
        const R2LPoseRay pr = r2l_pose_of(a.c2w, a.c2w_dev, (int64_t)a.H * a.Wimg, rc);
        const int pj = (int)(pr.pix / a.Wimg), pi = (int)(pr.pix % a.Wimg);
        const float dx = ((float)pi - (float)a.Wimg * 0.5f) / a.focal;
        const float dy = -(((float)pj - (float)a.H * 0.5f) / a.focal);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d[k] = (dx * pr.c[4 * k + 0] + dy * pr.c[4 * k + 1]) + (-1.0f) * pr.c[4 * k + 2];
            o[k] = pr.c[4 * k + 3];
        }
    }
    float z[8];  // the 8 sample depths of this half-wave (samples 8h .. 8h+7)
    {
        const f32x4 lo0 = *reinterpret_cast<const f32x4*>(a.ztab + 8 * h);
        const f32x4 lo1 = *reinterpret_cast<const f32x4*>(a.ztab + 8 * h + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { z[k] = lo0[k]; z[4 + k] = lo1[k]; }
        if (a.t_rand != nullptr) {
            const f32x4 sp0 = *reinterpret_cast<const f32x4*>(a.ztab + 16 + 8 * h);
            const f32x4 sp1 = *reinterpret_cast<const f32x4*>(a.ztab + 16 + 8 * h + 4);
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(a.t_rand + rc * 16 + 8 * h);
            const f32x4 u1 = *reinterpret_cast<const f32x4*>(a.t_rand + rc * 16 + 8 * h + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { z[k] = lo0[k] + sp0[k] * u0[k]; z[4 + k] = lo1[k] + sp1[k] * u1[k]; }
        }
    }

    // ---- weight staging: stage g of the stream is DMA'd into LDS buffer g % 6 by the four waves (a quarter each) ----------
    F2Pipe P;
    {
        const unsigned long long sa = (unsigned long long)a.stream;
        P.rs = u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sa),
                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sa >> 32)) & 0xffffu, 0xffffffffu, 0x00020000u};
        P.lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)&wbuf[0][0];
        P.voff = (unsigned)lane * 16u;
        P.wq = (unsigned)wave * 4096u;
        P.base = &wbuf[0][0];
        P.lane = lane;
        P.gb = 0;
        P.gq = 0;
        P.gqb = 0;
        P.amax = 0.f;
    }
#pragma unroll
    for (int k = 0; k < F2_NBUF - 1; ++k) P.issue();  // stages 0 .. F2_NBUF-2
#pragma unroll
    for (int k = 0; k < 8; ++k) P.ones.h[k] = (_Float16)((h == 0 && k < 2) ? 1.0f : 0.0f);
    P.ones.m = P.ones.h;

    f32x16 x[R2L_NT], t[R2L_NT], x0[R2L_NT];
    // prologue: stage 0 (head bias) becomes current
    if (F2_NBUF == 6) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // stage 0 landed
    else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __syncthreads();
    P.lb = P.base + lane * 16;
    {
        F3None none;
        F2Side<true, F3None> s0{P.a1, P.lb, 0, none, false, F3Dma{false, P.rs, 0u, 0u, 0u}, P.amax};
#pragma unroll
        for (int i = 0; i < 6; ++i) s0.step(i);
    }
    P.sb = P.ones;

    // ---- head ---------------------------------------------------------------------------------------------------------
    // per coordinate pair (xa, xb) five k-blocks: [xa f0-3] [xa f4-7] [xa f8,9 | xb f0,1] [xb f2-5] [xb f6-9]
    auto zsel = [&](int s) {
        float zz = z[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) zz = (s == k) ? z[k] : zz;
        return zz;
    };
    float xc[6];
    {
        const float za = z[0], zb = z[1];
#pragma unroll
        for (int ci = 0; ci < 6; ++ci) xc[ci] = o[ci % 3] + d[ci % 3] * (ci / 3 == 0 ? za : zb);
    }
    // Encoding values one stage ahead (r2l_f2.h F2TrigPre): a stage's gather copies four finished registers per half (tlo / thi)
    // while side steps 2 - 5 of the same half evaluate the NEXT stage's, a phase per step.  The first stage's are evaluated here.
    float tlo[4], thi[4], slo[5], shi[5];
    F3Trig2{xc[0], 0}(tlo);
    F3Trig2{xc[0], 2}(thi);
    struct RegsOrIdent {  // last stage of a head trip: the next trip's first block (registers), or the first identity block
        bool ident;
        F2TakeRegs tr;
        F3Ident4 id;
        __device__ __forceinline__ void operator()(float (&v)[4]) const {
            float w[4];
            tr(v);
            id(w);
#pragma unroll
            for (int s = 0; s < 4; ++s) v[s] = ident ? w[s] : v[s];
        }
    };
    const F2TakeRegs glo{tlo}, ghi{thi};
    f2_stage_pre<true, true>(x, P, glo, ghi, F2TrigPre{xc[0], 4, tlo, slo}, F2TrigPre{xc[0], 6, thi, shi});
#pragma unroll 1
    for (int it2 = 0; it2 < 4; ++it2) {  // two samples = six coordinates = three pairs = 15 k-blocks per trip
        // coordinates of the NEXT trip (the last stages of this trip prepare the first blocks of the next one)
        float xn[6];
        {
            const float za = zsel(2 * it2 + 2), zb = zsel(2 * it2 + 3);
#pragma unroll
            for (int ci = 0; ci < 6; ++ci) xn[ci] = o[ci % 3] + d[ci % 3] * (ci / 3 == 0 ? za : zb);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            // gathers of this pair's five stages: [xa f4-7] [xa f8,9 | xb f0,1] [xb f2-5] [xb f6-9] [next coordinate f0-3], each
            // evaluated by the stage in front of it
            const float xa = xc[2 * p], xb = xc[2 * p + 1], xnext = p < 2 ? xc[2 * p + 2] : xn[0];
            f2_stage_pre<false, false>(x, P, glo, ghi, F2TrigPre{xa, 8, tlo, slo}, F2TrigPre{xb, 0, thi, shi});
            f2_stage_pre<false, false>(x, P, glo, ghi, F2TrigPre{xb, 2, tlo, slo}, F2TrigPre{xb, 4, thi, shi});
            f2_stage_pre<false, false>(x, P, glo, ghi, F2TrigPre{xb, 6, tlo, slo}, F2TrigPre{xb, 8, thi, shi});
            f2_stage_pre<false, false>(x, P, glo, ghi, F2TrigPre{xnext, 0, tlo, slo}, F2TrigPre{xnext, 2, thi, shi});
            if (p < 2) {
                f2_stage_pre<false, false>(x, P, glo, ghi, F2TrigPre{xnext, 4, tlo, slo}, F2TrigPre{xnext, 6, thi, shi});
            } else {
                f2_stage_pre<false, false>(x, P, RegsOrIdent{it2 == 3, glo, F3Ident4{o, d, z, 0}},
                                           RegsOrIdent{it2 == 3, ghi, F3Ident4{o, d, z, 4}}, F2TrigPre{xnext, 4, tlo, slo},
                                           F2TrigPre{xnext, 6, thi, shi});
            }
        }
#pragma unroll
        for (int ci = 0; ci < 6; ++ci) xc[ci] = xn[ci];
    }
    // identity features: coordinates 8j .. 8j+7 of the half
    f2_stage<false, false, false>(x, P, F3Ident4{o, d, z, 8}, F3Ident4{o, d, z, 12});
    f2_stage<false, false, false>(x, P, F3Ident4{o, d, z, 16}, F3Ident4{o, d, z, 20});
    f2_stage<false, false, true>(x, P, F3None{}, F3None{});
    // the head ran on unscaled weights (r2l_f2.h range control): the chain continues on X_0 / act_s — one fp32 multiply by a
    // power of two per value (1.0 for every net below the guard: bit-identical to the unscaled chain)
    const float act_inv = f2_act_inv(a.status);
#pragma unroll
    for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            x[T][c] = fmaxf(x[T][c], 0.f) * act_inv;  // X_0 = relu(head) / act_s
            x0[T][c] = x[T][c];
        }
#if F2_PARK_X0 > 0
#pragma unroll
    for (int T = R2L_NT - F2_PARK_X0; T < R2L_NT; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            x0park[wave][(T - (R2L_NT - F2_PARK_X0)) * 4 + q][lane] = f32x4{x0[T][4 * q], x0[T][4 * q + 1], x0[T][4 * q + 2], x0[T][4 * q + 3]};
#endif

    // ---- body -----------------------------------------------------------------------------------------------------------
    // training (SAVE): the B values of every stage are the layer's input (x_b for the first layer of a block, relu(t_b) for the
    // second): their fp16 hi halves — the assembled B operand of the next stage — are the stash, ONE 16-byte store per lane and
    // stage, a contiguous KiB per wave (r2l_f2.h F2Hst; r2l_dw16.hip reads it back)
    const int64_t Np = R2L_PAD_ROWS(a.N);
    // (rows of the padding rays of the last tile exist: Np rows per slot)
    const int64_t slot = R2L_TRIO_SLOT(Np);
    const unsigned hvoff = (unsigned)((tile * R2L_H16_TILE_UNITS + lane) * 16);  // this lane's unit of stage piece 0 in a slot
    const F3Dma no_dma{false, u32x4{0u, 0u, 0u, 0u}, 0u, 0u, 0u};
    constexpr bool mid = SAVE && MID;
    // the activation scale this stream was packed for (r2l_f2.h range control): what the chain holds is x / act_s
    const float act_s = f2_act_scale(a.status);
    if (SAVE && blockIdx.x == 0 && threadIdx.x == 0) {  // stash format word: fp16 stage pieces (a fallback launch overwrites it),
        reinterpret_cast<unsigned*>(a.save_x)[R2L_STASH_FMT_WORD(a.n_block, Np)] = 0u;  // and the scale of the stashed x, relu(t)
        reinterpret_cast<float*>(a.save_x)[R2L_STASH_FMT_WORD(a.n_block, Np) + 1] = act_s;
    }
#pragma unroll 1
    for (int b = 0; b < a.n_block; ++b) {
        float* const hxr = SAVE ? a.save_x + (int64_t)b * slot : nullptr;  // this block's slots
        float* const htr = SAVE ? a.save_t + (int64_t)b * slot : nullptr;
        // t = W1 x + b1   (its ReLU is applied where t is consumed)
        f2_stage<true, true, false>(t, P, F3Take4<false, SAVE, false>{x[0], 0, nullptr, 0}, F3Take4<false, SAVE, false>{x[0], 4, nullptr, 0},
                                    no_dma, no_dma, F2Hst{SAVE, hxr, hvoff, 0u});
#pragma unroll
        for (int kb = 0; kb < 15; ++kb)
            f2_stage<false, false, false>(t, P, F3Take4<false, SAVE, false>{x[(kb + 1) >> 1], 8 * ((kb + 1) & 1), nullptr, (kb + 1) >> 1},
                                          F3Take4<false, SAVE, false>{x[(kb + 1) >> 1], 8 * ((kb + 1) & 1) + 4, nullptr, (kb + 1) >> 1},
                                          no_dma, no_dma, F2Hst{SAVE, hxr, hvoff, 1024u * (unsigned)(kb + 1)},
                                          F2Hst{mid, hxr, hvoff, 1024u * (unsigned)kb + a.stash_mid});
        f2_stage<false, false, true>(t, P, F3None{}, F3None{}, no_dma, no_dma, F2Hst{false, nullptr, 0u, 0u},
                                     F2Hst{mid, hxr, hvoff, 1024u * 15u + a.stash_mid});
        // x += W2 relu(t) + b2   (training: the gatherers also shift [t > 0] into the block's four mask words)
        unsigned mw[4] = {0u, 0u, 0u, 0u};
        f2_stage<true, false, false>(x, P, F3Take4<true, SAVE, false>{t[0], 0, nullptr, 0, &mw[0]},
                                     F3Take4<true, SAVE, false>{t[0], 4, nullptr, 0, &mw[0]}, no_dma, no_dma, F2Hst{SAVE, htr, hvoff, 0u});
#pragma unroll
        for (int kb = 0; kb < 15; ++kb)
            f2_stage<false, false, false>(x, P, F3Take4<true, SAVE, false>{t[(kb + 1) >> 1], 8 * ((kb + 1) & 1), nullptr, (kb + 1) >> 1, &mw[(kb + 1) >> 2]},
                                          F3Take4<true, SAVE, false>{t[(kb + 1) >> 1], 8 * ((kb + 1) & 1) + 4, nullptr, (kb + 1) >> 1, &mw[(kb + 1) >> 2]},
                                          no_dma, no_dma, F2Hst{SAVE, htr, hvoff, 1024u * (unsigned)(kb + 1)},
                                          F2Hst{mid, htr, hvoff, 1024u * (unsigned)kb + a.stash_mid});
        // next: the next block's bias stage (or the padding)
        f2_stage<false, false, true>(x, P, F3None{}, F3None{}, no_dma, no_dma, F2Hst{false, nullptr, 0u, 0u},
                                     F2Hst{mid, htr, hvoff, 1024u * 15u + a.stash_mid});
        if (SAVE) {  // values were shifted in MSB-first: bit (T&1)*16 + c after the reversal
            u32x4 mv;
#pragma unroll
            for (int w = 0; w < 4; ++w) mv[w] = __builtin_bitreverse32(mw[w]);
            *reinterpret_cast<u32x4*>(a.save_t + (int64_t)b * slot + R2L_MASK_OFFSET(Np) + tile * 256 + lane * 4) = mv;
        }
    }
    // y = x_n + x_0 (outer residual), in place
#pragma unroll
    for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#if F2_PARK_X0 > 0
            if (T >= R2L_NT - F2_PARK_X0) {
                const f32x4 p4 = x0park[wave][(T - (R2L_NT - F2_PARK_X0)) * 4 + q][lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) x[T][4 * q + e] += p4[e];
                continue;
            }
#endif
#pragma unroll
            for (int e = 0; e < 4; ++e) x[T][4 * q + e] += x0[T][4 * q + e];
        }
    if (SAVE) {  // slot n: y = x_n + x_0, row-major (the tail weight gradient reads nothing else)
        float* sy = a.save_x + (int64_t)a.n_block * slot + ray * R2L_W + 4 * h;
#pragma unroll
        for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4*>(sy + 32 * T + 8 * q) =
                    f32x4{x[T][4 * q] * act_s, x[T][4 * q + 1] * act_s, x[T][4 * q + 2] * act_s, x[T][4 * q + 3] * act_s};
    }

    // range control (r2l_f2.h): the wave's largest |B value| -> AMAX; one that close to 65504 may have become inf in a
    // product's operand -> FLAG: the bf16x3 kernel behind this launch redoes it and the stream is re-packed for a larger scale
    f2_report_amax(a.status, P.amax, lane);

    // ---- tail: rgb = sigmoid(Wt (x + X_0) + bt) on the VALU -------------------------------------------------------------
    const float* tw = tail_w + 4 * h;
    float p3[3];
    {
        // packed FMAs on operand pairs that sit in adjacent registers (two weights of a 16-byte LDS read x two neighbouring values:
        // plain v_pk_fma_f32, no op_sel), even / odd partial sums per channel
        typedef float f2_f32x2 __attribute__((ext_vector_type(2)));
        f2_f32x2 a2[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 wv[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) wv[c] = *reinterpret_cast<const f32x4*>(tw + c * R2L_W + 32 * T + 8 * q);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    a2[c] = __builtin_elementwise_fma(f2_f32x2{wv[c][0], wv[c][1]}, f2_f32x2{x[T][4 * q], x[T][4 * q + 1]}, a2[c]);
                    a2[c] = __builtin_elementwise_fma(f2_f32x2{wv[c][2], wv[c][3]}, f2_f32x2{x[T][4 * q + 2], x[T][4 * q + 3]}, a2[c]);
                }
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) p3[c] = a2[c][0] + a2[c][1];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) p3[c] += __shfl_xor(p3[c], 32);
    if (valid && h == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = p3[c] * act_s + tail_w[3 * R2L_W + c];  // (the chain ran on y / act_s: exact)
            a.rgb[ray * 3 + c] = 1.0f / (1.0f + expf(-v));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
int r2l_fwd2_pack(const float* params, int n_block, float* wstream2, hipStream_t stream) {
    unsigned* status = reinterpret_cast<unsigned*>(wstream2 + r2l_fwd2_status_offset(n_block));
    hipLaunchKernelGGL(r2l_pack_fwd2_kernel, dim3(2048), dim3(256), 0, stream, params,
                       reinterpret_cast<unsigned short*>(wstream2), n_block, status);
    R2L_CHECK(hipGetLastError());
    hipLaunchKernelGGL(r2l_fwd2_commit_kernel, dim3(1), dim3(64), 0, stream, status);
    R2L_CHECK(hipGetLastError());
    return 0;
}

int r2l_fwd2_forward(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab,
                     const float* c2w_host12, int H, int W, float focal, const float* wstream2, const float* params,
                     int n_block, float* rgb, float* save_x, float* save_t, int64_t N, hipStream_t stream) {
    // small launches: one 32-ray tile per workgroup instead of per wave (r2l_coopf_fwd.hip), same stream / stash / status word
    if (r2l_use_coopf(N, n_block))
        return r2l_coopf_forward(rays_o, rays_d, t_rand, ztab, c2w_host12, H, W, focal, wstream2, params, n_block, rgb, save_x,
                                 save_t, N, stream);
    F2Args a{};
    a.rays_o = rays_o; a.rays_d = rays_d; a.t_rand = t_rand; a.ztab = ztab;
    a.stream = reinterpret_cast<const unsigned char*>(wstream2); a.params = params;
    // the status word lives in the caller's stream buffer (library-private contents): written through, hence the cast
    a.status = reinterpret_cast<unsigned*>(const_cast<float*>(wstream2) + r2l_fwd2_status_offset(n_block));
    a.n_block = n_block; a.rgb = rgb; a.save_x = save_x; a.save_t = save_t; a.N = N; a.H = H; a.Wimg = W; a.focal = focal;
    if (c2w_host12) for (int i = 0; i < 12; ++i) a.c2w[i] = c2w_host12[i];
    a.c2w_dev = c2w_host12 ? g_r2l_c2w_dev : nullptr;
    a.stash_mid = (save_x != nullptr && r2l_dw_exact()) ? (unsigned)R2L_H16_MID_BYTES(R2L_PAD_ROWS(N)) : 0u;
    const int64_t tiles = (N + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    if (c2w_host12) hipLaunchKernelGGL((r2l_fwd2_kernel<true, false>), grid, block, 0, stream, a);
    else if (save_x && a.stash_mid != 0u) hipLaunchKernelGGL((r2l_fwd2_kernel<false, true, true>), grid, block, 0, stream, a);
    else if (save_x) hipLaunchKernelGGL((r2l_fwd2_kernel<false, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((r2l_fwd2_kernel<false, false>), grid, block, 0, stream, a);
    R2L_CHECK(hipGetLastError());
    return 0;
}
