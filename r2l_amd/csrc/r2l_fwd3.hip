// r2l_fwd3.hip — the R2L student forward at fp32 accuracy on the bf16 matrix pipe (gfx950).
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate.  An fp32 value is the exact sum of three bf16 numbers
// (hi + mid + lo: 8 + 8 + 8 mantissa bits, same exponent range as fp32), so a product a*b of fp32 operands is
// reproduced to ~2^-24 relative by the six bf16 products  hi*hi + hi*mid + mid*hi + hi*lo + mid*mid + lo*hi  (each exact
// in the MFMA's fp32 accumulator; the three dropped terms are below 2^-24).  Six bf16 MFMAs of K = 16 replace sixteen
// fp32 MFMAs of K = 2 at the same result quality: 2.67x fewer matrix-pipe cycles for the same 11.79 MFLOP/ray.
//
// Structure: the register-resident activation chain of r2l_forward.hip (one wave = 32 rays for the whole network, C/D
// fragment of layer n = B operand of layer n+1: with the k slots of a K=16 block numbered so that lane-half h supplies the
// eight features {8q+4h+e}, the eight B values of a block are eight CONSECUTIVE fragment registers) — but
//   * weights are pre-split into bf16 triples by r2l_pack_fwd3_kernel and streamed in 24 KiB stages (one k-block of 16 x
//     256 outputs x 3 splits) straight into LDS with `buffer_load_dwordx4 ... lds` (no registers), four buffers deep
//     (three stages ~1.9 us of latency cover), shared by the four waves of the workgroup: one barrier per stage;
//   * the eight B values of a block are split into (hi, mid, lo) on the VALU in the shadow of the block's 48 MFMAs;
//   * the bias of a layer is ONE MFMA per output tile: hi, mid, lo of the bias sit in three k slots against B = 1;
//   * X_0 stays in registers (the weight ring no longer needs them) for the outer residual.
// Since r2l_fwd2.hip (three fp16 products per fp32 product) became the default this kernel is its range-guard fallback
// (launched behind it, returns at once unless the fp16 kernel raised its status word) and the R2L_NO_FWD2=1 path; with SAVE
// it also writes the training stash (chunked layout + ReLU mask words, r2l_common.h).
#include "r2l_f2.h"  // (r2l_f3.h + the fp16x2 stream's range control: this file holds its fallback pack)

__host__ __device__ static inline int64_t f3_off_head_b() { return (int64_t)R2L_IN * R2L_W; }
__host__ __device__ static inline int64_t f3_off_body_w(int layer) {
    return (int64_t)R2L_IN * R2L_W + R2L_W + (int64_t)layer * (R2L_W * R2L_W + R2L_W);
}
__host__ __device__ static inline int64_t f3_off_body_b(int layer) { return f3_off_body_w(layer) + R2L_W * R2L_W; }
__host__ __device__ static inline int64_t f3_off_tail_w(int n_block) { return f3_off_body_w(2 * n_block); }
__host__ __device__ static inline int64_t f3_off_tail_b(int n_block) { return f3_off_tail_w(n_block) + 3 * R2L_W; }

// =================================================================================================================
// pack: flat fp32 parameters -> stage stream.  Stage g: 0 = head bias, 1..63 = head k-blocks, then per body layer
// [bias stage, 16 k-block stages].  A stage is [split sp][tile t][lane (i,h)][slot s] bf16: split sp of
// W[32t + i][feature(stage, h, s)].
//   head k-block b, half h, slot s:  v = 8b + s indexes the half's 504 encoding values: v < 480: coordinate ci = v/20 of
//     the half (sample 8h + ci/3, axis ci%3), frequency f = (v%20)/2, sin (even) / cos (odd);  v >= 480: identity of
//     coordinate v - 480.
//   body k-block kb = 2T + r: feature 32T + 8(2r + (s>>2)) + 4h + (s&3)  — fragment registers 8r .. 8r+7 of tile T.
//   bias stage: split region 0 only: slots 0,1,2 of half 0 = hi, mid, lo of bias[32t + i]; everything else 0.
// =================================================================================================================
__device__ __forceinline__ void f3_pack_fwd_elements(const float* __restrict__ params, unsigned short* __restrict__ out, int n_block) {
    const int64_t stages = r2l_fwd3_stages(n_block);
    const int64_t total = (stages + R2L_F3_PAD_STAGES) * 8 * 64 * 8;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(idx & 7), lane = (int)((idx >> 3) & 63), tile = (int)((idx >> 9) & 7);
        const int64_t g = idx >> 12;
        const int i = lane & 31, h = lane >> 5, o = 32 * tile + i;
        unsigned short* st = out + g * (F3_STAGE_BYTES / 2);
        unsigned short v0 = 0, v1 = 0, v2 = 0;
        if (g < stages) {
            bool bias_stage = false;
            float w = 0.f;
            if (g == 0) {
                bias_stage = true;
                w = params[f3_off_head_b() + o];
            } else if (g < 64) {
                const int v = 8 * (int)(g - 1) + s;
                int col;
                if (v < 480) {
                    const int ci = v / 20, within = v % 20, f = within >> 1;
                    col = 21 * (3 * (8 * h + ci / 3) + ci % 3) + ((within & 1) ? 10 + f : f);
                } else {
                    const int e = v - 480;
                    col = 21 * (3 * (8 * h + e / 3) + e % 3) + 20;
                }
                w = params[(int64_t)o * R2L_IN + col];
            } else {
                const int layer = (int)((g - 64) / 17), r17 = (int)((g - 64) % 17);
                if (r17 == 0) {
                    bias_stage = true;
                    w = params[f3_off_body_b(layer) + o];
                } else {
                    const int kb = r17 - 1, T = kb >> 1, r = kb & 1;
                    const int in = 32 * T + 8 * (2 * r + (s >> 2)) + 4 * h + (s & 3);
                    w = params[f3_off_body_w(layer) + (int64_t)o * R2L_W + in];
                }
            }
            const unsigned short hi = f3_bf16_rne(w);
            const float r1 = w - f3_bf16_to_f(hi);
            const unsigned short mid = f3_bf16_rne(r1);
            const unsigned short lo = f3_bf16_rne(r1 - f3_bf16_to_f(mid));
            if (bias_stage) {
                v0 = (h == 0) ? (s == 0 ? hi : (s == 1 ? mid : (s == 2 ? lo : (unsigned short)0))) : (unsigned short)0;
            } else {
                v0 = hi; v1 = mid; v2 = lo;
            }
        }
        const int64_t e = ((int64_t)tile * 64 + lane) * 8 + s;
        st[e] = v0;
        st[8 * 64 * 8 + e] = v1;
        st[2 * 8 * 64 * 8 + e] = v2;
    }
}
__global__ void r2l_pack_fwd3_kernel(const float* __restrict__ params, unsigned short* __restrict__ out, int n_block,
                                     const unsigned* __restrict__ run_if) {
    if (run_if != nullptr && __builtin_nontemporal_load(run_if) == 0u) return;
    f3_pack_fwd_elements(params, out, n_block);
}
// The fallback pack behind an fp16x2 forward launch (r2l_f2.h: range control).  FLAG == 0 (the launch stayed in range): GO = 0,
// done.  Else: this (unscaled) bf16x3 stream for the r2l_fwd3 launch behind, the scale-dependent stages of the fp16x2 stream
// for the scale f2_next_scale derives from the (still unchanged) status words — and the LAST workgroup to finish commits that
// scale: FLAG cleared (unless the scale is exhausted), GO = 1.  Every workgroup read FLAG before any could have cleared it.
__global__ void r2l_fwd2_fallback_pack_kernel(const float* __restrict__ params, unsigned short* __restrict__ out3,
                                              unsigned short* __restrict__ out2, int n_block, unsigned* status) {
    bool tripped;
    if (!f2_rescale_due(status, tripped)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) status[F2S_GO] = 0u;
        return;
    }
    const F2Next nx = f2_next_scale(status);
    if (tripped) f3_pack_fwd_elements(params, out3, n_block);  // (else: a refinement of the scale only, GO stays 0)
    f2_pack_fwd_elements(params, out2, n_block, nx.inv, true, (int64_t)blockIdx.x * blockDim.x + threadIdx.x,
                         (int64_t)gridDim.x * blockDim.x);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(status + F2S_DONE, 1u) == gridDim.x - 1u) f2_commit_scale(status, nx, tripped);
}

// =================================================================================================================
// forward
// =================================================================================================================
struct F3Args {
    const float* rays_o;
    const float* rays_d;
    const float* t_rand;
    const float* ztab;
    float c2w[12];
    const float* c2w_dev;  // several frames per launch: [K][12] on the device (r2l_common.h r2l_pose_of), else nullptr
    int H, Wimg;
    float focal;
    const unsigned char* stream;  // fwd3 stage stream
    const unsigned* run_if;       // nullptr, or: return at once while this word is 0 (range-guard fallback of r2l_fwd2.hip)
    const float* params;
    int n_block;
    float* rgb;
    float* save_x;  // training: [(n_block+1)][Np][256] X_0 .. X_n and [n_block][Np][256] relu(hidden) (r2l_forward.hip), or
    float* save_t;  //           nullptr
    int64_t N;
    const float* x0_in;  // X0 instantiation: [Np][256] row-major X_0 = relu(head) from an earlier launch (r2l_forward_emb_cfg)
};

// X0 (round 5, the module-boundary path on the bf16x3 chain): the head is NOT part of this launch — X_0 comes row-major from
// memory (the fp32-MFMA kernel computed it from the caller's [N,1008] encoding, r2l_forward.hip) and `stream` points at the
// FIRST BODY stage (64 stages into the packed stream): that stage is a bias stage like stage 0, so the staging prologue is the same.
template <bool POSE, bool SAVE, bool X0 = false>
__global__ __launch_bounds__(256, 1) void r2l_fwd3_kernel(const F3Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[F3_NBUF][F3_STAGE_BYTES];

    if (a.run_if != nullptr && __builtin_nontemporal_load(a.run_if) == 0u) return;
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
    float o[3], d[3];  // (X0: unused — the head is not part of the launch)
    if constexpr (X0) {
    } else if constexpr (!POSE) {
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
    if constexpr (!X0) {
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
    F3Pipe P;
    {
        const unsigned long long sa = (unsigned long long)a.stream;
        P.rs = u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sa),
                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sa >> 32)) & 0xffffu, 0xffffffffu, 0x00020000u};
        P.lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)&wbuf[0][0];
        P.voff = (unsigned)lane * 16u;
        P.wq = (unsigned)wave * 6144u;
        P.base = &wbuf[0][0];
        P.lane = lane;
        P.gb = 0;
        P.gq = 0;
        P.gqb = 0;
    }
    P.issue(); P.issue(); P.issue(); P.issue(); P.issue();  // stages 0..4
#pragma unroll
    for (int k = 0; k < 8; ++k) P.ones.h[k] = (__bf16)((h == 0 && k < 3) ? 1.0f : 0.0f);
    P.ones.m = P.ones.h;
    P.ones.l = P.ones.h;

    f32x16 x[R2L_NT], t[R2L_NT], x0[R2L_NT];
    // prologue: stage 0 (head bias) becomes current
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __syncthreads();
    P.lb = P.base + lane * 16;
    {
        F3None none;
        F3Side<true, F3None> s0{P.a1, P.lb, 0, none, false, F3Dma{false, P.rs, 0u, 0u, 0u}};
#pragma unroll
        for (int i = 0; i < 6; ++i) s0.step(i);
    }
    P.sb = P.ones;

    if constexpr (X0) {
        // X_0 from memory: lane (ray j, half h) holds features 32 T + 8 q + 4 h .. + 3 of its ray in fragment registers 4 q .. 4 q + 3
        const float* xr = a.x0_in + ray * R2L_W + 4 * h;  // (rows of the padding rays of the last tile exist: Np rows)
#pragma unroll
        for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 32 * T + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[T][4 * q + e] = v[e];
                    x0[T][4 * q + e] = v[e];
                }
            }
    } else {
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
        f3_stage<true, true, false>(x, P, F3Trig2{xc[0], 0}, F3Trig2{xc[0], 2});
    #pragma unroll 1
        for (int it2 = 0; it2 < 4; ++it2) {  // two samples = six coordinates = three pairs = 15 k-blocks per trip
            // coordinates of the NEXT trip (the last stage of this trip prepares the first B triple of the next one)
            float xn[6];
            {
                const float za = zsel(2 * it2 + 2), zb = zsel(2 * it2 + 3);
    #pragma unroll
                for (int ci = 0; ci < 6; ++ci) xn[ci] = o[ci % 3] + d[ci % 3] * (ci / 3 == 0 ? za : zb);
            }
    #pragma unroll
            for (int p = 0; p < 3; ++p) {
                const float xa = xc[2 * p], xb = xc[2 * p + 1];
                f3_stage<false, false, false>(x, P, F3Trig2{xa, 4}, F3Trig2{xa, 6});
                f3_stage<false, false, false>(x, P, F3Trig2{xa, 8}, F3Trig2{xb, 0});
                f3_stage<false, false, false>(x, P, F3Trig2{xb, 2}, F3Trig2{xb, 4});
                f3_stage<false, false, false>(x, P, F3Trig2{xb, 6}, F3Trig2{xb, 8});
                if (p < 2) {
                    f3_stage<false, false, false>(x, P, F3Trig2{xc[2 * p + 2], 0}, F3Trig2{xc[2 * p + 2], 2});
                } else {
                    f3_stage<false, false, false>(x, P, F3TrigOrIdent{it2 == 3, F3Trig2{xn[0], 0}, F3Ident4{o, d, z, 0}},
                                                  F3TrigOrIdent{it2 == 3, F3Trig2{xn[0], 2}, F3Ident4{o, d, z, 4}});
                }
            }
    #pragma unroll
            for (int ci = 0; ci < 6; ++ci) xc[ci] = xn[ci];
        }
        // identity features: coordinates 8j .. 8j+7 of the half
        f3_stage<false, false, false>(x, P, F3Ident4{o, d, z, 8}, F3Ident4{o, d, z, 12});
        f3_stage<false, false, false>(x, P, F3Ident4{o, d, z, 16}, F3Ident4{o, d, z, 20});
        f3_stage<false, false, true>(x, P, F3None{}, F3None{});
    #pragma unroll
        for (int T = 0; T < R2L_NT; ++T)
    #pragma unroll
            for (int c = 0; c < 16; ++c) {
                x[T][c] = fmaxf(x[T][c], 0.f);  // X_0 = relu(head)
                x0[T][c] = x[T][c];
            }
    }

    // ---- body -----------------------------------------------------------------------------------------------------------
    // training (SAVE): the B values of every stage are the layer's input, so the stash (x_b for the first layer of a block,
    // relu(t_b) for the second) is stored by the gatherers, two 16-byte pieces per stage (chunked layout: whole lines)
    const int64_t Np = R2L_PAD_ROWS(a.N);
    // (rows of the padding rays of the last tile exist: Np rows per slot)
    // chunked stash layout (r2l_common.h): lane base of the tile, pieces 1 KiB apart
    if (SAVE && blockIdx.x == 0 && threadIdx.x == 0)  // stash format word: chunked fp32 (also when this is the fallback launch)
        reinterpret_cast<unsigned*>(a.save_x)[R2L_STASH_FMT_WORD(a.n_block, Np)] = 1u;
    float* sx = SAVE ? a.save_x + r2l_chunk_lane(tile, lane & 31, h) : nullptr;
    float* st = SAVE ? a.save_t + r2l_chunk_lane(tile, lane & 31, h) : nullptr;
    const int64_t slot = R2L_TRIO_SLOT(Np);
#pragma unroll 1
    for (int b = 0; b < a.n_block; ++b) {
        // t = W1 x + b1   (its ReLU is applied where t is consumed)
        f3_stage<true, true, false>(t, P, F3Take4<false, SAVE>{x[0], 0, sx, 0}, F3Take4<false, SAVE>{x[0], 4, sx, 0});
#pragma unroll
        for (int kb = 0; kb < 15; ++kb)
            f3_stage<false, false, false>(t, P, F3Take4<false, SAVE>{x[(kb + 1) >> 1], 8 * ((kb + 1) & 1), sx, (kb + 1) >> 1},
                                          F3Take4<false, SAVE>{x[(kb + 1) >> 1], 8 * ((kb + 1) & 1) + 4, sx, (kb + 1) >> 1});
        f3_stage<false, false, true>(t, P, F3None{}, F3None{});
        // x += W2 relu(t) + b2   (training: the gatherers also shift [t > 0] into the block's four mask words)
        unsigned mw[4] = {0u, 0u, 0u, 0u};
        f3_stage<true, false, false>(x, P, F3Take4<true, SAVE>{t[0], 0, st, 0, &mw[0]}, F3Take4<true, SAVE>{t[0], 4, st, 0, &mw[0]});
#pragma unroll
        for (int kb = 0; kb < 15; ++kb)
            f3_stage<false, false, false>(x, P, F3Take4<true, SAVE>{t[(kb + 1) >> 1], 8 * ((kb + 1) & 1), st, (kb + 1) >> 1, &mw[(kb + 1) >> 2]},
                                          F3Take4<true, SAVE>{t[(kb + 1) >> 1], 8 * ((kb + 1) & 1) + 4, st, (kb + 1) >> 1, &mw[(kb + 1) >> 2]});
        f3_stage<false, false, true>(x, P, F3None{}, F3None{});  // next: the next block's bias stage (or the padding)
        if (SAVE) {  // values were shifted in MSB-first: bit (T&1)*16 + c after the reversal
            u32x4 mv;
#pragma unroll
            for (int w = 0; w < 4; ++w) mv[w] = __builtin_bitreverse32(mw[w]);
            *reinterpret_cast<u32x4*>(a.save_t + (int64_t)b * slot + R2L_MASK_OFFSET(Np) + tile * 256 + lane * 4) = mv;
        }
        if (SAVE) {
            sx += slot;
            st += slot;
        }
    }
    if (SAVE) {  // slot n: y = x_n + x_0, row-major (the tail weight gradient reads nothing else)
        float* sy = a.save_x + (int64_t)a.n_block * slot + ray * R2L_W + 4 * h;
#pragma unroll
        for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4*>(sy + 32 * T + 8 * q) =
                    f32x4{x[T][4 * q] + x0[T][4 * q], x[T][4 * q + 1] + x0[T][4 * q + 1], x[T][4 * q + 2] + x0[T][4 * q + 2],
                          x[T][4 * q + 3] + x0[T][4 * q + 3]};
    }

    // ---- tail: rgb = sigmoid(Wt (x + X_0) + bt) on the VALU -------------------------------------------------------------
    const float* tw = a.params + f3_off_tail_w(a.n_block) + 4 * h;
    float p3[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 wv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) wv[c] = *reinterpret_cast<const f32x4*>(tw + c * R2L_W + 32 * T + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = x[T][4 * q + e] + x0[T][4 * q + e];
#pragma unroll
                for (int c = 0; c < 3; ++c) p3[c] = __builtin_fmaf(wv[c][e], y, p3[c]);
            }
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) p3[c] += __shfl_xor(p3[c], 32);
    if (valid && h == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = p3[c] + a.params[f3_off_tail_b(a.n_block) + c];
            a.rgb[ray * 3 + c] = 1.0f / (1.0f + expf(-v));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
int r2l_fwd3_pack(const float* params, int n_block, float* wstream3, hipStream_t stream, const unsigned* run_if) {
    hipLaunchKernelGGL(r2l_pack_fwd3_kernel, dim3(2048), dim3(256), 0, stream, params,
                       reinterpret_cast<unsigned short*>(wstream3), n_block, run_if);
    R2L_CHECK(hipGetLastError());
    return 0;
}

int r2l_fwd2_fallback_pack(const float* params, int n_block, float* wstream3, float* wstream2, hipStream_t stream) {
    hipLaunchKernelGGL(r2l_fwd2_fallback_pack_kernel, dim3(2048), dim3(256), 0, stream, params,
                       reinterpret_cast<unsigned short*>(wstream3), reinterpret_cast<unsigned short*>(wstream2), n_block,
                       reinterpret_cast<unsigned*>(wstream2 + r2l_fwd2_status_offset(n_block)));
    R2L_CHECK(hipGetLastError());
    return 0;
}

int r2l_fwd3_forward(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab,
                     const float* c2w_host12, int H, int W, float focal, const float* wstream3, const float* params,
                     int n_block, float* rgb, float* save_x, float* save_t, int64_t N, hipStream_t stream,
                     const unsigned* run_if, const float* x0_in) {
    F3Args a{};
    a.run_if = run_if;
    a.rays_o = rays_o; a.rays_d = rays_d; a.t_rand = t_rand; a.ztab = ztab;
    a.stream = reinterpret_cast<const unsigned char*>(wstream3); a.params = params;
    a.n_block = n_block; a.rgb = rgb; a.save_x = save_x; a.save_t = save_t; a.N = N; a.H = H; a.Wimg = W; a.focal = focal;
    if (c2w_host12) for (int i = 0; i < 12; ++i) a.c2w[i] = c2w_host12[i];
    a.c2w_dev = c2w_host12 ? g_r2l_c2w_dev : nullptr;
    const int64_t tiles = (N + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    if (x0_in != nullptr) {  // body + tail from a given X_0: the stream starts at the first body stage
        a.x0_in = x0_in;
        a.stream += (size_t)64 * F3_STAGE_BYTES;
        hipLaunchKernelGGL((r2l_fwd3_kernel<false, false, true>), grid, block, 0, stream, a);
    } else if (c2w_host12) hipLaunchKernelGGL((r2l_fwd3_kernel<true, false>), grid, block, 0, stream, a);
    else if (save_x) hipLaunchKernelGGL((r2l_fwd3_kernel<false, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((r2l_fwd3_kernel<false, false>), grid, block, 0, stream, a);
    R2L_CHECK(hipGetLastError());
    return 0;
}
