// r2l_fwd2.hip — the R2L student forward on the fp16 matrix pipe with two-way operand splits (see r2l_f2.h): every fp32
// product as three fp16 MFMA products, ~2^-21 relative.  Structure, stage order and gatherers are r2l_fwd3.hip's; stages are
// 16 KiB ([split 2][tile 8][lane 64][8 fp16]).  Default of every one-wave-per-tile forward launch (render / evaluation and the training forward with its stash);
// R2L_NO_FWD2=1: bf16x3 only.
#include "r2l_f2.h"
#include "r2l_coopf.h"

__host__ __device__ static inline int64_t f2_off_head_b() { return (int64_t)R2L_IN * R2L_W; }
__host__ __device__ static inline int64_t f2_off_body_w(int layer) {
    return (int64_t)R2L_IN * R2L_W + R2L_W + (int64_t)layer * (R2L_W * R2L_W + R2L_W);
}
__host__ __device__ static inline int64_t f2_off_body_b(int layer) { return f2_off_body_w(layer) + R2L_W * R2L_W; }
__host__ __device__ static inline int64_t f2_off_tail_w(int n_block) { return f2_off_body_w(2 * n_block); }
__host__ __device__ static inline int64_t f2_off_tail_b(int n_block) { return f2_off_tail_w(n_block) + 3 * R2L_W; }

// =================================================================================================================
// pack: flat fp32 parameters -> stage stream, stage order and slot numbering exactly as r2l_pack_fwd2_kernel; a stage is
// [split sp (2)][tile t][lane (i,h)][slot s] fp16.  bias stage: split region 0 only: slots 0, 1 of half 0 = hi, mid.
// =================================================================================================================
__device__ __forceinline__ unsigned short f2_bits(_Float16 v) { return __builtin_bit_cast(unsigned short, v); }
__global__ void r2l_pack_fwd2_kernel(const float* __restrict__ params, unsigned short* __restrict__ out, int n_block) {
    const int64_t stages = r2l_fwd3_stages(n_block);
    const int64_t total = (stages + R2L_F3_PAD_STAGES) * 8 * 64 * 8;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(idx & 7), lane = (int)((idx >> 3) & 63), tile = (int)((idx >> 9) & 7);
        const int64_t g = idx >> 12;
        const int i = lane & 31, h = lane >> 5, o = 32 * tile + i;
        unsigned short* st = out + g * (F2_STAGE_BYTES / 2);
        unsigned short v0 = 0, v1 = 0;
        if (g < stages) {
            bool bias_stage = false;
            float w = 0.f;
            if (g == 0) {
                bias_stage = true;
                w = params[f2_off_head_b() + o];
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
                    w = params[f2_off_body_b(layer) + o];
                } else {
                    const int kb = r17 - 1, T = kb >> 1, r = kb & 1;
                    const int in = 32 * T + 8 * (2 * r + (s >> 2)) + 4 * h + (s & 3);
                    w = params[f2_off_body_w(layer) + (int64_t)o * R2L_W + in];
                }
            }
            const _Float16 hi = (_Float16)w;
            const _Float16 mid = (_Float16)(w - (float)hi);
            if (bias_stage) {
                v0 = (h == 0) ? (s == 0 ? f2_bits(hi) : (s == 1 ? f2_bits(mid) : (unsigned short)0)) : (unsigned short)0;
            } else {
                v0 = f2_bits(hi); v1 = f2_bits(mid);
            }
        }
        const int64_t e = ((int64_t)tile * 64 + lane) * 8 + s;
        st[e] = v0;
        st[8 * 64 * 8 + e] = v1;
    }
}
// status word behind the stages: cleared by every pack (new weights: new chance)
__global__ void r2l_fwd2_status_clear_kernel(unsigned* status) {
    if (threadIdx.x < 16) status[threadIdx.x] = 0u;
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

    // an earlier launch with these weights left fp16's range: the bf16x3 kernel behind this one does the work
    if (__builtin_nontemporal_load(a.status) != 0u) return;
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
    f2_stage<true, true, false>(x, P, F3Trig2{xc[0], 0}, F3Trig2{xc[0], 2});
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
            f2_stage<false, false, false>(x, P, F3Trig2{xa, 4}, F3Trig2{xa, 6});
            f2_stage<false, false, false>(x, P, F3Trig2{xa, 8}, F3Trig2{xb, 0});
            f2_stage<false, false, false>(x, P, F3Trig2{xb, 2}, F3Trig2{xb, 4});
            f2_stage<false, false, false>(x, P, F3Trig2{xb, 6}, F3Trig2{xb, 8});
            if (p < 2) {
                f2_stage<false, false, false>(x, P, F3Trig2{xc[2 * p + 2], 0}, F3Trig2{xc[2 * p + 2], 2});
            } else {
                f2_stage<false, false, false>(x, P, F3TrigOrIdent{it2 == 3, F3Trig2{xn[0], 0}, F3Ident4{o, d, z, 0}},
                                              F3TrigOrIdent{it2 == 3, F3Trig2{xn[0], 2}, F3Ident4{o, d, z, 4}});
            }
        }
#pragma unroll
        for (int ci = 0; ci < 6; ++ci) xc[ci] = xn[ci];
    }
    // identity features: coordinates 8j .. 8j+7 of the half
    f2_stage<false, false, false>(x, P, F3Ident4{o, d, z, 8}, F3Ident4{o, d, z, 12});
    f2_stage<false, false, false>(x, P, F3Ident4{o, d, z, 16}, F3Ident4{o, d, z, 20});
    f2_stage<false, false, true>(x, P, F3None{}, F3None{});
#pragma unroll
    for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            x[T][c] = fmaxf(x[T][c], 0.f);  // X_0 = relu(head)
            x0[T][c] = x[T][c];
        }

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
    if (SAVE && blockIdx.x == 0 && threadIdx.x == 0)  // stash format word: fp16 stage pieces (a fallback launch overwrites it)
        reinterpret_cast<unsigned*>(a.save_x)[R2L_STASH_FMT_WORD(a.n_block, Np)] = 0u;
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

    // range guard: a value that close to 65504 may have become inf in a product's operand -> hand the launch over
    if (!(P.amax < R2L_F2_RANGE)) atomicOr(a.status, 1u);

    // ---- tail: rgb = sigmoid(Wt (x + X_0) + bt) on the VALU -------------------------------------------------------------
    const float* tw = a.params + f2_off_tail_w(a.n_block) + 4 * h;
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
            const float v = p3[c] + a.params[f2_off_tail_b(a.n_block) + c];
            a.rgb[ray * 3 + c] = 1.0f / (1.0f + expf(-v));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
int r2l_fwd2_pack(const float* params, int n_block, float* wstream2, hipStream_t stream) {
    hipLaunchKernelGGL(r2l_pack_fwd2_kernel, dim3(2048), dim3(256), 0, stream, params,
                       reinterpret_cast<unsigned short*>(wstream2), n_block);
    R2L_CHECK(hipGetLastError());
    hipLaunchKernelGGL(r2l_fwd2_status_clear_kernel, dim3(1), dim3(64), 0, stream,
                       reinterpret_cast<unsigned*>(wstream2 + r2l_fwd2_status_offset(n_block)));
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
