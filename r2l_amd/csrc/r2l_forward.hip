// r2l_forward.hip — fused R2L student forward for gfx950 (MI355X):
//   ray -> 16 sample points -> positional encoding (1008-d, never materialised) -> Linear(1008,256)+ReLU
//   -> n_block x [Linear+ReLU+Linear + residual] -> outer residual -> Linear(256,3)+Sigmoid
// replacing the op sequence of /root/reference/model/nerf_raybased.py:94-126 (PointSampler),
// :198-208 (PositionalEmbedder.__call__), :461-465 (ResMLP.forward), :539-544 (NeRF_v3_2.forward).
//
// One wavefront = 32 rays for the whole network, activations register-resident in MFMA fragment layout
// (see r2l_common.h); exact-fp32 v_mfma_f32_32x32x2_f32; 4 independent waves per 256-thread block, one per SIMD.
#include "r2l_common.h"

struct R2LFwdArgs {
    // inputs (exactly one of {rays_o/rays_d}, {pose}, {emb} is used, selected by the MODE template argument)
    const float* rays_o;   // [N,3]
    const float* rays_d;   // [N,3]
    const float* t_rand;   // [N,16] stratified jitter U[0,1) or nullptr (perturb == 0)
    const float* emb;      // [N,1008] pre-embedded input (module-boundary compatibility path)
    const float* ztab;     // [32]: z_lower[16], z_span[16]  (z = z_lower + z_span * t_rand ; z = z_lower if !t_rand)
    float c2w[12];         // row-major [3,4] camera-to-world (pose mode)
    const float* c2w_dev;  // several frames per launch: [K][12] on the device (r2l_common.h r2l_pose_of), else nullptr
    int H, Wimg;
    float focal;
    // parameters
    const float* wstream;  // packed head+body weight stream (r2l_pack.hip)
    const float* params;   // flat fp32 parameter buffer in state_dict order (biases and tail are read from here)
    int n_block;
    // outputs
    float* rgb;            // [N,3]
    float* save_x;         // [(n_block+1)][Np][256] (Np = N padded to 32)  X_0 (=relu(head)), X_1 .. X_n   or nullptr
    float* save_t;         // [n_block][Np][256]     relu(hidden) of each block      or nullptr
    int64_t N;
};

// ---- flat parameter buffer offsets (state_dict order: head.0.{weight,bias}, body.b.body.{0,2}.{weight,bias}, tail.0.*)
__host__ __device__ __forceinline__ int64_t off_head_b() { return (int64_t)R2L_IN * R2L_W; }
__host__ __device__ __forceinline__ int64_t off_body_w(int layer) {
    return (int64_t)R2L_IN * R2L_W + R2L_W + (int64_t)layer * (R2L_W * R2L_W + R2L_W);
}
__host__ __device__ __forceinline__ int64_t off_body_b(int layer) { return off_body_w(layer) + R2L_W * R2L_W; }
__host__ __device__ __forceinline__ int64_t off_tail_w(int n_block) { return off_body_w(2 * n_block); }
__host__ __device__ __forceinline__ int64_t off_tail_b(int n_block) { return off_tail_w(n_block) + 3 * R2L_W; }

enum { MODE_RAYS = 0, MODE_POSE = 1, MODE_EMB = 2 };

template <int MODE, bool SAVE>
__global__ __launch_bounds__(256, 1) void r2l_fwd_kernel(const R2LFwdArgs a) {
    __shared__ float stash[4][R2L_NT * 16][64];  // X_0 of each wave's tile, needed again for the outer residual

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    const int64_t ray = ((int64_t)blockIdx.x * 4 + wave) * R2L_TILE_RAYS + (lane & 31);
    if ((int64_t)(blockIdx.x * 4 + wave) * R2L_TILE_RAYS >= a.N) return;  // whole wave idle (wave-uniform)
    const bool valid = ray < a.N;
    const int64_t rc = valid ? ray : a.N - 1;

    // ring slots (r2l_common.h): stream position 0 = head bias (slot 0), 1..120 trig groups, 121..126 identity groups,
    // 127 + 66 b + i = group i of block b; all loops over groups have even trip lengths, so slots are static.
    WRing2 ws;
    ws.init(a.wstream, lane);

    f32x16 x[R2L_NT], t[R2L_NT];
    const float one_h0 = h ? 0.f : 1.f;  // B operand of the bias k-step: k = 0 is carried by the lower half-wave
    mfma_bias_group<true, 0>(x, ws, one_h0);  // x = head bias

    if constexpr (MODE == MODE_EMB) {
        // B operand straight from the embedded input: stream step s of half h reads emb[ray][s + 504 h]
        const float* e = a.emb + rc * R2L_IN + 504 * h;
        // same step order as the fused path: trig steps (sample it, axis, f) then identity steps
#pragma unroll 1
        for (int it2 = 0; it2 < 4; ++it2) {
#pragma unroll
            for (int li = 0; li < 30; ++li) {  // (sample 2*it2 + li/15, axis (li%15)/5, group li%5)
                const float* ec = e + ((2 * it2 + li / 15) * 3 + (li % 15) / 5) * 21 + 4 * (li % 5);
                if ((1 + li) % 2 == 0) mfma_group<0>(x, ws, ec[0], ec[1], ec[2], ec[3]);
                else mfma_group<1>(x, ws, ec[0], ec[1], ec[2], ec[3]);
            }
        }
#pragma unroll
        for (int g = 0; g < R2L_HEAD_ID_GROUPS; ++g) {
            const float i0 = e[(4 * g + 0) * 21 + 20], i1 = e[(4 * g + 1) * 21 + 20], i2 = e[(4 * g + 2) * 21 + 20],
                        i3 = e[(4 * g + 3) * 21 + 20];
            if ((1 + g) % 2 == 0) mfma_group<0>(x, ws, i0, i1, i2, i3);
            else mfma_group<1>(x, ws, i0, i1, i2, i3);
        }
    } else {
        float o[3], d[3];
        if constexpr (MODE == MODE_RAYS) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                o[k] = a.rays_o[rc * 3 + k];
                d[k] = a.rays_d[rc * 3 + k];
            }
        } else {
            // PointSampler.__init__/sample_test (nerf_raybased.py:80-99): dirs = [(i-W/2)/f, -(j-H/2)/f, -1],
            // rays_d[k] = sum_b dirs[b] * c2w[k][b], rays_o = c2w[:,3]
            const R2LPoseRay pr = r2l_pose_of(a.c2w, a.c2w_dev, (int64_t)a.H * a.Wimg, rc);
            const int pj = (int)(pr.pix / a.Wimg), pi = (int)(pr.pix % a.Wimg);
            const float dx = ((float)pi - (float)a.Wimg * 0.5f) / a.focal;
            const float dy = -(((float)pj - (float)a.H * 0.5f) / a.focal);
            const float dz = -1.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d[k] = (dx * pr.c[4 * k + 0] + dy * pr.c[4 * k + 1]) + dz * pr.c[4 * k + 2];
                o[k] = pr.c[4 * k + 3];
            }
        }
        // the 8 sample depths of this half-wave (samples 8h .. 8h+7)
        float z[8];
        {
            const f32x4 lo0 = *reinterpret_cast<const f32x4*>(a.ztab + 8 * h);
            const f32x4 lo1 = *reinterpret_cast<const f32x4*>(a.ztab + 8 * h + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                z[k] = lo0[k];
                z[4 + k] = lo1[k];
            }
            if (a.t_rand != nullptr) {  // z = lower + (upper-lower) * t_rand   (nerf_raybased.py:119-123)
                const f32x4 sp0 = *reinterpret_cast<const f32x4*>(a.ztab + 16 + 8 * h);
                const f32x4 sp1 = *reinterpret_cast<const f32x4*>(a.ztab + 16 + 8 * h + 4);
                const f32x4 u0 = *reinterpret_cast<const f32x4*>(a.t_rand + rc * 16 + 8 * h);
                const f32x4 u1 = *reinterpret_cast<const f32x4*>(a.t_rand + rc * 16 + 8 * h + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    z[k] = lo0[k] + sp0[k] * u0[k];
                    z[4 + k] = lo1[k] + sp1[k] * u1[k];
                }
            }
        }
        // trig features: per coordinate [sin(2^0 x) .. sin(2^9 x), cos(2^0 x) .. cos(2^9 x)]   (nerf_raybased.py:199-203)
        // Software pipeline: while the 5 groups (160 MFMAs) of coordinate c run, the VALU evaluates the 10 sin/cos
        // pairs of coordinate c+1 (two per group, slotted between the MFMAs by the schedule pin).
        auto zsel = [&](int s) {  // z[s] for a runtime sample index (registers cannot be indexed dynamically)
            float zz = z[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) zz = (s == k) ? z[k] : zz;
            return zz;
        };
        float fc[20], fn[20];
        {
            const float xc = o[0] + d[0] * z[0];  // pts = o + d*z, mul and add rounded separately (-ffp-contract=off)
#pragma unroll
            for (int k = 0; k < R2L_L; ++k) r2l_sincos(xc * (float)(1 << k), fc[k], fc[R2L_L + k]);
        }
#pragma unroll 1
        for (int it2 = 0; it2 < 4; ++it2) {
            const float za = zsel(2 * it2), zb = zsel(2 * it2 + 1), zc = zsel(2 * it2 + 2 < 8 ? 2 * it2 + 2 : 7);
#pragma unroll
            for (int ci = 0; ci < 6; ++ci) {  // coordinates (sample 2*it2 + ci/3, axis ci%3)
                // the NEXT coordinate (the one after the last is a harmless recomputation)
                const int nax = (ci + 1) % 3;
                const float nz = (ci + 1) / 3 == 0 ? za : ((ci + 1) / 3 == 1 ? zb : zc);
                const float xn = o[nax] + d[nax] * nz;
#pragma unroll
                for (int g = 0; g < 5; ++g) {
                    r2l_sincos(xn * (float)(1 << (2 * g)), fn[2 * g], fn[R2L_L + 2 * g]);
                    r2l_sincos(xn * (float)(1 << (2 * g + 1)), fn[2 * g + 1], fn[R2L_L + 2 * g + 1]);
                    const int li = ci * 5 + g;
                    if ((1 + li) % 2 == 0) mfma_group<0, 0, 0, 9>(x, ws, fc[4 * g + 0], fc[4 * g + 1], fc[4 * g + 2], fc[4 * g + 3]);
                    else mfma_group<1, 0, 0, 9>(x, ws, fc[4 * g + 0], fc[4 * g + 1], fc[4 * g + 2], fc[4 * g + 3]);
                }
#pragma unroll
                for (int k = 0; k < 20; ++k) fc[k] = fn[k];
            }
        }
        // identity features (the trailing x of each coordinate's 21)
        {
            float id[24];
#pragma unroll
            for (int e = 0; e < 24; ++e) id[e] = o[e % 3] + d[e % 3] * z[e / 3];
#pragma unroll
            for (int g = 0; g < R2L_HEAD_ID_GROUPS; ++g) {
                if ((1 + g) % 2 == 0) mfma_group<0>(x, ws, id[4 * g + 0], id[4 * g + 1], id[4 * g + 2], id[4 * g + 3]);
                else mfma_group<1>(x, ws, id[4 * g + 0], id[4 * g + 1], id[4 * g + 2], id[4 * g + 3]);
            }
        }
    }
    relu_inplace(x);

    // stash X_0 for the outer residual  (NeRF_v3_2.forward: body(x) + x, nerf_raybased.py:543)
#pragma unroll
    for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
        for (int c = 0; c < 16; ++c) stash[wave][T * 16 + c][lane] = x[T][c];

    // body: x <- x + W2 relu(W1 x + b1) + b2     (ResMLP.forward with res_scale 1, nerf_raybased.py:461-465)
    const int64_t Np = R2L_PAD_ROWS(a.N);  // rows per stash slot
#pragma unroll 1
    for (int b = 0; b < a.n_block; ++b) {
        // t = W1 x + b1 (pre-activation; its ReLU is applied on the fly when t feeds the second GEMM)
        mfma_bias_group<true, 1>(t, ws, one_h0);  // block-local group 0 -> ring slot 1
        if constexpr (SAVE) {  // X_b is this GEMM's B operand: its stash store rides along, one 16-byte piece per group
            StoreHook sx(a.save_x + (int64_t)b * Np * R2L_W, ray, h, x);
            gemm256x<false, 0>(t, x, ws, sx);
        } else {
            NoHook nh;
            gemm256x<false, 0>(t, x, ws, nh);
        }
        // x += W2 relu(t) + b2
        mfma_bias_group<false, 0>(x, ws, one_h0);  // block-local group 33 -> slot 0
        if constexpr (SAVE) {
            // relu(t) rides along as this GEMM's B operand; its signs are folded into 128 bits per lane on the way and leave as
            // ONE 16-byte store per lane (1 KiB per tile): the dX chain reads those instead of relu(t) (r2l_common.h)
            unsigned mb[4] = {0u, 0u, 0u, 0u};
            StoreMaskHook st(a.save_t + (int64_t)b * Np * R2L_W, ray, h, t, mb);
            gemm256x<true, 1>(x, t, ws, st);
            *reinterpret_cast<u32x4*>(a.save_t + r2l_mask32_offset(a.n_block, Np, b) + (ray >> 5) * 256 + lane * 4) =
                u32x4{mb[0], mb[1], mb[2], mb[3]};
        } else {
            NoHook nh;
            gemm256x<true, 1>(x, t, ws, nh);
        }
    }
    if constexpr (SAVE) store_frag(a.save_x + (int64_t)a.n_block * Np * R2L_W, ray, h, x);

    // tail: rgb = sigmoid(Wt (x + X_0) + bt)
    const float* tw = a.params + off_tail_w(a.n_block);
    float acc3[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 wv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) wv[c] = *reinterpret_cast<const f32x4*>(tw + c * R2L_W + 32 * T + 8 * q + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = x[T][4 * q + j] + stash[wave][T * 16 + 4 * q + j][lane];
#pragma unroll
                for (int c = 0; c < 3; ++c) acc3[c] = __builtin_fmaf(wv[c][j], y, acc3[c]);
            }
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        acc3[c] += __shfl_xor(acc3[c], 32);
        const float v = acc3[c] + a.params[off_tail_b(a.n_block) + c];
        acc3[c] = 1.0f / (1.0f + expf(-v));
    }
    if (valid && h == 0) {
        a.rgb[ray * 3 + 0] = acc3[0];
        a.rgb[ray * 3 + 1] = acc3[1];
        a.rgb[ray * 3 + 2] = acc3[2];
    }
}

template <int MODE>
static int launch_fwd(const R2LFwdArgs& a, hipStream_t stream) {
    if (a.N <= 0) return 0;
    const int64_t tiles = (a.N + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    if (a.save_x != nullptr)
        hipLaunchKernelGGL((r2l_fwd_kernel<MODE, true>), grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL((r2l_fwd_kernel<MODE, false>), grid, block, 0, stream, a);
    R2L_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// C ABI (declared in include/r2l_hip.h)
// ------------------------------------------------------------------------------------------------------------------
extern "C" int r2l_forward_rays(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab,
                                const float* wstream, const float* params, int n_block, float* rgb, float* save_x,
                                float* save_t, int64_t N, void* stream) {
    return r2l_forward_rays_cfg(rays_o, rays_d, t_rand, ztab, wstream, params, n_block, rgb, save_x, save_t, N, stream, nullptr);
}
extern "C" int r2l_forward_rays_cfg(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab,
                                    const float* wstream, const float* params, int n_block, float* rgb, float* save_x,
                                    float* save_t, int64_t N, void* stream, const r2l_config* cfg) {
    R2L_CFG_ENTER(cfg);
    R2L_REQUIRE(N >= 0 && n_block >= 0 && n_block <= R2L_MAX_BLOCKS, "r2l_forward_rays: N / n_block out of range");
    if (N == 0) return 0;
    R2L_REQUIRE(rays_o && rays_d && ztab && wstream && params && rgb, "r2l_forward_rays: a required pointer is NULL");
    R2L_REQUIRE((save_x == nullptr) == (save_t == nullptr) || n_block == 0, "r2l_forward_rays: save_x and save_t go together");
    R2LFwdArgs a{};
    a.rays_o = rays_o; a.rays_d = rays_d; a.t_rand = t_rand; a.ztab = ztab;
    a.wstream = wstream; a.params = params; a.n_block = n_block;
    a.rgb = rgb; a.save_x = save_x; a.save_t = save_t; a.N = N;
    const int variant = N > 0 ? r2l_chain_variant(N) : R2L_VARIANT_MAIN;
    if (variant == R2L_VARIANT_COOP16)
        return r2l_coop16_forward(rays_o, rays_d, t_rand, ztab, nullptr, 0, 0, 0.f, wstream + r2l_fwd32_stream_floats(n_block),
                                  params, n_block, rgb, save_x, save_t, N, (hipStream_t)stream);
    // (with the training stash: only as part of the default fp16 trio, whose stash format it writes)
    if (N > 0 && (save_x != nullptr ? r2l_use_trio16() : r2l_use_fwd2())) {
        const float* w3 = wstream + r2l_fwd32_stream_floats(n_block) + r2l_fwd16_stream_floats(n_block);
        const float* w2 = w3 + r2l_fwd3_stream_floats(n_block);
        const int rc = r2l_fwd2_forward(rays_o, rays_d, t_rand, ztab, nullptr, 0, 0, 0.f, w2, params, n_block, rgb, save_x,
                                        save_t, N, (hipStream_t)stream);
        if (rc) return rc;
        // range-guard fallback: stream pack + launch, both returning at once unless the fp16x2 launch raised its status word
        const unsigned* st = reinterpret_cast<const unsigned*>(w2 + r2l_fwd2_status_offset(n_block));
        const int rp = r2l_fwd2_fallback_pack(params, n_block, const_cast<float*>(w3), const_cast<float*>(w2), (hipStream_t)stream);
        if (rp) return rp;
        return r2l_fwd3_forward(rays_o, rays_d, t_rand, ztab, nullptr, 0, 0, 0.f, w3, params, n_block, rgb, save_x, save_t, N,
                                (hipStream_t)stream, st + F2S_GO);
    }
    if (N > 0 && r2l_use_fwd3())
        return r2l_fwd3_forward(rays_o, rays_d, t_rand, ztab, nullptr, 0, 0, 0.f,
                                wstream + r2l_fwd32_stream_floats(n_block) + r2l_fwd16_stream_floats(n_block), params,
                                n_block, rgb, save_x, save_t, N, (hipStream_t)stream);
    return launch_fwd<MODE_RAYS>(a, (hipStream_t)stream);
}

extern "C" int r2l_forward_pose(const float* c2w_host12, int H, int W, float focal, const float* ztab,
                                const float* wstream, const float* params, int n_block, float* rgb, void* stream) {
    return r2l_forward_pose_cfg(c2w_host12, H, W, focal, ztab, wstream, params, n_block, rgb, stream, nullptr);
}
static int forward_pose_impl(const float* c2w_host12, int64_t n_frames, int H, int W, float focal, const float* ztab,
                             const float* wstream, const float* params, int n_block, float* rgb, void* stream);

extern "C" int r2l_forward_pose_cfg(const float* c2w_host12, int H, int W, float focal, const float* ztab,
                                    const float* wstream, const float* params, int n_block, float* rgb, void* stream,
                                    const r2l_config* cfg) {
    R2L_CFG_ENTER(cfg);
    R2L_REQUIRE(H > 0 && W > 0 && focal != 0.f && n_block >= 0 && n_block <= R2L_MAX_BLOCKS, "r2l_forward_pose: H / W / focal / n_block out of range");
    R2L_REQUIRE(c2w_host12 && ztab && wstream && params && rgb, "r2l_forward_pose: a required pointer is NULL");
    return forward_pose_impl(c2w_host12, 1, H, W, focal, ztab, wstream, params, n_block, rgb, stream);
}

// K frames in ONE launch: rgb[K*H*W,3], poses from a device table (include/r2l_hip.h).  The cooperative tilings (small
// launches; only when pinned by the config / environment) take one pose by value: not here.
extern "C" int r2l_forward_poses_cfg(const float* c2w_dev, int K, int H, int W, float focal, const float* ztab,
                                     const float* wstream, const float* params, int n_block, float* rgb, void* stream,
                                     const r2l_config* cfg) {
    R2L_CFG_ENTER(cfg);
    if (K <= 0) return 0;
    R2L_REQUIRE(H > 0 && W > 0 && focal != 0.f && n_block >= 0 && n_block <= R2L_MAX_BLOCKS, "r2l_forward_poses: H / W / focal / n_block out of range");
    R2L_REQUIRE(ztab && wstream && params && rgb, "r2l_forward_poses: a required pointer is NULL");
    const int64_t N = (int64_t)K * H * W;
    if (c2w_dev == nullptr || r2l_chain_variant(N) != R2L_VARIANT_MAIN || r2l_use_coopf(N, n_block)) {
        r2l_set_error_msg("r2l_forward_poses: needs a device pose table and a one-wave-per-tile tiling (cooperative tilings: "
                          "call r2l_forward_pose per frame)");
        return (int)hipErrorInvalidValue;
    }
    const float dummy[12] = {0};
    g_r2l_c2w_dev = c2w_dev;
    const int rc = forward_pose_impl(dummy, K, H, W, focal, ztab, wstream, params, n_block, rgb, stream);
    g_r2l_c2w_dev = nullptr;
    return rc;
}

static int forward_pose_impl(const float* c2w_host12, int64_t n_frames, int H, int W, float focal, const float* ztab,
                             const float* wstream, const float* params, int n_block, float* rgb, void* stream) {
    R2LFwdArgs a{};
    for (int i = 0; i < 12; ++i) a.c2w[i] = c2w_host12[i];
    a.c2w_dev = g_r2l_c2w_dev;
    a.H = H; a.Wimg = W; a.focal = focal; a.ztab = ztab;
    a.wstream = wstream; a.params = params; a.n_block = n_block;
    a.rgb = rgb; a.N = n_frames * H * W;
    const int variant = a.N > 0 ? r2l_chain_variant(a.N) : R2L_VARIANT_MAIN;
    if (variant == R2L_VARIANT_COOP16)
        return r2l_coop16_forward(nullptr, nullptr, nullptr, ztab, c2w_host12, H, W, focal,
                                  wstream + r2l_fwd32_stream_floats(n_block), params, n_block, rgb, nullptr, nullptr, a.N,
                                  (hipStream_t)stream);
    if (a.N > 0 && r2l_use_fwd2()) {
        const float* w3 = wstream + r2l_fwd32_stream_floats(n_block) + r2l_fwd16_stream_floats(n_block);
        const float* w2 = w3 + r2l_fwd3_stream_floats(n_block);
        const int rc = r2l_fwd2_forward(nullptr, nullptr, nullptr, ztab, c2w_host12, H, W, focal, w2, params, n_block, rgb,
                                        nullptr, nullptr, a.N, (hipStream_t)stream);
        if (rc) return rc;
        const unsigned* st = reinterpret_cast<const unsigned*>(w2 + r2l_fwd2_status_offset(n_block));
        const int rp = r2l_fwd2_fallback_pack(params, n_block, const_cast<float*>(w3), const_cast<float*>(w2), (hipStream_t)stream);
        if (rp) return rp;
        return r2l_fwd3_forward(nullptr, nullptr, nullptr, ztab, c2w_host12, H, W, focal, w3, params, n_block, rgb, nullptr,
                                nullptr, a.N, (hipStream_t)stream, st + F2S_GO);
    }
    if (a.N > 0 && r2l_use_fwd3())
        return r2l_fwd3_forward(nullptr, nullptr, nullptr, ztab, c2w_host12, H, W, focal,
                                wstream + r2l_fwd32_stream_floats(n_block) + r2l_fwd16_stream_floats(n_block), params,
                                n_block, rgb, nullptr, nullptr, a.N, (hipStream_t)stream);
    return launch_fwd<MODE_POSE>(a, (hipStream_t)stream);
}

// The 16 status words of the fp16x2 forward stream inside `wstream` (include/r2l_hip.h: range control, telemetry)
extern "C" const unsigned* r2l_forward_status_words(const float* wstream, int n_block) {
    if (wstream == nullptr || n_block < 0 || n_block > R2L_MAX_BLOCKS) return nullptr;
    const float* w2 = wstream + r2l_fwd32_stream_floats(n_block) + r2l_fwd16_stream_floats(n_block) + r2l_fwd3_stream_floats(n_block);
    return reinterpret_cast<const unsigned*>(w2 + r2l_fwd2_status_offset(n_block));
}

// The module-boundary path with a config (round 5).  precision = bf16x3 (or fp16x2: this path has no range-guard
// fallback, so the request is served by the bf16x3 chain), forward-only (no stash), x0_scratch given: the HEAD runs on the fp32-MFMA
// kernel (its B operand is the caller's encoding, read from HBM — the 16-bit chains take their B operands from registers and
// keep vmcnt for their weight DMA) as a zero-block launch that leaves X_0 = relu(head) row-major in x0_scratch, and the 86 body
// layers + tail run on r2l_fwd3_kernel<X0> (six bf16 products per fp32 product: fp32-grade, 1.7x the fp32-MFMA rate).  Anything
// else — AUTO / fp32_mfma, or a launch with the training stash (the backward of this path reads a row-major fp32 stash) — is
// r2l_forward_emb.
extern "C" int r2l_forward_emb_cfg(const float* emb, const float* wstream, const float* params, int n_block, float* rgb,
                                   float* save_x, float* save_t, int64_t N, float* x0_scratch, void* stream,
                                   const r2l_config* cfg) {
    R2L_CFG_ENTER(cfg);
    const bool chain16 = g_r2l_cfg.precision == R2L_PRECISION_BF16X3 || g_r2l_cfg.precision == R2L_PRECISION_FP16X2;
    if (!chain16 || save_x != nullptr || save_t != nullptr || x0_scratch == nullptr || n_block <= 0 || N <= 0)
        return r2l_forward_emb(emb, wstream, params, n_block, rgb, save_x, save_t, N, stream);
    R2L_REQUIRE(n_block <= R2L_MAX_BLOCKS, "r2l_forward_emb_cfg: n_block out of range");
    R2L_REQUIRE(emb && wstream && params && rgb, "r2l_forward_emb_cfg: a required pointer is NULL");
    R2LFwdArgs a{};
    a.emb = emb; a.wstream = wstream; a.params = params; a.n_block = 0;  // head only: X_0 -> x0_scratch (rgb: overwritten below)
    a.rgb = rgb; a.save_x = x0_scratch; a.save_t = nullptr; a.N = N;
    const int rc = launch_fwd<MODE_EMB>(a, (hipStream_t)stream);
    if (rc) return rc;
    return r2l_fwd3_forward(nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0.f,
                            wstream + r2l_fwd32_stream_floats(n_block) + r2l_fwd16_stream_floats(n_block), params, n_block, rgb,
                            nullptr, nullptr, N, (hipStream_t)stream, nullptr, x0_scratch);
}

extern "C" int r2l_forward_emb(const float* emb, const float* wstream, const float* params, int n_block, float* rgb,
                               float* save_x, float* save_t, int64_t N, void* stream) {
    R2L_REQUIRE(N >= 0 && n_block >= 0 && n_block <= R2L_MAX_BLOCKS, "r2l_forward_emb: N / n_block out of range");
    if (N == 0) return 0;
    R2L_REQUIRE(emb && wstream && params && rgb, "r2l_forward_emb: a required pointer is NULL");
    R2L_REQUIRE((save_x == nullptr) == (save_t == nullptr) || n_block == 0, "r2l_forward_emb: save_x and save_t go together");
    R2LFwdArgs a{};
    a.emb = emb; a.wstream = wstream; a.params = params; a.n_block = n_block;
    a.rgb = rgb; a.save_x = save_x; a.save_t = save_t; a.N = N;
    return launch_fwd<MODE_EMB>(a, (hipStream_t)stream);
}
