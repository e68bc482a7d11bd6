// r2l_coop.hip — small-batch variants of the R2L student chains (forward and dX backward) for gfx950.
//
// The main kernels (r2l_forward.hip / r2l_backward.hip) give one wavefront a whole 32-ray tile for the whole network:
// perfect for >= 32 768 rays (1024 wave slots x 32), but a 4096-ray training step (BASELINE configs[2] read literally)
// would occupy 128 of the 1024 SIMDs.  Here the FOUR waves of a workgroup share one 32-ray tile: wave w owns output
// features [64w, 64w+64) (two 32-row MFMA tiles) of every layer and reads all 256 input features as B operands from a
// double-buffered LDS copy of the activation ([32 rays][256+4] fp32, padded against ds_read_b128 bank conflicts).  One
// barrier per layer.  The SAME packed weight streams are read (each wave only its two tiles of every group), through a
// ring of 8 groups (a group is now 8 MFMAs = 512 cycles).  Semantics, stash layout and gradients are identical to the
// main kernels; the host picks the variant by N (r2l_forward_rays / r2l_backward: N < R2L_COOP_MAX_RAYS).
#include "r2l_common.h"

#define COOP_LD 260                       // LDS row pitch in floats (256 + 4: 16-byte skew per ray)
#define COOP_ACT_FLOATS (32 * COOP_LD)    // one activation buffer

__host__ __device__ static inline int64_t c_off_head_b() { return (int64_t)R2L_IN * R2L_W; }
__host__ __device__ static inline int64_t c_off_body_w(int layer) {
    return (int64_t)R2L_IN * R2L_W + R2L_W + (int64_t)layer * (R2L_W * R2L_W + R2L_W);
}
__host__ __device__ static inline int64_t c_off_body_b(int layer) { return c_off_body_w(layer) + R2L_W * R2L_W; }
__host__ __device__ static inline int64_t c_off_tail_w(int n_block) { return c_off_body_w(2 * n_block); }
__host__ __device__ static inline int64_t c_off_tail_b(int n_block) { return c_off_tail_w(n_block) + 3 * R2L_W; }

// ---- weight ring of one wave: D groups x its 2 tiles -------------------------------------------------------------------
template <int D>
struct CRing {
    WPtr p;  // next group to LOAD (voff already includes this wave's tile offset)
    f32x4 w[D][2];
    __device__ __forceinline__ void init(const float* stream, int64_t first_group, int lane, int wave) {
        p.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(stream), 0, 0xffffffff, 0x00020000);
        p.voff = (unsigned)lane * 16u + (unsigned)(2 * wave) * 1024u;
        p.soff = (unsigned)(first_group * (R2L_GROUP_FLOATS * 4));
#pragma unroll
        for (int d = 0; d < D; ++d) {
            w[d][0] = p[0];
            w[d][1] = p[64];
            p += R2L_NT * 64;
        }
    }
    __device__ __forceinline__ void skip_group() { p += R2L_NT * 64; }
};

// acc[2 tiles] += W_group[own tiles] . b for the group in ring slot SLOT; reload the slot with the group D ahead.
// SKIP_BEFORE_LOAD: the group D ahead lies behind a bias group of the stream (layer boundary) -> jump over it first.
template <int SLOT, bool SKIP_BEFORE_LOAD = false, int VPT = 0, int D>
__device__ __forceinline__ void cgroup(f32x16 (&acc)[2], CRing<D>& r, float b0, float b1, float b2, float b3) {
    r.p.opaque();
    if (SKIP_BEFORE_LOAD) r.skip_group();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(r.w[SLOT][t][0], b0, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(r.w[SLOT][t][1], b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(r.w[SLOT][t][2], b2, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(r.w[SLOT][t][3], b3, acc[t], 0, 0, 0);
        r.w[SLOT][t] = r.p[t * 64];
    }
    r.p += R2L_NT * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if (VPT > 0) __builtin_amdgcn_sched_group_barrier(0x002, VPT, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// One 256 -> 64 (own slice) layer: acc += W[own rows] . bop.  BIAS_GROUPS: the stream has a bias group in front of
// every layer (forward stream): the prefetch of group G+8 must jump over it when it crosses into the next layer.
template <bool BIAS_GROUPS>
__device__ __forceinline__ void clayer(f32x16 (&acc)[2], const f32x16 (&bop)[R2L_NT], CRing<8>& r) {
#pragma unroll
    for (int G = 0; G < R2L_LAYER_GROUPS; ++G) {
        const int T = G >> 2, q = (G & 3) * 4;
        if (BIAS_GROUPS && G == R2L_LAYER_GROUPS - 8) {
            if ((G & 7) == 0) cgroup<0, true>(acc, r, bop[T][q], bop[T][q + 1], bop[T][q + 2], bop[T][q + 3]);
        } else {
            switch (G & 7) {
                case 0: cgroup<0>(acc, r, bop[T][q], bop[T][q + 1], bop[T][q + 2], bop[T][q + 3]); break;
                case 1: cgroup<1>(acc, r, bop[T][q], bop[T][q + 1], bop[T][q + 2], bop[T][q + 3]); break;
                case 2: cgroup<2>(acc, r, bop[T][q], bop[T][q + 1], bop[T][q + 2], bop[T][q + 3]); break;
                case 3: cgroup<3>(acc, r, bop[T][q], bop[T][q + 1], bop[T][q + 2], bop[T][q + 3]); break;
                case 4: cgroup<4>(acc, r, bop[T][q], bop[T][q + 1], bop[T][q + 2], bop[T][q + 3]); break;
                case 5: cgroup<5>(acc, r, bop[T][q], bop[T][q + 1], bop[T][q + 2], bop[T][q + 3]); break;
                case 6: cgroup<6>(acc, r, bop[T][q], bop[T][q + 1], bop[T][q + 2], bop[T][q + 3]); break;
                default: cgroup<7>(acc, r, bop[T][q], bop[T][q + 1], bop[T][q + 2], bop[T][q + 3]); break;
            }
        }
    }
}

// ---- LDS activation exchange -------------------------------------------------------------------------------------------
// all 256 features of the tile as B operands: lane (ray j, half h) register (T, 4q+e) <- act[j][32T + 8q + 4h + e]
__device__ __forceinline__ void lds_read_bops(const float* act, int lane, f32x16 (&bop)[R2L_NT]) {
    const float* row = act + (lane & 31) * COOP_LD + 4 * (lane >> 5);
#pragma unroll
    for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + 32 * T + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) bop[T][4 * q + e] = v[e];
        }
}
// this wave's 64-feature slice (fragment registers of its two tiles) -> act[j][64w + 32t + 8q + 4h + e]
__device__ __forceinline__ void lds_write_slice(float* act, int lane, int wave, const f32x16 (&v)[2]) {
    float* row = act + (lane & 31) * COOP_LD + 64 * wave + 4 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 o = {v[t][4 * q + 0], v[t][4 * q + 1], v[t][4 * q + 2], v[t][4 * q + 3]};
            *reinterpret_cast<f32x4*>(row + 32 * t + 8 * q) = o;
        }
}
// slice <-> row-major [N][256] global tensors (stash / gradients)
// Whole rows of the tile from the LDS activation buffer to a row-major [N][256] tensor: wave w moves rays 8w .. 8w+7, one
// 1 KiB row per instruction (full cache lines -> non-temporal stores pay off, r2l_common.h: r2l_stash_store_nt).
__device__ __forceinline__ void rows_lds_to_global(const float* act, float* base, int64_t tile_row0, int lane, int wave) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int ray = 8 * wave + r;
        const f32x4 v = *reinterpret_cast<const f32x4*>(act + ray * COOP_LD + 4 * lane);
        r2l_stash_store_nt(base + (tile_row0 + ray) * R2L_W + 4 * lane, v);
    }
}
__device__ __forceinline__ void g_store_slice(float* base, int64_t ray, int lane, int wave, const f32x16 (&v)[2]) {
    float* row = base + ray * R2L_W + 64 * wave + 4 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 o = {v[t][4 * q + 0], v[t][4 * q + 1], v[t][4 * q + 2], v[t][4 * q + 3]};
            r2l_stash_store(row + 32 * t + 8 * q, o);
        }
}
__device__ __forceinline__ void g_load_slice(const float* base, int64_t ray, int lane, int wave, f32x16 (&v)[2]) {
    const float* row = base + ray * R2L_W + 64 * wave + 4 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(row + 32 * t + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[t][4 * q + e] = o[e];
        }
}
// own-slice bias: acc[t][4q+e] (+)= bias[64w + 32t + 8q + 4h + e]
template <bool ACCUM>
__device__ __forceinline__ void slice_bias(f32x16 (&acc)[2], const float* bias, int lane, int wave) {
    const float* b = bias + 64 * wave + 4 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(b + 32 * t + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][4 * q + e] = ACCUM ? acc[t][4 * q + e] + v[e] : v[e];
        }
}

// =================================================================================================================
// forward
// =================================================================================================================
struct CoopFwdArgs {
    const float* rays_o;
    const float* rays_d;
    const float* t_rand;
    const float* ztab;
    float c2w[12];
    int H, Wimg;
    float focal;
    const float* wstream;
    const float* params;
    int n_block;
    float* rgb;
    float* save_x;
    float* save_t;
    int64_t N;
};

template <bool POSE, bool SAVE>
__global__ __launch_bounds__(256, 1) void r2l_fwd_coop_kernel(const CoopFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float act[2][COOP_ACT_FLOATS];
    __shared__ float tailred[4][32][4];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5;
    const int64_t tile_row0 = (int64_t)blockIdx.x * R2L_TILE_RAYS;
    const int64_t ray = tile_row0 + (lane & 31);
    const bool valid = ray < a.N;
    const int64_t rc = valid ? ray : a.N - 1;
    const int64_t Np = R2L_PAD_ROWS(a.N);

    // ---- head: every wave evaluates the whole 1008-d encoding as B operands and produces its 64 output features -----
    float o[3], d[3];
    if constexpr (!POSE) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = a.rays_o[rc * 3 + k];
            d[k] = a.rays_d[rc * 3 + k];
        }
    } else {
        const int pj = (int)(rc / a.Wimg), pi = (int)(rc % a.Wimg);
        const float dx = ((float)pi - (float)a.Wimg * 0.5f) / a.focal;
        const float dy = -(((float)pj - (float)a.H * 0.5f) / a.focal);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d[k] = (dx * a.c2w[4 * k + 0] + dy * a.c2w[4 * k + 1]) + (-1.0f) * a.c2w[4 * k + 2];
            o[k] = a.c2w[4 * k + 3];
        }
    }
    float z[8];
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
    f32x16 acc[2], xo[2], x0[2];  // accumulators / own slice of the residual stream x / own slice of X_0
    slice_bias<false>(acc, a.params + c_off_head_b(), lane, wave);
    // head stream: position 0 is the bias group (biases come from `params` here), useful groups at 1..126
    CRing<2> r2;
    r2.init(a.wstream, 1, lane, wave);
    {
        auto zsel = [&](int s) {
            float zz = z[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) zz = (s == k) ? z[k] : zz;
            return zz;
        };
#pragma unroll 1
        for (int it2 = 0; it2 < 4; ++it2) {
            const float za = zsel(2 * it2), zb = zsel(2 * it2 + 1);
#pragma unroll
            for (int ci = 0; ci < 6; ++ci) {  // coordinates (sample 2*it2 + ci/3, axis ci%3), 5 groups each
                const float xc = o[ci % 3] + d[ci % 3] * (ci / 3 == 0 ? za : zb);
                float f[20];
#pragma unroll
                for (int k = 0; k < R2L_L; ++k) r2l_sincos(xc * (float)(1 << k), f[k], f[R2L_L + k]);
#pragma unroll
                for (int g = 0; g < 5; ++g) {
                    const int li = ci * 5 + g;
                    if (li % 2 == 0) cgroup<0>(acc, r2, f[4 * g], f[4 * g + 1], f[4 * g + 2], f[4 * g + 3]);
                    else cgroup<1>(acc, r2, f[4 * g], f[4 * g + 1], f[4 * g + 2], f[4 * g + 3]);
                }
            }
        }
        float id[24];
#pragma unroll
        for (int e = 0; e < 24; ++e) id[e] = o[e % 3] + d[e % 3] * z[e / 3];
        // identity groups are stream positions 121..126; the prefetch two ahead of position 125 (= position 127) is the
        // bias group of body layer 0: jump over it (position 128 = first weight group of layer 0)
#pragma unroll
        for (int g = 0; g < R2L_HEAD_ID_GROUPS; ++g) {
            if (g == 4) cgroup<0, true>(acc, r2, id[4 * g], id[4 * g + 1], id[4 * g + 2], id[4 * g + 3]);
            else if (g % 2 == 0) cgroup<0>(acc, r2, id[4 * g], id[4 * g + 1], id[4 * g + 2], id[4 * g + 3]);
            else cgroup<1>(acc, r2, id[4 * g], id[4 * g + 1], id[4 * g + 2], id[4 * g + 3]);
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            xo[t][c] = fmaxf(acc[t][c], 0.f);  // X_0 = relu(head)
            x0[t][c] = xo[t][c];
        }
    lds_write_slice(act[0], lane, wave, xo);
    // widen the ring to 8 groups: r2 holds body-layer-0 groups 0 and 1 (slots 0, 1)
    CRing<8> r8;
    r8.p = r2.p;
#pragma unroll
    for (int t = 0; t < 2; ++t) { r8.w[0][t] = r2.w[0][t]; r8.w[1][t] = r2.w[1][t]; }
#pragma unroll
    for (int dd = 2; dd < 8; ++dd) {
        r8.w[dd][0] = r8.p[0];
        r8.w[dd][1] = r8.p[64];
        r8.p += R2L_NT * 64;
    }

    // ---- body -----------------------------------------------------------------------------------------------------------
    f32x16 bop[R2L_NT];
    const float* bias = a.params + c_off_body_b(0);
#pragma unroll 1
    for (int b = 0; b < a.n_block; ++b) {
        // t = relu(W1 x + b1): B operands = x from act[0]
        __syncthreads();
        lds_read_bops(act[0], lane, bop);
        if constexpr (SAVE) rows_lds_to_global(act[0], a.save_x + (int64_t)b * Np * R2L_W, tile_row0, lane, wave);  // x_b
        slice_bias<false>(acc, bias, lane, wave);
        clayer<true>(acc, bop, r8);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[t][c] = fmaxf(acc[t][c], 0.f);
        lds_write_slice(act[1], lane, wave, acc);
        // x += W2 t + b2: B operands = t from act[1]
        __syncthreads();
        lds_read_bops(act[1], lane, bop);
        if constexpr (SAVE) rows_lds_to_global(act[1], a.save_t + (int64_t)b * Np * R2L_W, tile_row0, lane, wave);  // t_b
        slice_bias<true>(xo, bias + (R2L_W * R2L_W + R2L_W), lane, wave);
        clayer<true>(xo, bop, r8);
        lds_write_slice(act[0], lane, wave, xo);
        bias += 2 * (R2L_W * R2L_W + R2L_W);
    }
    if constexpr (SAVE) {  // x_n (the loop stored x_0 .. x_{n-1} at the top of each trip)
        __syncthreads();
        rows_lds_to_global(act[0], a.save_x + (int64_t)a.n_block * Np * R2L_W, tile_row0, lane, wave);
    }

    // ---- tail: rgb = sigmoid(Wt (x + X_0) + bt), partial dot products per wave, reduced through LDS ----------------------
    const float* tw = a.params + c_off_tail_w(a.n_block) + 64 * wave + 4 * h;
    float p3[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 wv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) wv[c] = *reinterpret_cast<const f32x4*>(tw + c * R2L_W + 32 * t + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = xo[t][4 * q + e] + x0[t][4 * q + e];
#pragma unroll
                for (int c = 0; c < 3; ++c) p3[c] = __builtin_fmaf(wv[c][e], y, p3[c]);
            }
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) p3[c] += __shfl_xor(p3[c], 32);
    if (h == 0) {
        tailred[wave][lane][0] = p3[0];
        tailred[wave][lane][1] = p3[1];
        tailred[wave][lane][2] = p3[2];
    }
    __syncthreads();
    if (wave == 0 && h == 0 && valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = ((tailred[0][lane][c] + tailred[1][lane][c]) + (tailred[2][lane][c] + tailred[3][lane][c])) +
                            a.params[c_off_tail_b(a.n_block) + c];
            a.rgb[ray * 3 + c] = 1.0f / (1.0f + expf(-v));
        }
    }
}

// =================================================================================================================
// backward (dX chain)
// =================================================================================================================
struct CoopBwdArgs {
    const float* rgb;
    const float* target;
    const float* drgb;
    const float* save_x;
    const float* save_t;
    const float* wstream;  // transposed stream (no bias groups)
    const float* params;
    int n_block;
    float grad_scale;
    float* dpre;
    float* gx;
    float* gt;
    float* sqerr_partial;
    int64_t N;
};

__global__ __launch_bounds__(256, 1) void r2l_bwd_coop_kernel(const CoopBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float act[2][COOP_ACT_FLOATS];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5;
    const int64_t tile = blockIdx.x;
    const int64_t ray = tile * R2L_TILE_RAYS + (lane & 31);
    const bool valid = ray < a.N;
    const int64_t rc = valid ? ray : a.N - 1;
    const int64_t Np = R2L_PAD_ROWS(a.N);

    CRing<8> r8;
    r8.init(a.wstream, 0, lane, wave);

    float dp[3], se = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float r = a.rgb[rc * 3 + c];
        float dl;
        if (a.target != nullptr) {
            const float e = r - a.target[rc * 3 + c];
            se += e * e;
            dl = a.grad_scale * e;
        } else {
            dl = a.drgb[rc * 3 + c];
        }
        dp[c] = valid ? dl * (r * (1.0f - r)) : 0.f;
    }
    if (!valid) se = 0.f;
    if (wave == 0) {
        if (valid && h == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) a.dpre[ray * 3 + c] = dp[c];
        }
        if (a.sqerr_partial != nullptr) {
            float s = (h == 0) ? se : 0.f;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
            if (lane == 0) a.sqerr_partial[tile] = s;
        }
    }
    // own slice of g = dy = Wt^T dpre, kept also as dy for the outer-residual branch at the head
    f32x16 g[2], dy[2], u[2];
    {
        const float* tw = a.params + c_off_tail_w(a.n_block) + 64 * wave + 4 * h;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 wv[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) wv[c] = *reinterpret_cast<const f32x4*>(tw + c * R2L_W + 32 * t + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = wv[0][e] * dp[0];
                    v = __builtin_fmaf(wv[1][e], dp[1], v);
                    v = __builtin_fmaf(wv[2][e], dp[2], v);
                    g[t][4 * q + e] = v;
                    dy[t][4 * q + e] = v;
                }
            }
    }
    lds_write_slice(act[0], lane, wave, g);
    const int64_t tile_row0 = tile * R2L_TILE_RAYS;

    f32x16 bop[R2L_NT];
#pragma unroll 1
    for (int b = a.n_block - 1; b >= 0; --b) {
        // u = (W2^T g) * [t_b > 0]
        __syncthreads();
        lds_read_bops(act[0], lane, bop);
        rows_lds_to_global(act[0], a.gx + (int64_t)(b + 1) * Np * R2L_W, tile_row0, lane, wave);  // g = dL/dx_{b+1}
        f32x16 tm[2];
        g_load_slice(a.save_t + (int64_t)b * Np * R2L_W, ray, lane, wave, tm);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 16; ++c) u[t][c] = 0.f;
        clayer<false>(u, bop, r8);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 16; ++c) u[t][c] = tm[t][c] > 0.f ? u[t][c] : 0.f;
        lds_write_slice(act[1], lane, wave, u);
        // g += W1^T u
        __syncthreads();
        lds_read_bops(act[1], lane, bop);
        rows_lds_to_global(act[1], a.gt + (int64_t)b * Np * R2L_W, tile_row0, lane, wave);  // u_b
        clayer<false>(g, bop, r8);
        lds_write_slice(act[0], lane, wave, g);
    }
    // head: dL/d(head pre-activation) = (g + dy) * (x_0 > 0)
    f32x16 xm[2];
    g_load_slice(a.save_x, ray, lane, wave, xm);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 16; ++c) g[t][c] = xm[t][c] > 0.f ? g[t][c] + dy[t][c] : 0.f;
    g_store_slice(a.gx, ray, lane, wave, g);
}

// ------------------------------------------------------------------------------------------------------------------
// launchers used by the C ABI entry points in r2l_forward.hip / r2l_backward.hip
// ------------------------------------------------------------------------------------------------------------------
int r2l_coop_forward(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab,
                     const float* c2w_host12, int H, int W, float focal, const float* wstream, const float* params,
                     int n_block, float* rgb, float* save_x, float* save_t, int64_t N, hipStream_t stream) {
    CoopFwdArgs a{};
    a.rays_o = rays_o; a.rays_d = rays_d; a.t_rand = t_rand; a.ztab = ztab; a.wstream = wstream; a.params = params;
    a.n_block = n_block; a.rgb = rgb; a.save_x = save_x; a.save_t = save_t; a.N = N; a.H = H; a.Wimg = W; a.focal = focal;
    if (c2w_host12) for (int i = 0; i < 12; ++i) a.c2w[i] = c2w_host12[i];
    const dim3 grid((unsigned)((N + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS)), block(256);
    if (c2w_host12) hipLaunchKernelGGL((r2l_fwd_coop_kernel<true, false>), grid, block, 0, stream, a);
    else if (save_x) hipLaunchKernelGGL((r2l_fwd_coop_kernel<false, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((r2l_fwd_coop_kernel<false, false>), grid, block, 0, stream, a);
    R2L_CHECK(hipGetLastError());
    return 0;
}

int r2l_coop_backward(const float* rgb, const float* target, const float* drgb, const float* save_x, const float* save_t,
                      const float* wstream_bwd, const float* params, int n_block, float grad_scale, float* dpre, float* gx,
                      float* gt, float* sqerr_partial, int64_t N, hipStream_t stream) {
    CoopBwdArgs a{};
    a.rgb = rgb; a.target = target; a.drgb = drgb; a.save_x = save_x; a.save_t = save_t; a.wstream = wstream_bwd;
    a.params = params; a.n_block = n_block; a.grad_scale = grad_scale; a.dpre = dpre; a.gx = gx; a.gt = gt;
    a.sqerr_partial = sqerr_partial; a.N = N;
    const dim3 grid((unsigned)((N + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS)), block(256);
    hipLaunchKernelGGL(r2l_bwd_coop_kernel, grid, block, 0, stream, a);
    R2L_CHECK(hipGetLastError());
    return 0;
}
