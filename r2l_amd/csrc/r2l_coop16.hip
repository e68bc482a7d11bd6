// r2l_coop16.hip — 16-ray cooperative variants of the R2L student chains (forward and dX backward) for gfx950.
//
// A 4096-ray training step (BASELINE configs[2] read literally) is 128 tiles of 32 rays: even with four waves sharing a
// tile (r2l_coop.hip) only 128 of the 256 CUs get a workgroup.  Here a workgroup owns a 16-ray tile and the GEMMs run on
// v_mfma_f32_16x16x4_f32 (same 64 FLOP/cycle/SIMD as the 32x32x2 form): 4096 rays = 256 workgroups = every CU busy.
//   * wave w owns output features [64w, 64w+64) of every layer: four 16-feature MFMA tiles, 16 accumulator registers;
//   * the activation of the tile lives in LDS ([16 rays][256+4] fp32, double buffered, one barrier per layer); a lane
//     (ray j = l%16, quarter kk = l/16) reads its B operands for k-group G as ONE ds_read_b128: features 16G+4kk..+3 —
//     the k slots of the four MFMAs of a group are numbered so that they are contiguous in memory (r2l_common.h);
//   * weights stream from the 16-layout packed stream through a ring of 8 groups (a group = 16 MFMAs = 512 cycles);
//   * the head evaluates the 1008-d encoding on the fly in a permuted k order ((sin f, cos f) pairs, coordinate-major):
//     a lane computes exactly the 252 encoding values it feeds, 2 sincos per 16 MFMAs;
//   * the bias of the next layer is prefetched while the current layer computes.
// Semantics, stash layout and gradients are identical to the other variants (tests run all three).
#include "r2l_common.h"

#define C16_RAYS 16
#define C16_LD 260                          // LDS row pitch in floats: 16-byte skew per ray -> conflict-free b128 access
#define C16_ACT_FLOATS (C16_RAYS * C16_LD)  // one activation buffer
#define C16_GROUP_BYTES (R2L_C16_GROUP_FLOATS * 4)

typedef float f32x4v __attribute__((ext_vector_type(4)));

__host__ __device__ static inline int64_t h_off_head_b() { return (int64_t)R2L_IN * R2L_W; }
__host__ __device__ static inline int64_t h_off_body_w(int layer) {
    return (int64_t)R2L_IN * R2L_W + R2L_W + (int64_t)layer * (R2L_W * R2L_W + R2L_W);
}
__host__ __device__ static inline int64_t h_off_body_b(int layer) { return h_off_body_w(layer) + R2L_W * R2L_W; }
__host__ __device__ static inline int64_t h_off_tail_w(int n_block) { return h_off_body_w(2 * n_block); }
__host__ __device__ static inline int64_t h_off_tail_b(int n_block) { return h_off_tail_w(n_block) + 3 * R2L_W; }
#define C16_BIAS_STRIDE (R2L_W * R2L_W + R2L_W)

// ---- weight ring of one wave: D groups x its 4 tiles ---------------------------------------------------------------------
template <int D>
struct Ring16 {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff;  // lane*16 + (4*wave)*1024: this wave's first tile inside a group
    unsigned soff;  // byte position of the next group to LOAD
    f32x4 w[D][4];
    __device__ __forceinline__ f32x4 load(int t) const {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (unsigned)t * 1024u, soff, 0));
    }
    __device__ __forceinline__ void opaque() { asm volatile("" : "+v"(voff)); }
    __device__ __forceinline__ void init(const float* stream, int64_t first_group, int lane, int wave) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(stream), 0, 0xffffffff, 0x00020000);
        voff = (unsigned)lane * 16u + (unsigned)(4 * wave) * 1024u;
        soff = (unsigned)(first_group * C16_GROUP_BYTES);
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int t = 0; t < 4; ++t) w[d][t] = load(t);
            soff += C16_GROUP_BYTES;
        }
    }
};

// acc[4 tiles] += W_group[own tiles] . b for the group in ring slot SLOT, then reload the slot with the group D ahead.
// MFMA order: k-step outer, tile inner (no back-to-back dependent MFMAs); each tile's reload follows its last use.
// EXTRA_RD: VMEM reads issued by the caller just before (bias prefetch) are scheduled first.  VPT: VALU slots after
// every MFMA (the head hides the evaluation of the NEXT group's sin/cos values under this group's MFMAs).
template <int SLOT, int EXTRA_RD = 0, int VPT = 0, int D>
__device__ __forceinline__ void group16(f32x4 (&acc)[4], Ring16<D>& r, const f32x4& b) {
    static_assert(SLOT >= 0 && SLOT < D, "ring slot");
    r.opaque();
#pragma unroll
    for (int e = 0; e < 3; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.w[SLOT][t][e], b[e], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.w[SLOT][t][3], b[3], acc[t], 0, 0, 0);
        r.w[SLOT][t] = r.load(t);
    }
    r.soff += C16_GROUP_BYTES;
    if (EXTRA_RD > 0) __builtin_amdgcn_sched_group_barrier(0x020, EXTRA_RD, 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (VPT > 0) __builtin_amdgcn_sched_group_barrier(0x002, VPT, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if (VPT > 0) __builtin_amdgcn_sched_group_barrier(0x002, VPT, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// One 256 -> 64 (own slice) layer: acc += W[own rows] . bop; while it runs, the bias slice of the NEXT layer is fetched.
__device__ __forceinline__ void layer16(f32x4 (&acc)[4], const f32x4 (&bop)[16], Ring16<8>& r, const float* next_bias,
                                        f32x4 (&bn)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) bn[t] = *reinterpret_cast<const f32x4*>(next_bias + 16 * t);
    group16<0, 4>(acc, r, bop[0]);
    group16<1>(acc, r, bop[1]);
    group16<2>(acc, r, bop[2]);
    group16<3>(acc, r, bop[3]);
    group16<4>(acc, r, bop[4]);
    group16<5>(acc, r, bop[5]);
    group16<6>(acc, r, bop[6]);
    group16<7>(acc, r, bop[7]);
    group16<0>(acc, r, bop[8]);
    group16<1>(acc, r, bop[9]);
    group16<2>(acc, r, bop[10]);
    group16<3>(acc, r, bop[11]);
    group16<4>(acc, r, bop[12]);
    group16<5>(acc, r, bop[13]);
    group16<6>(acc, r, bop[14]);
    group16<7>(acc, r, bop[15]);
}

// ---- LDS / global exchange of fragments ------------------------------------------------------------------------------------
// all 256 features of the tile as B operands: lane (j, kk) register G <- act[j][16G + 4kk .. +3]
__device__ __forceinline__ void lds_read_bops16(const float* act, int lane, f32x4 (&bop)[16]) {
    const float* row = act + (lane & 15) * C16_LD + 4 * (lane >> 4);
#pragma unroll
    for (int G = 0; G < 16; ++G) bop[G] = *reinterpret_cast<const f32x4*>(row + 16 * G);
}
// own 64-feature slice: D fragment of tile t = features 64w + 16t + 4kk .. +3 of ray j
__device__ __forceinline__ void lds_write_slice16(float* act, int lane, int wave, const f32x4 (&v)[4]) {
    float* row = act + (lane & 15) * C16_LD + 64 * wave + 4 * (lane >> 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(row + 16 * t) = v[t];
}
__device__ __forceinline__ void g_store_slice16(float* base, int64_t ray, int lane, int wave, const f32x4 (&v)[4]) {
    float* row = base + ray * R2L_W + 64 * wave + 4 * (lane >> 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) r2l_stash_store(row + 16 * t, v[t]);
}
__device__ __forceinline__ void g_load_slice16(const float* base, int64_t ray, int lane, int wave, f32x4 (&v)[4]) {
    const float* row = base + ray * R2L_W + 64 * wave + 4 * (lane >> 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const f32x4*>(row + 16 * t);
}
// Whole rows of the tile from the LDS activation buffer to a row-major [N][256] tensor: wave w moves rays 4w .. 4w+3,
// one 1 KiB row per instruction (64 lanes x 16 B: full cache lines, unlike the 64-byte pieces of a fragment store).
__device__ __forceinline__ void rows_lds_to_global(const float* act, float* base, int64_t tile_row0, int lane, int wave) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ray = 4 * wave + r;
        const f32x4 v = *reinterpret_cast<const f32x4*>(act + ray * C16_LD + 4 * lane);
        r2l_stash_store_nt(base + (tile_row0 + ray) * R2L_W + 4 * lane, v);
    }
}
__device__ __forceinline__ float sel4(int kk, float a0, float a1, float a2, float a3) {
    const float lo = kk == 1 ? a1 : a0, hi = kk == 3 ? a3 : a2;
    return kk >= 2 ? hi : lo;
}

// =================================================================================================================
// forward
// =================================================================================================================
struct C16FwdArgs {
    const float* rays_o;
    const float* rays_d;
    const float* t_rand;
    const float* ztab;
    float c2w[12];
    int H, Wimg;
    float focal;
    const float* wstream;  // 16-layout forward stream
    const float* params;
    int n_block;
    float* rgb;
    float* save_x;
    float* save_t;
    int64_t N;
};

// 5 trig groups (80 k' values) cover the 4 coordinates xb[0..3] of one block.  Group Gp, quarter kk: k' offset
// off = 16*Gp + 4*kk -> coordinate off/20, first frequency (off%20)/2; the lane's 4 values are
// (sin 2^f x, cos 2^f x, sin 2^(f+1) x, cos 2^(f+1) x).
template <int Gp>
__device__ __forceinline__ f32x4 trig_chunk(int kk, const float (&xb)[4]) {
    constexpr int o0 = 16 * Gp, o1 = o0 + 4, o2 = o0 + 8, o3 = o0 + 12;
    const float x = sel4(kk, xb[o0 / 20], xb[o1 / 20], xb[o2 / 20], xb[o3 / 20]);
    const float sc = sel4(kk, (float)(1 << ((o0 % 20) / 2)), (float)(1 << ((o1 % 20) / 2)), (float)(1 << ((o2 % 20) / 2)),
                          (float)(1 << ((o3 % 20) / 2)));
    f32x4 v;
    float s, c;
    r2l_sincos(x * sc, s, c);
    v[0] = s;
    v[1] = c;
    r2l_sincos(x * (sc + sc), s, c);
    v[2] = s;
    v[3] = c;
    return v;
}

template <bool POSE, bool SAVE>
__global__ __launch_bounds__(256, 1) void r2l_fwd_c16_kernel(const C16FwdArgs a) {
    __shared__ __attribute__((aligned(16))) float act[2][C16_ACT_FLOATS];
    __shared__ float tailred[4][C16_RAYS][4];
    __shared__ float zl[4][C16_RAYS][20];  // per-wave copy of the sample depths (indexed by the runtime sample number)

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, kk = lane >> 4;
    const int64_t tile_row0 = (int64_t)blockIdx.x * C16_RAYS;
    const int64_t ray = tile_row0 + j;  // < r2l_padded_rows(N): padding rows are computed too
    const bool valid = ray < a.N;
    const int64_t rc = valid ? ray : a.N - 1;
    const int64_t Np = R2L_PAD_ROWS(a.N);

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
    float z[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(a.ztab + 4 * q);
#pragma unroll
        for (int k = 0; k < 4; ++k) z[4 * q + k] = lo[k];
        if (a.t_rand != nullptr) {
            const f32x4 sp = *reinterpret_cast<const f32x4*>(a.ztab + 16 + 4 * q);
            const f32x4 u = *reinterpret_cast<const f32x4*>(a.t_rand + rc * 16 + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) z[4 * q + k] = lo[k] + sp[k] * u[k];
        }
    }

    // ---- head ---------------------------------------------------------------------------------------------------------
    f32x4 acc[4], xo[4], x0[4], bn[4];
    {
        const float* hb = a.params + h_off_head_b() + 64 * wave + 4 * kk;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = *reinterpret_cast<const f32x4*>(hb + 16 * t);
    }
    Ring16<5> r5;  // 15 groups per loop trip: a ring of 5 keeps the slot numbers compile-time
    r5.init(a.wstream, 0, lane, wave);
    // coordinates 12S .. 12S+11 (samples 4S .. 4S+3) of this lane's ray
    // (the four lanes of a ray hold the same z: they write the same values; a lane reads back only its own row)
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(&zl[wave][j][4 * q]) = f32x4{z[4 * q], z[4 * q + 1], z[4 * q + 2], z[4 * q + 3]};
    auto coords = [&](int S, float (&xc)[12]) {
        const f32x4 zs = *reinterpret_cast<const f32x4*>(&zl[wave][j][4 * (S & 3)]);
#pragma unroll
        for (int i = 0; i < 12; ++i) xc[i] = o[i % 3] + d[i % 3] * zs[i / 3];
    };
    {
        float xc[12], xn[12];
        coords(0, xc);
        f32x4 cur, nxt;
        {
            const float xb[4] = {xc[0], xc[1], xc[2], xc[3]};
            cur = trig_chunk<0>(kk, xb);
        }
        // software pipeline: the region of group n evaluates the sin/cos chunk of group n+1 (VALU) under its own MFMAs
#define C16_HEAD_STEP(SLOT, NEXT_GP, XB)                   \
    nxt = trig_chunk<NEXT_GP>(kk, XB);                     \
    group16<SLOT, 0, 6>(acc, r5, cur);                     \
    cur = nxt;
#pragma unroll 1
        for (int S = 0; S < 4; ++S) {  // three blocks of 4 coordinates = 15 groups per trip
            coords(S + 1, xn);         // S == 3: unused values (the identity groups follow)
            const float b0[4] = {xc[0], xc[1], xc[2], xc[3]}, b1[4] = {xc[4], xc[5], xc[6], xc[7]};
            const float b2[4] = {xc[8], xc[9], xc[10], xc[11]}, b3[4] = {xn[0], xn[1], xn[2], xn[3]};
            C16_HEAD_STEP(0, 1, b0) C16_HEAD_STEP(1, 2, b0) C16_HEAD_STEP(2, 3, b0) C16_HEAD_STEP(3, 4, b0)
            C16_HEAD_STEP(4, 0, b1)
            C16_HEAD_STEP(0, 1, b1) C16_HEAD_STEP(1, 2, b1) C16_HEAD_STEP(2, 3, b1) C16_HEAD_STEP(3, 4, b1)
            C16_HEAD_STEP(4, 0, b2)
            C16_HEAD_STEP(0, 1, b2) C16_HEAD_STEP(1, 2, b2) C16_HEAD_STEP(2, 3, b2) C16_HEAD_STEP(3, 4, b2)
            C16_HEAD_STEP(4, 0, b3)
#pragma unroll
            for (int i = 0; i < 12; ++i) xc[i] = xn[i];
        }
#undef C16_HEAD_STEP
    }
    {
        // identity groups: k' = 960 + c, c = 16g + 4kk + e: the point coordinate c itself
        float p[48];
#pragma unroll
        for (int c = 0; c < 48; ++c) p[c] = o[c % 3] + d[c % 3] * z[c / 3];
        f32x4 idv[3];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) idv[g][e] = sel4(kk, p[16 * g + e], p[16 * g + 4 + e], p[16 * g + 8 + e], p[16 * g + 12 + e]);
        group16<0>(acc, r5, idv[0]);
        group16<1>(acc, r5, idv[1]);
        group16<2>(acc, r5, idv[2]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            xo[t][c] = fmaxf(acc[t][c], 0.f);  // X_0 = relu(head)
            x0[t][c] = xo[t][c];
        }
    lds_write_slice16(act[0], lane, wave, xo);
    // widen the ring to 8: r5 holds stream groups 63..67 (= body layer 0, groups 0..4) in slots 3, 4, 0, 1, 2
    Ring16<8> r8;
    r8.rsrc = r5.rsrc;
    r8.voff = r5.voff;
    r8.soff = r5.soff;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        r8.w[0][t] = r5.w[3][t];
        r8.w[1][t] = r5.w[4][t];
        r8.w[2][t] = r5.w[0][t];
        r8.w[3][t] = r5.w[1][t];
        r8.w[4][t] = r5.w[2][t];
    }
#pragma unroll
    for (int dd = 5; dd < 8; ++dd) {
#pragma unroll
        for (int t = 0; t < 4; ++t) r8.w[dd][t] = r8.load(t);
        r8.soff += C16_GROUP_BYTES;
    }

    // ---- body -----------------------------------------------------------------------------------------------------------
    f32x4 bop[16];
    const float* bias = a.params + h_off_body_b(0) + 64 * wave + 4 * kk;  // own slice of the bias of layer (b, 0)
#pragma unroll
    for (int t = 0; t < 4; ++t) bn[t] = *reinterpret_cast<const f32x4*>(bias + 16 * t);
#pragma unroll 1
    for (int b = 0; b < a.n_block; ++b) {
        // t = relu(W1 x + b1): B operands = x from act[0]
        __syncthreads();
        lds_read_bops16(act[0], lane, bop);
        if constexpr (SAVE) rows_lds_to_global(act[0], a.save_x + (int64_t)b * Np * R2L_W, tile_row0, lane, wave);  // x_b
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = bn[t];
        layer16(acc, bop, r8, bias + C16_BIAS_STRIDE, bn);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[t][c] = fmaxf(acc[t][c], 0.f);
        lds_write_slice16(act[1], lane, wave, acc);
        // x += W2 t + b2: B operands = t from act[1]
        __syncthreads();
        lds_read_bops16(act[1], lane, bop);
        if constexpr (SAVE) rows_lds_to_global(act[1], a.save_t + (int64_t)b * Np * R2L_W, tile_row0, lane, wave);  // t_b
#pragma unroll
        for (int t = 0; t < 4; ++t) xo[t] += bn[t];
        // prefetch b1 of the next block (behind the last block: re-read the current one, never used)
        layer16(xo, bop, r8, b + 1 < a.n_block ? bias + 2 * C16_BIAS_STRIDE : bias, bn);
        lds_write_slice16(act[0], lane, wave, xo);
        bias += 2 * C16_BIAS_STRIDE;
    }
    if constexpr (SAVE) {  // x_n (the loop stored x_0 .. x_{n-1} at the top of each trip)
        __syncthreads();
        rows_lds_to_global(act[0], a.save_x + (int64_t)a.n_block * Np * R2L_W, tile_row0, lane, wave);
    }

    // ---- tail: rgb = sigmoid(Wt (x + X_0) + bt): partial dots per lane, reduced over kk (shuffles) and waves (LDS) -------
    const float* tw = a.params + h_off_tail_w(a.n_block) + 64 * wave + 4 * kk;
    float p3[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x4 wv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) wv[c] = *reinterpret_cast<const f32x4*>(tw + c * R2L_W + 16 * t);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float y = xo[t][e] + x0[t][e];
#pragma unroll
            for (int c = 0; c < 3; ++c) p3[c] = __builtin_fmaf(wv[c][e], y, p3[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        p3[c] += __shfl_xor(p3[c], 16);
        p3[c] += __shfl_xor(p3[c], 32);
    }
    if (kk == 0) {
        tailred[wave][j][0] = p3[0];
        tailred[wave][j][1] = p3[1];
        tailred[wave][j][2] = p3[2];
    }
    __syncthreads();
    if (wave == 0 && kk == 0 && valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = ((tailred[0][j][c] + tailred[1][j][c]) + (tailred[2][j][c] + tailred[3][j][c])) +
                            a.params[h_off_tail_b(a.n_block) + c];
            a.rgb[ray * 3 + c] = 1.0f / (1.0f + expf(-v));
        }
    }
}

// =================================================================================================================
// backward (dX chain)
// =================================================================================================================
struct C16BwdArgs {
    const float* rgb;
    const float* target;
    const float* drgb;
    const float* save_x;
    const float* save_t;
    const float* wstream;  // 16-layout transposed stream
    const float* params;
    int n_block;
    float grad_scale;
    float* dpre;
    float* gx;
    float* gt;
    float* sqerr_partial;
    int64_t N;
};

// same GEMM without the bias prefetch
__device__ __forceinline__ void layer16_plain(f32x4 (&acc)[4], const f32x4 (&bop)[16], Ring16<8>& r) {
    group16<0>(acc, r, bop[0]);
    group16<1>(acc, r, bop[1]);
    group16<2>(acc, r, bop[2]);
    group16<3>(acc, r, bop[3]);
    group16<4>(acc, r, bop[4]);
    group16<5>(acc, r, bop[5]);
    group16<6>(acc, r, bop[6]);
    group16<7>(acc, r, bop[7]);
    group16<0>(acc, r, bop[8]);
    group16<1>(acc, r, bop[9]);
    group16<2>(acc, r, bop[10]);
    group16<3>(acc, r, bop[11]);
    group16<4>(acc, r, bop[12]);
    group16<5>(acc, r, bop[13]);
    group16<6>(acc, r, bop[14]);
    group16<7>(acc, r, bop[15]);
}

__global__ __launch_bounds__(256, 1) void r2l_bwd_c16_kernel(const C16BwdArgs a) {
    __shared__ __attribute__((aligned(16))) float act[2][C16_ACT_FLOATS];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, kk = lane >> 4;
    const int64_t tile = blockIdx.x;
    const int64_t ray = tile * C16_RAYS + j;
    const bool valid = ray < a.N;
    const int64_t rc = valid ? ray : a.N - 1;
    const int64_t Np = R2L_PAD_ROWS(a.N);

    Ring16<8> r8;
    r8.init(a.wstream, 0, lane, wave);

    float dp[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float r = a.rgb[rc * 3 + c];
        const float dl = a.target != nullptr ? a.grad_scale * (r - a.target[rc * 3 + c]) : a.drgb[rc * 3 + c];
        dp[c] = valid ? dl * (r * (1.0f - r)) : 0.f;
    }
    if (wave == 0 && kk == 0 && valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.dpre[ray * 3 + c] = dp[c];
    }
    // squared-error partial of the 32-ray tile this 16-ray tile is the first half of (same slots as the other variants)
    if (a.sqerr_partial != nullptr && (tile & 1) == 0 && wave == 1) {
        const int64_t r32 = tile * C16_RAYS + (lane & 31);
        float s = 0.f;
        if (a.target != nullptr && lane < 32 && r32 < a.N) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float e = a.rgb[r32 * 3 + c] - a.target[r32 * 3 + c];
                s += e * e;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) a.sqerr_partial[tile >> 1] = s;
    }
    // own slice of g = dy = Wt^T dpre, kept also as dy for the outer-residual branch at the head
    f32x4 g[4], dy[4], u[4];
    {
        const float* tw = a.params + h_off_tail_w(a.n_block) + 64 * wave + 4 * kk;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 wv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) wv[c] = *reinterpret_cast<const f32x4*>(tw + c * R2L_W + 16 * t);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = wv[0][e] * dp[0];
                v = __builtin_fmaf(wv[1][e], dp[1], v);
                v = __builtin_fmaf(wv[2][e], dp[2], v);
                g[t][e] = v;
                dy[t][e] = v;
            }
        }
    }
    lds_write_slice16(act[0], lane, wave, g);

    f32x4 bop[16];
    const int64_t tile_row0 = tile * C16_RAYS;
#pragma unroll 1
    for (int b = a.n_block - 1; b >= 0; --b) {
        // u = (W2^T g) * [t_b > 0]
        __syncthreads();
        lds_read_bops16(act[0], lane, bop);
        rows_lds_to_global(act[0], a.gx + (int64_t)(b + 1) * Np * R2L_W, tile_row0, lane, wave);  // g = dL/dx_{b+1}
        f32x4 tm[4];
        g_load_slice16(a.save_t + (int64_t)b * Np * R2L_W, ray, lane, wave, tm);
#pragma unroll
        for (int t = 0; t < 4; ++t) u[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        layer16_plain(u, bop, r8);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) u[t][c] = tm[t][c] > 0.f ? u[t][c] : 0.f;
        lds_write_slice16(act[1], lane, wave, u);
        // g += W1^T u
        __syncthreads();
        lds_read_bops16(act[1], lane, bop);
        rows_lds_to_global(act[1], a.gt + (int64_t)b * Np * R2L_W, tile_row0, lane, wave);  // u_b
        layer16_plain(g, bop, r8);
        lds_write_slice16(act[0], lane, wave, g);
    }
    // head: dL/d(head pre-activation) = (g + dy) * (x_0 > 0)
    f32x4 xm[4];
    g_load_slice16(a.save_x, ray, lane, wave, xm);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) g[t][c] = xm[t][c] > 0.f ? g[t][c] + dy[t][c] : 0.f;
    g_store_slice16(a.gx, ray, lane, wave, g);
}

// ------------------------------------------------------------------------------------------------------------------
// launchers used by the C ABI entry points in r2l_forward.hip / r2l_backward.hip
// ------------------------------------------------------------------------------------------------------------------
int r2l_coop16_forward(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab,
                       const float* c2w_host12, int H, int W, float focal, const float* wstream16, const float* params,
                       int n_block, float* rgb, float* save_x, float* save_t, int64_t N, hipStream_t stream) {
    C16FwdArgs a{};
    a.rays_o = rays_o; a.rays_d = rays_d; a.t_rand = t_rand; a.ztab = ztab; a.wstream = wstream16; a.params = params;
    a.n_block = n_block; a.rgb = rgb; a.save_x = save_x; a.save_t = save_t; a.N = N; a.H = H; a.Wimg = W; a.focal = focal;
    if (c2w_host12) for (int i = 0; i < 12; ++i) a.c2w[i] = c2w_host12[i];
    // every padded row is computed: the dW kernels read whole 32-row tiles of the stash
    const dim3 grid((unsigned)(R2L_PAD_ROWS(N) / C16_RAYS)), block(256);
    if (c2w_host12) hipLaunchKernelGGL((r2l_fwd_c16_kernel<true, false>), grid, block, 0, stream, a);
    else if (save_x) hipLaunchKernelGGL((r2l_fwd_c16_kernel<false, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((r2l_fwd_c16_kernel<false, false>), grid, block, 0, stream, a);
    R2L_CHECK(hipGetLastError());
    return 0;
}

int r2l_coop16_backward(const float* rgb, const float* target, const float* drgb, const float* save_x,
                        const float* save_t, const float* wstream_bwd16, const float* params, int n_block,
                        float grad_scale, float* dpre, float* gx, float* gt, float* sqerr_partial, int64_t N,
                        hipStream_t stream) {
    C16BwdArgs a{};
    a.rgb = rgb; a.target = target; a.drgb = drgb; a.save_x = save_x; a.save_t = save_t; a.wstream = wstream_bwd16;
    a.params = params; a.n_block = n_block; a.grad_scale = grad_scale; a.dpre = dpre; a.gx = gx; a.gt = gt;
    a.sqerr_partial = sqerr_partial; a.N = N;
    const dim3 grid((unsigned)(R2L_PAD_ROWS(N) / C16_RAYS)), block(256);
    hipLaunchKernelGGL(r2l_bwd_c16_kernel, grid, block, 0, stream, a);
    R2L_CHECK(hipGetLastError());
    return 0;
}
