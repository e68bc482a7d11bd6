// r2l_backward.hip — hand-written backward of the R2L student for gfx950 (replaces the autograd graph the reference
// builds at /root/reference/main.py:1374-1404: loss = mean((rgb-target)^2) * lw_rgb, loss.backward()).
//
//   1. r2l_bwd_chain_kernel : dL/drgb -> sigmoid' -> tail^T -> the 2*n_block transposed layers, register-resident
//                             exactly like the forward (W^T streams packed by r2l_pack_backward); writes the
//                             per-layer output gradients G (row-major [N,256]) that the weight-gradient GEMMs need.
//   2. r2l_dw_body_kernel   : dW[l] = G[l]^T A[l] (reduction over rays) for the 2*n_block 256x256 layers + db,
//                             fp32 MFMA, operands streamed straight from HBM with coalesced 16-byte loads.
//   3. r2l_dw_head_kernel   : dW_head = G_head^T PE(rays) with the 1008-d positional encoding recomputed on the fly.
//   4. r2l_dw_tail_kernel   : tail weight/bias gradients (3x256) by plain reduction.
#include "r2l_common.h"
#include "r2l_f2.h"
#include "r2l_dw.h"
#include "r2l_hip.h"

struct R2LBwdArgs {
    const float* rgb;      // [N,3] forward output
    const float* target;   // [N,3]  MSE mode: dL/drgb = grad_scale*(rgb-target)            (or nullptr)
    const float* drgb;     // [N,3]  generic mode (target == nullptr): dL/drgb given by the caller
    const float* save_x;   // [(n_block+1),N,256]
    const float* save_t;   // [n_block,N,256]
    const float* wstream;  // packed transposed weight stream
    const float* params;   // flat params (tail weights)
    int n_block;
    float grad_scale;      // dL/drgb = grad_scale * (rgb - target);  = 2*lw_rgb / (3*N_global)
    float* dpre;           // [N,3]   dL/d(tail pre-activation)
    float* gx;             // [(n_block+1),N,256]  gx[b] = dL/dx_b ; gx[0] already includes the outer-residual branch and
                           //                      is masked by relu'(head) i.e. it is dL/d(head pre-activation)
    float* gt;             // [n_block,N,256]      dL/d(hidden pre-activation) of each block
    float* sqerr_partial;  // [ceil(N/32)] per-tile sums of (rgb-target)^2
    int64_t N;
};

// relu'(t) as 128 bits per lane: bit R2L_MASK32_BIT(T, c) of word T >> 1 belongs to fragment register (T, c) (r2l_common.h: the
// forward's mask words)
struct MaskAct {
    unsigned mb[4];
    __device__ __forceinline__ float operator()(float v, int T, int c) const {
        return ((mb[T >> 1] >> R2L_MASK32_BIT(T, c)) & 1u) ? v : 0.f;
    }
};

// GEMM A of the backward chain (u = W2^T g): g = dL/dx_{b+1} is the B operand, its store to gx[b+1] rides along (StoreHook).
// The ReLU mask of the block's hidden activation comes from the forward's mask words (r2l_common.h r2l_mask32_offset): one
// 16-byte load per lane, issued in front of the GEMM and first used behind it.  (Rounds 1 - 5 prefetched relu(t) itself, one
// 16-byte piece per group, and folded its signs: 32 loads per lane and block, all of save_t read a second time.)

__global__ __launch_bounds__(256, 1) void r2l_bwd_chain_kernel(const R2LBwdArgs a) {
    __shared__ float stash[4][R2L_NT * 16][64];  // dy of each wave's tile (outer residual branch), re-added at the head

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile * R2L_TILE_RAYS >= a.N) return;
    const int64_t ray = tile * R2L_TILE_RAYS + (lane & 31);
    const bool valid = ray < a.N;
    const int64_t rc = valid ? ray : a.N - 1;

    WRing2 ws;  // 64 groups per block: ring slots are static (r2l_common.h)
    ws.init(a.wstream, lane);
    const int64_t Np = R2L_PAD_ROWS(a.N);  // rows per stash / gradient slot

    // loss gradient through the sigmoid: dpre = grad_scale*(rgb-target) * rgb*(1-rgb)
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
    if (valid && h == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.dpre[ray * 3 + c] = dp[c];
    }
    // per-tile squared error (lanes 0..31 hold the 32 rays)
    if (a.sqerr_partial != nullptr) {
        float s = (h == 0) ? se : 0.f;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) a.sqerr_partial[tile] = s;
    }

    // g = dy = Wt^T dpre   (tail Linear(256,3))
    f32x16 g[R2L_NT], u[R2L_NT];
    const float* tw = a.params + b_off_tail_w(a.n_block);
#pragma unroll
    for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 wv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) wv[c] = *reinterpret_cast<const f32x4*>(tw + c * R2L_W + 32 * T + 8 * q + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = wv[0][j] * dp[0];
                v = __builtin_fmaf(wv[1][j], dp[1], v);
                v = __builtin_fmaf(wv[2][j], dp[2], v);
                g[T][4 * q + j] = v;
            }
        }
#pragma unroll
    for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
        for (int c = 0; c < 16; ++c) stash[wave][T * 16 + c][lane] = g[T][c];
#pragma unroll 1
    for (int b = a.n_block - 1; b >= 0; --b) {
        // u = W2^T g (accumulators initialised by the first group, C = 0); g (= dL/dx_{b+1}) is stored to gx[b+1] along the way;
        // the ReLU mask words of the block's hidden activation are requested in front of it
        const u32x4 mw = *reinterpret_cast<const u32x4*>(a.save_t + r2l_mask32_offset(a.n_block, Np, b) + tile * 256 + lane * 4);
        {
            StoreHook hk(a.gx + (int64_t)(b + 1) * Np * R2L_W, ray, h, g);
            gemm256a<IdentityAct, 0, true>(u, g, ws, hk, IdentityAct());
        }
        const unsigned mb[4] = {mw[0], mw[1], mw[2], mw[3]};
        // g += W1^T (u . mask): the mask relu'(hidden) = (t_b > 0) is applied lazily to the B operands of each group and
        // to the pieces of u (= dL/d hidden pre-activation) stored to gt[b] along the way
        {
            const MaskAct mask{{mb[0], mb[1], mb[2], mb[3]}};
            StoreHookT<false, MaskAct> su(a.gt + (int64_t)b * Np * R2L_W, ray, h, u, mask);
            gemm256a<MaskAct, 0>(g, u, ws, su, mask);
        }
    }

    // head: dL/d(head pre-activation) = (g + dy) * (x_0 > 0)
    {
        const float* r = a.save_x + ray * R2L_W + 4 * h;
#pragma unroll
        for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(r + 32 * T + 8 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = g[T][4 * q + j] + stash[wave][T * 16 + 4 * q + j][lane];
                    g[T][4 * q + j] = xv[j] > 0.f ? v : 0.f;
                }
            }
    }
    store_frag(a.gx, ray, h, g);
}

// =================================================================================================================
// Weight gradients of the 2*n_block body layers:   dW[l][o][i] = sum_r G_l[r][o] * A_l[r][i],   db[l][o] = sum_r G_l[r][o]
//   layer (b,0): G = gt[b]   , A = x_b    ;   layer (b,2): G = gx[b+1] , A = t_b
// One workgroup (4 waves, one per SIMD) accumulates a full 256x256 tile set: wave (wo,wi) owns output rows
// wo*128 + 4*i + e (e = 0..3: four 32-row MFMA tiles) x input columns wi*128 + 4*i' + e'.  An MFMA k-step consumes two
// rays (lanes 0-31 ray 2s, lanes 32-63 ray 2s+1); each lane feeds its 4 tiles from ONE 16-byte load per operand:
// 512 contiguous bytes per half-wave.  The (layer, ray-chunk) work list is cut into equal contiguous ranges, one per
// workgroup.  A range touches at most two layers; the workgroup stores its partial (dW, db) of each with plain coalesced
// stores into its own slab slots and r2l_dw_reduce_kernel adds the partials of a layer to the gradient in workgroup
// order: deterministic, and ~0.2 ms cheaper per step than the 65 536 scattered fp32 atomics per flush it replaced
// (which remain as the dw_slab == NULL path).  The gradient buffer is zeroed by the caller (accumulation for free).
// =================================================================================================================

__device__ __forceinline__ void dw_flush(f32x16 (&acc)[4][4], f32x4& bsum, float* __restrict__ gw, float* __restrict__ gb,
                                         int wo, int wi, int lane) {
    const int jl = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int eo = 0; eo < 4; ++eo)
#pragma unroll
        for (int ei = 0; ei < 4; ++ei)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                // D[row][col]: row = (c&3) + 8*(c>>2) + 4*hh -> output feature index within the tile, col = jl -> input
                const int ro = (c & 3) + 8 * (c >> 2) + 4 * hh;
                const int o = wo * 128 + 4 * ro + eo;
                const int i = wi * 128 + 4 * jl + ei;
                atomicAdd(gw + o * R2L_W + i, acc[eo][ei][c]);
                acc[eo][ei][c] = 0.f;
            }
    if (wi == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float s = bsum[e] + __shfl_xor(bsum[e], 32);
            if (hh == 0) atomicAdd(gb + wo * 128 + 4 * jl + e, s);
        }
    }
    bsum = f32x4{0.f, 0.f, 0.f, 0.f};
}

// Same partial tile, written with plain coalesced 16-byte stores into this workgroup's slab slot (every element exactly
// once: lanes (jl, hh) of wave (wo, wi) own rows wo*128 + 4*ro + eo, columns wi*128 + 4*jl .. +3).
__device__ __forceinline__ void dw_flush_slab(f32x16 (&acc)[4][4], f32x4& bsum, float* __restrict__ sl, int wo, int wi,
                                              int lane) {
    const int jl = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int eo = 0; eo < 4; ++eo)
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int ro = (c & 3) + 8 * (c >> 2) + 4 * hh;
            const int o = wo * 128 + 4 * ro + eo;
            const f32x4 v = {acc[eo][0][c], acc[eo][1][c], acc[eo][2][c], acc[eo][3][c]};
            *reinterpret_cast<f32x4*>(sl + o * R2L_W + wi * 128 + 4 * jl) = v;
#pragma unroll
            for (int ei = 0; ei < 4; ++ei) acc[eo][ei][c] = 0.f;
        }
    if (wi == 0) {
        f32x4 sv;
#pragma unroll
        for (int e = 0; e < 4; ++e) sv[e] = bsum[e] + __shfl_xor(bsum[e], 32);
        if (hh == 0) *reinterpret_cast<f32x4*>(sl + R2L_W * R2L_W + wo * 128 + 4 * jl) = sv;
    }
    bsum = f32x4{0.f, 0.f, 0.f, 0.f};
}

// grads[layer] += sum of the workgroup partials of that layer, added in workgroup order (deterministic).  Workgroup w
// covered units [w*upw, (w+1)*upw): its first layer is (w*upw)/upl and a layer's partial sits in slot layer - first.
__global__ __launch_bounds__(256) void r2l_dw_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grads,
                                                            int64_t upl, int64_t upw, int64_t wgs, int layer0) {
    const int layer = blockIdx.y;  // relative to layer0, like the work list
    const int i4 = blockIdx.x * 256 + threadIdx.x;
    if (i4 >= DW_SLAB_FLOATS / 4) return;
    const int64_t w0 = ((int64_t)layer * upl) / upw;
    int64_t w1 = ((int64_t)(layer + 1) * upl - 1) / upw;
    if (w1 > wgs - 1) w1 = wgs - 1;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int64_t w = w0; w <= w1; ++w) {
        const int slot = layer - (int)((w * upw) / upl);
        s += *reinterpret_cast<const f32x4*>(slab + (w * 2 + slot) * (int64_t)DW_SLAB_FLOATS + 4 * i4);
    }
    f32x4* g = reinterpret_cast<f32x4*>(grads + b_off_body_w(layer0 + layer)) + i4;
    *g = *g + s;
}

__global__ __launch_bounds__(256, 1) void r2l_dw_body_kernel(const R2LDwArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wo = wave >> 1, wi = wave & 1;
    const int hh = lane >> 5, jl = lane & 31;
    const int64_t total = a.units_per_layer * a.n_layers;
    int64_t u0 = (int64_t)blockIdx.x * a.units_per_wg;
    int64_t u1 = u0 + a.units_per_wg;
    if (u1 > total) u1 = total;
    if (u0 >= u1) return;

    f32x16 acc[4][4];
#pragma unroll
    for (int eo = 0; eo < 4; ++eo)
#pragma unroll
        for (int ei = 0; ei < 4; ++ei)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[eo][ei][c] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    int64_t u = u0;
    const int first_layer = a.layer0 + (int)(u0 / a.units_per_layer);
    while (u < u1) {
        const int layer = a.layer0 + (int)(u / a.units_per_layer);
        const int64_t cu = u % a.units_per_layer;
        int64_t cend = cu + (u1 - u);
        if (cend > a.units_per_layer) cend = a.units_per_layer;
        const int b = layer >> 1;
        const int64_t Np = R2L_PAD_ROWS(a.N);
        const float* G = (layer & 1) ? a.gx + (int64_t)(b + 1) * Np * R2L_W : a.gt + (int64_t)b * Np * R2L_W;
        const float* A = (layer & 1) ? a.save_t + (int64_t)b * Np * R2L_W : a.save_x + (int64_t)b * Np * R2L_W;
        const int64_t r0 = cu * DW_CHUNK;
        int64_t r1 = cend * DW_CHUNK;
        if (r1 > a.N) r1 = a.N;
        const float* gp = G + (int64_t)wo * 128 + 4 * jl;
        const float* ap = A + (int64_t)wi * 128 + 4 * jl;
        // Software pipeline without predicates: operands of k-step s+2 are loaded (row index clamped into the
        // segment, so the loads are unconditional and hipcc can use counted vmcnt waits) while k-step s computes.
        const int64_t nfull = (r1 - r0) / 2;  // k-steps with both rays present
        // operands through buffer descriptors based at the segment start (SGPR base + 32-bit offsets: the 64-bit VGPR
        // address form costs MFMA issue slots); per-lane part in voff, the k-step position in the scalar offset
        const __amdgpu_buffer_rsrc_t grs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G + r0 * R2L_W), 0, 0xffffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t ars =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + r0 * R2L_W), 0, 0xffffffff, 0x00020000);
        const unsigned gvo = (unsigned)(hh * R2L_W + wo * 128 + 4 * jl) * 4u;
        const unsigned avo = (unsigned)(hh * R2L_W + wi * 128 + 4 * jl) * 4u;
        auto ld = [&](int64_t s, f32x4& gv, f32x4& av) {
            const int64_t sc = s < nfull ? s : (nfull > 0 ? nfull - 1 : 0);
            const unsigned so = (unsigned)sc * (2u * R2L_W * 4u);
            gv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(grs, gvo, so, 0));
            av = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, avo, so, 0));
        };
        auto kstep = [&](const f32x4& gv, const f32x4& av) {
#pragma unroll
            for (int eo = 0; eo < 4; ++eo)
#pragma unroll
                for (int ei = 0; ei < 4; ++ei)
                    acc[eo][ei] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv[eo], av[ei], acc[eo][ei], 0, 0, 0);
            bsum += gv;
        };
        if (nfull > 0) {
            // 4 rotating operand buffers, each reloaded right after the k-step that consumed it: every load is issued
            // three k-steps (3072 MFMA cycles) before its use, with no register copies.  The loop body is a whole
            // 64-ray chunk (32 k-steps) because hipcc drains vmcnt to 0 at every loop header.
            f32x4 gb[DW_DEPTH], ab[DW_DEPTH];
#pragma unroll
            for (int k = 0; k < DW_DEPTH; ++k) ld(k, gb[k], ab[k]);
#ifdef DW_SCHED_V1  // rounds 1 - 5 (A/B builds): the buffer a k-step consumed is refilled BEHIND its 16 MFMAs
            int64_t s = 0;
            // hipcc drains vmcnt to 0 at every loop header, which exposes the full HBM latency of the newest load (~2 us):
            // long trips amortise it (one drain per 2048 / 1024 / 512 MFMAs)
#define DW_TRIPS(TRIP)                                                                       \
            for (; s + TRIP <= nfull; s += TRIP) {                                           \
                _Pragma("unroll") for (int k = 0; k < TRIP; ++k) {                           \
                    kstep(gb[k % DW_DEPTH], ab[k % DW_DEPTH]);                               \
                    ld(s + k + DW_DEPTH, gb[k % DW_DEPTH], ab[k % DW_DEPTH]);                \
                    __builtin_amdgcn_sched_barrier(0);                                       \
                }                                                                            \
            }
#else
            // Round 6.  With one wave per SIMD nothing else fills the issue port, and the round-5 loop put a k-step's two loads, its
            // clamp (a 64-bit VALU compare + three scalar ops) and the four bias adds in a lump BEHIND its 16 MFMAs (they could not go
            // earlier: the loads overwrite the operands those MFMAs read): ~130 issue cycles against the 64 the last MFMA covers —
            // the matrix pipe idled ~11 % of the kernel (PMC: 88.9 % busy at 2.39 GHz, 1152 cycles per k-step instead of 1024).
            // Now the buffer consumed ONE K-STEP AGO is refilled under this k-step's MFMAs (same look-ahead: three k-steps), the
            // clamp is one s_min_i32; the loads issue in the shadow of the previous k-step's last MFMA, the bias adds among this one's.  (The first k-step
            // of a segment refills the last prologue buffer with the data it already holds: one redundant load pair per segment.)
            const int nfull32 = (int)nfull;
            auto ldi = [&](int s32, f32x4& gv, f32x4& av) {
                const int sc = s32 < nfull32 ? s32 : nfull32 - 1;
                const unsigned so = (unsigned)sc * (2u * R2L_W * 4u);
                gv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(grs, gvo, so, 0));
                av = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, avo, so, 0));
            };
            int s = 0;
            // (not pinned with sched_group_barrier: with [2 MFMA, 1 load] x 2 in front hipcc moved accumulators between VGPRs and AGPRs in
            // every k-step — 7000 v_accvgpr moves per 128 k-steps)
#define DW_TRIPS(TRIP)                                                                                                   \
            for (; s + TRIP <= nfull32; s += TRIP) {                                                                     \
                _Pragma("unroll") for (int k = 0; k < TRIP; ++k) {                                                       \
                    ldi(s + k - 1 + DW_DEPTH, gb[(k + DW_DEPTH - 1) % DW_DEPTH], ab[(k + DW_DEPTH - 1) % DW_DEPTH]);     \
                    kstep(gb[k % DW_DEPTH], ab[k % DW_DEPTH]);                                                           \
                    __builtin_amdgcn_sched_barrier(0);                                                                   \
                }                                                                                                        \
            }
#endif
            DW_TRIPS(DW_LONG_TRIP)
            DW_TRIPS(64)
            DW_TRIPS(32)
#undef DW_TRIPS
            for (; s < nfull; ++s) {  // remainder: only the last, partial chunk at the end of N
                f32x4 gv, av;
                ld(s, gv, av);
                kstep(gv, av);
            }
        }
        if ((r1 - r0) & 1) {  // odd tail (only at the very end of N): the second ray of the pair does not exist
            const int64_t r = r1 - 1;
            f32x4 gv = *reinterpret_cast<const f32x4*>(gp + r * R2L_W);
            f32x4 av = *reinterpret_cast<const f32x4*>(ap + r * R2L_W);
            if (hh) { gv = f32x4{0.f, 0.f, 0.f, 0.f}; av = gv; }
            kstep(gv, av);
        }
#ifdef DW_NO_FLUSH  // diagnostics build only
        if (a.N < 0) dw_flush_slab(acc, bsum, a.slab, wo, wi, lane);
#else
        if (a.slab != nullptr) {
            dw_flush_slab(acc, bsum, a.slab + ((int64_t)blockIdx.x * 2 + (layer - first_layer)) * DW_SLAB_FLOATS, wo, wi,
                          lane);
        } else {
            float* gw = a.grads + b_off_body_w(layer);
            float* gb = a.grads + b_off_body_b(layer);
            dw_flush(acc, bsum, gw, gb, wo, wi, lane);
        }
#endif
        u += cend - cu;
    }
}

// =================================================================================================================
// The same GEMMs on the bf16 matrix pipe at fp32 accuracy (scheme of r2l_fwd3.hip: every fp32 operand value is the exact sum
// of three bf16 numbers, six bf16 products stand for one fp32 product).  Both operands are activations, so both are split
// here, on the VALU: one k-step is 16 rays; a lane loads the 4-feature pieces of its 8 rays for both operands (16 x 16 B),
// splits the 64 values (packed v_cvt_pk_bf16_f32 + shift/mask + subtract) and issues 16 tiles x 6 = 96 MFMAs.  The rows
// past N of a slot are exact zeros in the gradient operands (the chains write zero gradients for padding rays), so the k
// loop simply runs to the padded row count: no tails, no predicates.  Work split, accumulator layout and the slab flush are
// those of r2l_dw_body_kernel.
// =================================================================================================================
typedef __bf16 dw3_bf16x8 __attribute__((ext_vector_type(8)));
struct Dw3Split {
    dw3_bf16x8 h, m, l;
};
__device__ __forceinline__ unsigned dw3_pk(float a0, float a1) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a0, a1}, bf16x2));
}
__device__ __forceinline__ float dw3_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float dw3_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
// component e of the eight loaded pieces (8 rays) -> bf16 (hi, mid, lo) operand registers
__device__ __forceinline__ Dw3Split dw3_split(const f32x4 (&v)[8], int e) {
    u32x4 uh, um, ul;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = v[2 * p][e], x1 = v[2 * p + 1][e];
        const unsigned h = dw3_pk(x0, x1);
        const float r0 = x0 - dw3_lo(h), r1 = x1 - dw3_hi(h);
        const unsigned m = dw3_pk(r0, r1);
        const unsigned l = dw3_pk(r0 - dw3_lo(m), r1 - dw3_hi(m));
        uh[p] = h; um[p] = m; ul[p] = l;
    }
    Dw3Split s;
    s.h = __builtin_bit_cast(dw3_bf16x8, uh);
    s.m = __builtin_bit_cast(dw3_bf16x8, um);
    s.l = __builtin_bit_cast(dw3_bf16x8, ul);
    return s;
}

__global__ __launch_bounds__(256, 1) void r2l_dw_body3_kernel(const R2LDwArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wo = wave >> 1, wi = wave & 1;
    const int hh = lane >> 5, jl = lane & 31;
    const int64_t total = a.units_per_layer * a.n_layers;
    int64_t u0 = (int64_t)blockIdx.x * a.units_per_wg;
    int64_t u1 = u0 + a.units_per_wg;
    if (u1 > total) u1 = total;
    if (u0 >= u1) return;

    f32x16 acc[4][4];
#pragma unroll
    for (int eo = 0; eo < 4; ++eo)
#pragma unroll
        for (int ei = 0; ei < 4; ++ei)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[eo][ei][c] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const int64_t Np = R2L_PAD_ROWS(a.N);

    int64_t u = u0;
    const int first_layer = a.layer0 + (int)(u0 / a.units_per_layer);
    while (u < u1) {
        const int layer = a.layer0 + (int)(u / a.units_per_layer);
        const int64_t cu = u % a.units_per_layer;
        int64_t cend = cu + (u1 - u);
        if (cend > a.units_per_layer) cend = a.units_per_layer;
        const int b = layer >> 1;
        const float* G = (layer & 1) ? a.gx + (int64_t)(b + 1) * Np * R2L_W : a.gt + (int64_t)b * Np * R2L_W;
        const float* A = (layer & 1) ? a.save_t + (int64_t)b * Np * R2L_W : a.save_x + (int64_t)b * Np * R2L_W;
        const int64_t r0 = cu * DW_CHUNK;
        int64_t r1 = cend * DW_CHUNK;
        if (r1 > Np) r1 = Np;
        const int64_t nsteps = (r1 - r0) / 16;  // Np is a multiple of 32
        const __amdgpu_buffer_rsrc_t grs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G + r0 * R2L_W), 0, 0xffffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t ars =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + r0 * R2L_W), 0, 0xffffffff, 0x00020000);
        // lane (jl, hh): rays 8hh .. 8hh+7 of the 16-ray step, features 4jl .. 4jl+3 of the wave's 128-feature slice
        const unsigned gvo = (unsigned)(8 * hh * R2L_W + wo * 128 + 4 * jl) * 4u;
        const unsigned avo = (unsigned)(8 * hh * R2L_W + wi * 128 + 4 * jl) * 4u;
        auto ld = [&](int64_t s, f32x4 (&gv)[8], f32x4 (&av)[8]) {
            const int64_t sc = s < nsteps ? s : nsteps - 1;  // the prefetch behind the last step re-reads it
            const unsigned so = (unsigned)sc * (16u * R2L_W * 4u);
            // rows r < 4 via the instruction's 12-bit offset field, rows 4..7 through the scalar offset (+4096); the opaque
            // copies keep hipcc from materialising voff + const in 16 loop-invariant VGPRs
            unsigned gq = gvo, aq = avo;
            asm volatile("" : "+v"(gq), "+v"(aq));
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const unsigned ro = (unsigned)(r & 3) * (R2L_W * 4u), sx = so + (unsigned)(r >> 2) * 4096u;
                gv[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(grs, gq + ro, sx, 0));
                av[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, aq + ro, sx, 0));
            }
        };
        auto kstep = [&](const f32x4 (&gv)[8], const f32x4 (&av)[8]) {
            Dw3Split bs[4];
#pragma unroll
            for (int ei = 0; ei < 4; ++ei) bs[ei] = dw3_split(av, ei);
#pragma unroll
            for (int eo = 0; eo < 4; ++eo) {
                const Dw3Split as = dw3_split(gv, eo);
#pragma unroll
                for (int ei = 0; ei < 4; ++ei) acc[eo][ei] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as.l, bs[ei].h, acc[eo][ei], 0, 0, 0);
#pragma unroll
                for (int ei = 0; ei < 4; ++ei) acc[eo][ei] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as.h, bs[ei].l, acc[eo][ei], 0, 0, 0);
#pragma unroll
                for (int ei = 0; ei < 4; ++ei) acc[eo][ei] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as.m, bs[ei].m, acc[eo][ei], 0, 0, 0);
#pragma unroll
                for (int ei = 0; ei < 4; ++ei) acc[eo][ei] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as.m, bs[ei].h, acc[eo][ei], 0, 0, 0);
#pragma unroll
                for (int ei = 0; ei < 4; ++ei) acc[eo][ei] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as.h, bs[ei].m, acc[eo][ei], 0, 0, 0);
#pragma unroll
                for (int ei = 0; ei < 4; ++ei) acc[eo][ei] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as.h, bs[ei].h, acc[eo][ei], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) bsum += gv[r];
        };
        if (nsteps > 0) {
            f32x4 g0[8], x0[8], g1[8], x1[8];
            ld(0, g0, x0);
            int64_t s = 0;
            for (; s + 2 <= nsteps; s += 2) {  // two steps per trip: the buffers swap roles without register copies
                ld(s + 1, g1, x1);
                kstep(g0, x0);
                ld(s + 2, g0, x0);
                kstep(g1, x1);
            }
        }
        if (a.slab != nullptr) {
            dw_flush_slab(acc, bsum, a.slab + ((int64_t)blockIdx.x * 2 + (layer - first_layer)) * DW_SLAB_FLOATS, wo, wi,
                          lane);
        } else {
            float* gw = a.grads + b_off_body_w(layer);
            float* gb = a.grads + b_off_body_b(layer);
            dw_flush(acc, bsum, gw, gb, wo, wi, lane);
        }
        u += cend - cu;
    }
}

// =================================================================================================================
// The same GEMMs fed from the CHUNKED fp32 stash of the bf16x3 chains (r2l_common.h), every operand value split ONCE per
// workgroup.  (r2l_dw_body3_kernel lets each wave load and split its own operands: every value is split by two waves, and
// the split — ~6 VALU instructions per value — is what bounds it.)
//   * one k-step = 16 rays (half a tile): per operand 32 chunks x 512 contiguous bytes.  Wave w loads chunks 8w .. 8w+7 of
//     both operands (8 x 16 B per lane: lane = (chunk parity, ray, half chunk) -> 4 consecutive features of one ray),
//     splits them into bf16 (hi, mid, lo) (packed v_cvt_pk_bf16_f32, shift / mask, subtract) and writes the three 8-byte
//     quads into the step image in LDS: [split][chunk][slot S][8 B];
//   * the MFMA operands (lane = feature, 8 consecutive rays in its 16 bytes) come out of the image through
//     ds_read_b64_tr_b16: within a 16-lane group lane L = 4*row + q supplies the address of four consecutive bf16 (ray row,
//     feature quad q) and lane l receives rows 0..3 of column l — 4 rays of feature l (measured: tools/tr_probe.hip);
//   * slot S of (ray, quad fq = 2 chunk + half) = half*16 + ((ray + 4 (chunk & 3)) & 15): the 32 lanes of a read pass (two
//     16-feature blocks x four quads x four rays) and of a write pass (16 rays x two halves) hit 32 different 8-byte slots;
//   * software pipeline per step s (two images, one barrier per step): the 4 x TERMS groups of four MFMAs of step s carry
//     along the transposing reads of step s+1 (image s+1 -> registers), the split + LDS write of step s+2 (raw registers ->
//     image s+2, which takes the buffer of image s) and the global loads of step s+3 (in place into the raw registers, one
//     step of latency cover; inline asm with hand-placed vmcnt(7) waits: hipcc would drain vmcnt to 0 at the loop header);
//   * db: the loader lanes add up their raw fp32 gradient quads (4 adds per quad); the 16 rays of a quad column are
//     combined by xor-shuffles at the flush.
// Work split, accumulators and the slab reduce are those of r2l_dw_body_kernel; tile eo of a wave's 128-feature slice is
// features 32 eo .. 32 eo + 31 here (row m of the MFMA result = feature 32 eo + m).
// (Round 1's 3-product and fp16 variants of this kernel were retired with r2l_dw16.hip.)
// =================================================================================================================
#define DW3C_OP_BYTES 24576   // one operand, one step: 3 splits x 32 chunks x 256 B
#define DW3C_BUF_BYTES 49152  // G image, A image
typedef short dw3c_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned dw3c_u32x2 __attribute__((ext_vector_type(2)));

// 8 rays of this lane's feature: two transposing reads (rays 8hh + 0..3 at p0, 8hh + 4..7 at p1), byte offset off (a constant
// after unrolling: it ends up in the instruction's offset field)
__device__ __forceinline__ dw3_bf16x8 dw3c_read(unsigned p0, unsigned p1, unsigned off) {
    typedef __attribute__((address_space(3))) dw3c_s16x4 lds_v;
    const dw3c_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(size_t)(p0 + off));
    const dw3c_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(size_t)(p1 + off));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(dw3_bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}
// the operand registers of one k-step: gs = tiles of the gradient operand (MFMA A), xs = tiles of the activation operand
struct Dw3cRegs {
    Dw3Split gs[4], xs[4];
};
// fragment F of a step (24 fragments, F < 12: activation tile F/3, split F%3, else gradient tile (F-12)/3, split (F-12)%3);
// gp / ap [t]: lane bases in the
// G / A image for ray quad t
__device__ __forceinline__ void dw3c_frag(int F, Dw3cRegs& R, const unsigned (&gp)[2], const unsigned (&ap)[2]) {
    constexpr int NS = 3;
    const bool grad = F >= 4 * NS;
    const int f = grad ? F - 4 * NS : F, e = f / NS, sp = f % NS;
    const dw3_bf16x8 val = grad ? dw3c_read(gp[0], gp[1], (unsigned)(e * 1024 + sp * 8192))
                                : dw3c_read(ap[0], ap[1], (unsigned)(e * 1024 + sp * 8192));
    Dw3Split& d = grad ? R.gs[e] : R.xs[e];
    if (sp == 0) d.h = val;
    else if (sp == 1) d.m = val;
    else d.l = val;
}
// 16-byte global load the compiler does not track (so that it does not wait for it at loop headers): the consumer waits
// with dw3c_wait7 — the loads retire in order and exactly 7 younger ones are in flight at every use
__device__ __forceinline__ void dw3c_load(f32x4& dst, u32x4 rsrc, unsigned voff, unsigned soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void dw3c_wait7(f32x4& v) { asm volatile("s_waitcnt vmcnt(7)" : "+v"(v)::"memory"); }
__device__ __forceinline__ void dw3c_wait0(f32x4& v) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory"); }
// split state of one raw quad (4 features of one ray)
struct Dw3cQuad {
    float r[4];
    unsigned uh[2], um[2], ul[2];
};
// part 0: hi + first residual; part 1: mid + second residual; part 2: lo
__device__ __forceinline__ void dw3c_split_part(Dw3cQuad& q, const f32x4& x, int part) {
    if (part == 0) {
        q.uh[0] = dw3_pk(x[0], x[1]);
        q.uh[1] = dw3_pk(x[2], x[3]);
        q.r[0] = x[0] - dw3_lo(q.uh[0]); q.r[1] = x[1] - dw3_hi(q.uh[0]);
        q.r[2] = x[2] - dw3_lo(q.uh[1]); q.r[3] = x[3] - dw3_hi(q.uh[1]);
    } else if (part == 1) {
        q.um[0] = dw3_pk(q.r[0], q.r[1]);
        q.um[1] = dw3_pk(q.r[2], q.r[3]);
        q.r[0] -= dw3_lo(q.um[0]); q.r[1] -= dw3_hi(q.um[0]);
        q.r[2] -= dw3_lo(q.um[1]); q.r[3] -= dw3_hi(q.um[1]);
    } else {
        q.ul[0] = dw3_pk(q.r[0], q.r[1]);
        q.ul[1] = dw3_pk(q.r[2], q.r[3]);
    }
}
__device__ __forceinline__ void dw3c_write(unsigned addr, unsigned d0, unsigned d1) {
    typedef __attribute__((address_space(3))) dw3c_u32x2 lds_u2;
    *(lds_u2*)(size_t)addr = dw3c_u32x2{d0, d1};
}

__global__ __launch_bounds__(256, 1) void r2l_dw_body3c_kernel(const R2LDwArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char img[2][DW3C_BUF_BYTES];
    constexpr int TERMS = 6;  // bf16 products per fp32 product
    if (a.run_if != nullptr && __builtin_nontemporal_load(a.run_if) == 0u) return;
    const float unscale = a.scale_dev != nullptr ? a.scale_dev[1] : a.unscale;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wo = wave >> 1, wi = wave & 1;
    const int64_t total = a.units_per_layer * a.n_layers;
    int64_t u0 = (int64_t)blockIdx.x * a.units_per_wg;
    int64_t u1 = u0 + a.units_per_wg;
    if (u1 > total) u1 = total;
    if (u0 >= u1) return;

    f32x16 acc[4][4];
#pragma unroll
    for (int eo = 0; eo < 4; ++eo)
#pragma unroll
        for (int ei = 0; ei < 4; ++ei)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[eo][ei][c] = 0.f;
    f32x4 bacc[4];  // db partials of this loader lane: gradient chunk pair kk, its 4 features, summed over its ray column
#pragma unroll
    for (int k = 0; k < 4; ++k) bacc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t Np = R2L_PAD_ROWS(a.N);
    const int64_t slot = R2L_TRIO_SLOT(Np);
    const unsigned img_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)&img[0][0];

    // loader lane: chunk parity cc inside the instruction's chunk pair, ray j of the half tile, half chunk hf
    const int lcc = lane >> 5, lj = (lane >> 1) & 15, lhf = lane & 1;
    const unsigned lvoff = (unsigned)(lcc * 1024 + lj * 32 + lhf * 16);
    // its write address in an operand image for instruction kk (chunk 8 wave + 2 kk + cc): [split][chunk][S][8 B];
    // (chunk & 3) = (2 kk + cc) & 3 -> two rotations, even / odd kk
    unsigned wbase[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int rot = 4 * ((2 * par + lcc) & 3);
        wbase[par] = img_lds + (unsigned)(wave * 2048 + lcc * 256 + (lhf * 16 + ((lj + rot) & 15)) * 8);
    }
    // read bases: 16-lane group (hh, par): 16-feature block parity par inside the tile, rays 8hh..; lane L = 4*row + q in it
    unsigned gb0[2], ab0[2];
    {
        const int g = lane >> 4, L = lane & 15, par = g & 1, hh = g >> 1, q = L & 3, row = L >> 2;
        const int cl = 2 * par + (q >> 1);  // chunk inside the tile's four
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int S = (q & 1) * 16 + ((8 * hh + 4 * t + row + 4 * (cl & 3)) & 15);
            const unsigned base = (unsigned)(cl * 256 + S * 8);
            gb0[t] = img_lds + base + (unsigned)wo * 4096u;
            ab0[t] = img_lds + DW3C_OP_BYTES + base + (unsigned)wi * 4096u;
        }
    }

    int64_t u = u0;
    const int first_layer = a.layer0 + (int)(u0 / a.units_per_layer);
    while (u < u1) {
        const int layer = a.layer0 + (int)(u / a.units_per_layer);
        const int64_t cu = u % a.units_per_layer;
        int64_t cend = cu + (u1 - u);
        if (cend > a.units_per_layer) cend = a.units_per_layer;
        const int b = layer >> 1;
        const float* G = (layer & 1) ? a.gx + (int64_t)(b + 1) * slot : a.gt + (int64_t)b * slot;
        const float* A = (layer & 1) ? a.save_t + (int64_t)b * slot : a.save_x + (int64_t)b * slot;
        const int64_t r0 = cu * DW_CHUNK;  // a multiple of 64 rays: two whole tiles
        int64_t r1 = cend * DW_CHUNK;
        if (r1 > Np) r1 = Np;
        const int nsteps = (int)((r1 - r0) / 16);  // Np is a multiple of 32: always even
        // descriptors based at the first tile of the segment (+ this wave's eight chunks)
        const unsigned long long ga = (unsigned long long)(G + r0 * R2L_W + wave * 8 * R2L_CHUNK_PIECE);
        const unsigned long long aa = (unsigned long long)(A + r0 * R2L_W + wave * 8 * R2L_CHUNK_PIECE);
        const u32x4 grs = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ga),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ga >> 32)) & 0xffffu, 0xffffffffu, 0x00020000u};
        const u32x4 ars = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)aa),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(aa >> 32)) & 0xffffu, 0xffffffffu, 0x00020000u};
        // raw quad k of a step: k < 4 gradient chunk pair k, else activation chunk pair k - 4.  Steps past the end are
        // clamped to the last one (harmless reloads: every step issues exactly 8 loads, which keeps vmcnt(7) uniform)
        f32x4 raw[8];
        auto load = [&](int s, int k) {
            const int sc = s < nsteps ? s : nsteps - 1;
            const unsigned so = (unsigned)(sc >> 1) * (unsigned)(R2L_CHUNK_TILE * 4) + (unsigned)(sc & 1) * 512u + (unsigned)(k & 3) * 2048u;
            dw3c_load(raw[k], (k < 4) ? grs : ars, lvoff, so);
        };
        Dw3cQuad qs;
        // parts of the split of raw quad k into the image in buffer `buf`; the last part writes the quads and reloads raw[k]
        // with the data of step s_next
        auto quad_part = [&](int k, int part, int buf, int s_next) {
            if (part == 0) {
                dw3c_wait7(raw[k]);
                // (raw holds step s_next - 1: a clamped reload past the end must not be counted)
                if (k < 4) bacc[k] += raw[k] * ((s_next - 1 < nsteps) ? 1.f : 0.f);
            }
            dw3c_split_part(qs, raw[k], part);
            if (part == 2) {
                const unsigned wa = wbase[k & 1] + (unsigned)buf * DW3C_BUF_BYTES + (unsigned)(k >> 2) * DW3C_OP_BYTES + (unsigned)(k & 3) * 512u;
                dw3c_write(wa, qs.uh[0], qs.uh[1]);
                dw3c_write(wa + 8192u, qs.um[0], qs.um[1]);
                dw3c_write(wa + 16384u, qs.ul[0], qs.ul[1]);
                load(s_next, k);
            }
        };
        // One k-step: 16 x TERMS MFMAs on the registers C; riding along, per group of four MFMAs: fragments of step s+1 (LDS
        // image s+1 -> registers Nx) and a part of the split of step s+2 (raw -> image s+2 in the buffer image s occupied;
        // raw reloaded with step s+3)
        auto step = [&](Dw3cRegs& C, Dw3cRegs& Nx, int s, int buf) {
            // image s+1 is complete (everybody wrote its share during step s-1) and nobody reads image s any more
            __syncthreads();
            const unsigned bo = (unsigned)(buf ^ 1) * DW3C_BUF_BYTES;  // image of step s+1
            const unsigned gp[2] = {gb0[0] + bo, gb0[1] + bo}, ap[2] = {ab0[0] + bo, ab0[1] + bo};
#pragma unroll
            for (int g = 0; g < 4 * TERMS; ++g) {
                const int eo = g / TERMS, term = g % TERMS;
                // small terms first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
                const int tk = term;
                const dw3_bf16x8& ga2 = (tk == 0) ? C.gs[eo].l : (tk == 2 || tk == 3) ? C.gs[eo].m : C.gs[eo].h;
#pragma unroll
                for (int ei = 0; ei < 4; ++ei) {
                    const dw3_bf16x8& xb = (tk == 1) ? C.xs[ei].l : (tk == 2 || tk == 4) ? C.xs[ei].m : C.xs[ei].h;
                    acc[eo][ei] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga2, xb, acc[eo][ei], 0, 0, 0);
                }
                // 24 groups: 24 fragments, 8 quads x 3 parts
                dw3c_frag(g, Nx, gp, ap);
                quad_part(g / 3, g % 3, buf, s + 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (nsteps > 0) {
            Dw3cRegs RA, RB;
            // prologue: images 0 and 1, raw = step 2 (latencies exposed once per segment)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int k = 0; k < 8; ++k) load(s, k);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    dw3c_wait0(raw[k]);
                    if (k < 4) bacc[k] += raw[k];
#pragma unroll
                    for (int part = 0; part < 3; ++part) dw3c_split_part(qs, raw[k], part);
                    const unsigned wa = wbase[k & 1] + (unsigned)s * DW3C_BUF_BYTES + (unsigned)(k >> 2) * DW3C_OP_BYTES + (unsigned)(k & 3) * 512u;
                    dw3c_write(wa, qs.uh[0], qs.uh[1]);
                    dw3c_write(wa + 8192u, qs.um[0], qs.um[1]);
                    dw3c_write(wa + 16384u, qs.ul[0], qs.ul[1]);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) load(2, k);
            __syncthreads();
#pragma unroll
            for (int F = 0; F < 24; ++F) dw3c_frag(F, RA, gb0, ab0);  // step 0 from buffer 0
            for (int s = 0; s < nsteps; s += 2) {
                step(RA, RB, s, 0);
                step(RB, RA, s + 1, 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // the images are dead (and the clamped reloads landed) before the next segment refills them
        }
        // flush: D[m][n] of tile (eo, ei): m = 8 (c>>2) + 4 hh + (c&3) -> output feature wo*128 + 32 eo + m, n = lane & 31 ->
        // input feature wi*128 + 32 ei + n
        {
            const int n = lane & 31, hh = lane >> 5;
            float* sl = (a.slab != nullptr) ? a.slab + ((int64_t)blockIdx.x * 2 + (layer - first_layer)) * DW_SLAB_FLOATS : nullptr;
            float* gw = a.grads + b_off_body_w(layer);
            float* gbias = a.grads + b_off_body_b(layer);
            float* rowp = (sl != nullptr ? sl : gw) + (wo * 128 + 4 * hh) * R2L_W + wi * 128 + n;
            if (sl != nullptr) {
#pragma unroll
                for (int eo = 0; eo < 4; ++eo)
#pragma unroll
                    for (int ei = 0; ei < 4; ++ei)
#pragma unroll
                        for (int c = 0; c < 16; ++c) {
                            rowp[(32 * eo + 8 * (c >> 2) + (c & 3)) * R2L_W + 32 * ei] = acc[eo][ei][c] * unscale;
                            acc[eo][ei][c] = 0.f;
                        }
            } else {
#pragma unroll
                for (int eo = 0; eo < 4; ++eo)
#pragma unroll
                    for (int ei = 0; ei < 4; ++ei)
#pragma unroll
                        for (int c = 0; c < 16; ++c) {
                            atomicAdd(rowp + (32 * eo + 8 * (c >> 2) + (c & 3)) * R2L_W + 32 * ei, acc[eo][ei][c] * unscale);
                            acc[eo][ei][c] = 0.f;
                        }
            }
            // db: loader lane (cc, j, hf) holds, for kk = 0..3, features 8 (8 wave + 2 kk + cc) + 4 hf .. +3 summed over its ray
            // column j; the 16 columns sit in lane bits 1..4
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 v = bacc[k];
#pragma unroll
                for (int m = 2; m <= 16; m <<= 1)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += __shfl_xor(v[e], m);
                v *= unscale;
                if (lj == 0) {
                    const int f = 8 * (8 * wave + 2 * k + lcc) + 4 * lhf;
                    if (sl != nullptr) *reinterpret_cast<f32x4*>(sl + R2L_W * R2L_W + f) = v;
                    else
#pragma unroll
                        for (int e = 0; e < 4; ++e) atomicAdd(gbias + f + e, v[e]);
                }
                bacc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        u += cend - cu;
    }
}

// =================================================================================================================
// Head weight gradient:  dWh[o][k] = sum_r Gh[r][o] * PE[r][k]   (k in 1008, padded to 1024), dbh[o] = sum_r Gh[r][o]
// The encoding is recomputed from the rays (never stored: 4 KB/ray).  Workgroup (kq, slice): kq selects 256 encoding
// columns; wave w of it owns columns kq*256 + w*64 .. +63 (two 32-column tiles) x all 256 output rows (8 tiles).
// Lane (i, h) evaluates encoding column k = base + i (+32) for ray 2s+h: one sin or cos (or the identity) per tile.
// =================================================================================================================
struct HeadStep {  // raw operands of one k-step (two rays), as loaded
    f32x4 g0, g1;
    float o0, d0, u0, o1, d1, u1;
};

// FROM_EMB / JITTER are compile-time so that the pipelined loop body is branch-free (runtime flags inside it made hipcc
// emit per-load branches, spills and vmcnt(0) drains).
template <bool FROM_EMB, bool JITTER>
__global__ __launch_bounds__(256, 1) void r2l_dw_head_kernel(const R2LDwHeadArgs a) {
    if (a.run_if != nullptr && __builtin_nontemporal_load(a.run_if) == 0u) return;  // fallback behind r2l_dw_head16_kernel
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hh = lane >> 5, jl = lane & 31;
    const int kq = blockIdx.x & 3;
    const int64_t slice = blockIdx.x >> 2;
    const int64_t r0 = slice * a.rays_per_wg;
    int64_t r1 = r0 + a.rays_per_wg;
    if (r1 > a.N) r1 = a.N;
    if (r0 >= r1) return;
    const int kbase = kq * 256 + wave * 64;
    constexpr bool from_emb = FROM_EMB;
    constexpr bool jitter = JITTER && !FROM_EMB;
    const int k0 = kbase + jl, k1 = kbase + 32 + jl;
    const PECol c0 = pe_col(k0), c1 = pe_col(k1);
    const int e0 = k0 < R2L_IN ? k0 : R2L_IN - 1, e1 = k1 < R2L_IN ? k1 : R2L_IN - 1;  // clamped emb columns
    const float m0 = k0 < R2L_IN ? 1.f : 0.f, m1 = k1 < R2L_IN ? 1.f : 0.f;
    const float zl0 = from_emb ? 0.f : a.ztab[c0.smp], zs0 = jitter ? a.ztab[16 + c0.smp] : 0.f;
    const float zl1 = from_emb ? 0.f : a.ztab[c1.smp], zs1 = jitter ? a.ztab[16 + c1.smp] : 0.f;

    f32x16 acc[8][2];
#pragma unroll
    for (int eo = 0; eo < 8; ++eo)
#pragma unroll
        for (int ei = 0; ei < 2; ++ei)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[eo][ei][c] = 0.f;
    f32x4 bs0 = {0.f, 0.f, 0.f, 0.f}, bs1 = {0.f, 0.f, 0.f, 0.f};

    const int64_t nfull = (r1 - r0) / 2;
    // unconditional loads of k-step s (row index clamped into the slice)
    auto ld = [&](int64_t s, HeadStep& v) {
        const int64_t sc = s < nfull ? s : (nfull > 0 ? nfull - 1 : 0);
        const int64_t r = r0 + 2 * sc + hh;
        v.g0 = *reinterpret_cast<const f32x4*>(a.gh + r * R2L_W + 4 * jl);
        v.g1 = *reinterpret_cast<const f32x4*>(a.gh + r * R2L_W + 128 + 4 * jl);
        if constexpr (from_emb) {
            v.o0 = a.emb[r * R2L_IN + e0];
            v.o1 = a.emb[r * R2L_IN + e1];
            v.d0 = v.d1 = v.u0 = v.u1 = 0.f;
        } else {
            v.o0 = a.rays_o[r * 3 + c0.ax]; v.d0 = a.rays_d[r * 3 + c0.ax];
            v.o1 = a.rays_o[r * 3 + c1.ax]; v.d1 = a.rays_d[r * 3 + c1.ax];
            if constexpr (jitter) {
                v.u0 = a.t_rand[r * 16 + c0.smp];
                v.u1 = a.t_rand[r * 16 + c1.smp];
            } else {
                v.u0 = v.u1 = 0.f;
            }
        }
    };
    auto encode = [&](const HeadStep& v, float& p0, float& p1) {
        if constexpr (from_emb) { p0 = v.o0 * m0; p1 = v.o1 * m1; return; }
        const float z0 = jitter ? zl0 + zs0 * v.u0 : zl0;
        const float z1 = jitter ? zl1 + zs1 * v.u1 : zl1;
        p0 = pe_eval(c0, v.o0 + v.d0 * z0);
        p1 = pe_eval(c1, v.o1 + v.d1 * z1);
    };
    auto kstep = [&](const f32x4& g0, const f32x4& g1, float p0, float p1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[e][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0[e], p0, acc[e][0], 0, 0, 0);
            acc[e][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0[e], p1, acc[e][1], 0, 0, 0);
            acc[4 + e][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1[e], p0, acc[4 + e][0], 0, 0, 0);
            acc[4 + e][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1[e], p1, acc[4 + e][1], 0, 0, 0);
        }
        bs0 += g0;
        bs1 += g1;
    };
    if (nfull > 0) {
        // 4 rotating raw-operand buffers (loads three k-steps ahead); the encoding of k-step s+1 is evaluated on the
        // VALU while the 32 MFMAs of k-step s run.  16 k-steps per loop trip (hipcc drains vmcnt at loop headers).
        HeadStep hb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ld(k, hb[k]);
        float pa0, pa1;
        encode(hb[0], pa0, pa1);
        int64_t s = 0;
        for (; s + 16 <= nfull; s += 16) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const f32x4 g0 = hb[k & 3].g0, g1 = hb[k & 3].g1;
                float pn0, pn1;
                encode(hb[(k + 1) & 3], pn0, pn1);  // next k-step's columns
                kstep(g0, g1, pa0, pa1);
                ld(s + k + 4, hb[k & 3]);
                pa0 = pn0;
                pa1 = pn1;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (; s < nfull; ++s) {  // remainder of the slice
            HeadStep v;
            ld(s, v);
            float p0, p1;
            encode(v, p0, p1);
            kstep(v.g0, v.g1, p0, p1);
        }
    }
    if ((r1 - r0) & 1) {  // odd tail: only the lower half-wave's ray exists
        const int64_t r = r1 - 1;
        HeadStep v;
        v.g0 = *reinterpret_cast<const f32x4*>(a.gh + r * R2L_W + 4 * jl);
        v.g1 = *reinterpret_cast<const f32x4*>(a.gh + r * R2L_W + 128 + 4 * jl);
        if constexpr (from_emb) {
            v.o0 = a.emb[r * R2L_IN + e0];
            v.o1 = a.emb[r * R2L_IN + e1];
            v.d0 = v.d1 = v.u0 = v.u1 = 0.f;
        } else {
            v.o0 = a.rays_o[r * 3 + c0.ax]; v.d0 = a.rays_d[r * 3 + c0.ax];
            v.o1 = a.rays_o[r * 3 + c1.ax]; v.d1 = a.rays_d[r * 3 + c1.ax];
            v.u0 = jitter ? a.t_rand[r * 16 + c0.smp] : 0.f;
            v.u1 = jitter ? a.t_rand[r * 16 + c1.smp] : 0.f;
        }
        float p0, p1;
        encode(v, p0, p1);
        if (hh) { v.g0 = f32x4{0.f, 0.f, 0.f, 0.f}; v.g1 = v.g0; p0 = 0.f; p1 = 0.f; }
        kstep(v.g0, v.g1, p0, p1);
    }
    // flush: output row o = half*128 + 4*ro + e, encoding column k = kbase + ei*32 + jl
    float* gw = a.grads;  // head.0.weight is first in the flat buffer, [256][1008]
    float* sl = a.slab ? a.slab + slice * (int64_t)(R2L_W * 1024) : nullptr;
#pragma unroll
    for (int eo = 0; eo < 8; ++eo)
#pragma unroll
        for (int ei = 0; ei < 2; ++ei)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int ro = (c & 3) + 8 * (c >> 2) + 4 * hh;
                const int o = (eo >> 2) * 128 + 4 * ro + (eo & 3);
                const int k = kbase + ei * 32 + jl;
                if (k < R2L_IN) {
                    if (sl) sl[o * 1024 + k] = acc[eo][ei][c];  // 64 slices x 64-way same-address atomics are slow
                    else atomicAdd(gw + (int64_t)o * R2L_IN + k, acc[eo][ei][c]);
                }
            }
    if (kq == 0 && wave == 0) {
        float* gb = a.grads + b_off_head_b();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float s0 = bs0[e] + __shfl_xor(bs0[e], 32);
            const float s1 = bs1[e] + __shfl_xor(bs1[e], 32);
            if (hh == 0) {
                if (sl) {  // bias partial of row o rides in the (otherwise unused) padding column 1008 of the slab row
                    sl[(4 * jl + e) * 1024 + R2L_IN] = s0;
                    sl[(128 + 4 * jl + e) * 1024 + R2L_IN] = s1;
                } else {
                    atomicAdd(gb + 4 * jl + e, s0);
                    atomicAdd(gb + 128 + 4 * jl + e, s1);
                }
            }
        }
    }
}

// dWh[o][k] += sum over slices of slab[slice][o][k] (k < 1008), dbh[o] += sum of slab[slice][o][1008]; slices are
// added in index order (deterministic)
__global__ void r2l_head_reduce_kernel(const float* __restrict__ slab, int n_slices, float* __restrict__ grads) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [256][1024]
    if (i >= (int64_t)R2L_W * 1024) return;
    const int o = (int)(i >> 10), k = (int)(i & 1023);
    if (k > R2L_IN) return;
    float s = 0.f;
    for (int sidx = 0; sidx < n_slices; ++sidx) s += slab[(int64_t)sidx * (R2L_W * 1024) + i];
    if (k == R2L_IN) grads[b_off_head_b() + o] += s;  // column 1008 carries the bias partials
    else grads[(int64_t)o * R2L_IN + k] += s;
}

// =================================================================================================================
// Tail gradients: dWt[c][f] = sum_r dpre[r][c] * (x_n[r][f] + x_0[r][f]),  dbt[c] = sum_r dpre[r][c]
// =================================================================================================================
__global__ __launch_bounds__(256) void r2l_dw_tail_kernel(const float* __restrict__ dpre, const float* __restrict__ x0,
                                                          const float* __restrict__ xn, float* __restrict__ grads,
                                                          float* __restrict__ part, int n_block, int64_t N,
                                                          int64_t rays_per_wg) {
    const int f = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * rays_per_wg;
    int64_t r1 = r0 + rays_per_wg;
    if (r1 > N) r1 = N;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f;
    // (8 rows in flight per thread: one row per trip left this loop load-latency bound, 0.10 ms per 98 304 rays for 100 MB)
#pragma unroll 8
    for (int64_t r = r0; r < r1; ++r) {
        const float y = x0 != nullptr ? xn[r * R2L_W + f] + x0[r * R2L_W + f] : xn[r * R2L_W + f];  // x0 == nullptr: xn holds y
        const float d0 = dpre[r * 3 + 0], d1 = dpre[r * 3 + 1], d2 = dpre[r * 3 + 2];
        s0 = __builtin_fmaf(d0, y, s0);
        s1 = __builtin_fmaf(d1, y, s1);
        s2 = __builtin_fmaf(d2, y, s2);
        b0 += d0; b1 += d1; b2 += d2;
    }
    if (part != nullptr) {  // [wg][4][256]: rows 0-2 = dWt partial, row 3 = dbt partial (first 3 entries)
        float* p = part + (int64_t)blockIdx.x * (4 * R2L_W);
        p[0 * R2L_W + f] = s0;
        p[1 * R2L_W + f] = s1;
        p[2 * R2L_W + f] = s2;
        if (f == 0) { p[3 * R2L_W + 0] = b0; p[3 * R2L_W + 1] = b1; p[3 * R2L_W + 2] = b2; }
        return;
    }
    float* gw = grads + b_off_tail_w(n_block);
    atomicAdd(gw + 0 * R2L_W + f, s0);
    atomicAdd(gw + 1 * R2L_W + f, s1);
    atomicAdd(gw + 2 * R2L_W + f, s2);
    if (f == 0 && r0 < r1) {
        float* gb = grads + b_off_tail_b(n_block);
        atomicAdd(gb + 0, b0);
        atomicAdd(gb + 1, b1);
        atomicAdd(gb + 2, b2);
    }
}

// tail.0.{weight,bias} += partials of all workgroups in a fixed order: thread (f, j) adds the partials of workgroups
// w = j, j+8, ... (8 independent chains keep enough loads in flight), then the 8 sums are added in j order.
__global__ __launch_bounds__(256) void r2l_tail_reduce_kernel(const float* __restrict__ part, int64_t wgs,
                                                              float* __restrict__ grads, int n_block) {
    __shared__ float red[8][32];
    const int row = blockIdx.y, f = blockIdx.x * 32 + (threadIdx.x & 31), j = threadIdx.x >> 5;
    float s = 0.f;
#pragma unroll 8
    for (int64_t w = j; w < wgs; w += 8) s += part[w * (4 * R2L_W) + row * R2L_W + f];
    red[j][threadIdx.x & 31] = s;
    __syncthreads();
    if (j == 0) {
        float t = red[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][threadIdx.x];
        if (row < 3) grads[b_off_tail_w(n_block) + row * R2L_W + f] += t;
        else if (f < 3) grads[b_off_tail_b(n_block) + f] += t;
    }
}

// Generic mode of the fp16 trio: the caller's dL/drgb has no known scale, so the power of two that puts the dX chain into
// fp16's range is chosen on the device: out = {gscale, 1 / gscale}, gscale = 2^(8 - e) for max |drgb| = m * 2^e — the rule
// r2l_backward_part applies on the host to the MSE gradient scale.  One workgroup (generic mode is not the training loop's).
__global__ __launch_bounds__(1024) void r2l_gscale_kernel(const float* __restrict__ drgb, int64_t n, float* __restrict__ out) {
    __shared__ float red[1024];
    float m = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(drgb[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + k]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float g = 1.0f;
        if (red[0] > 0.f && red[0] < 3.0e38f) {
            int e = 0;
            (void)frexpf(red[0], &e);
            g = ldexpf(1.0f, 8 - e);
        }
        out[0] = g;
        out[1] = 1.0f / g;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------------
// The 16 status words behind the fp16 dX stream (r2l_common.h B2S_*), at the start of every step of the fp16 trio: FLAG and
// AMAX cleared, and — MSE mode — the power of two the step's dX chain runs on written to GSCALE / GINV:
//   * first step (or after a step that fell back, or a zero gradient): the a-priori rule of rounds 1 - 3, 2^(8 - e) for
//     grad_scale = m * 2^e (the seed is then <= 64 |rgb - target|), corrected by the size of the tail weights the seed goes
//     through: x 2^(-3 - e_w) for max |W_tail| = m * 2^e_w (the default init's 1/16 gives 1);
//   * afterwards the scale is KEPT while the last clean step's largest |chain value| (times grad_scale / its grad_scale: batch
//     sizes may change) lies in [2^-2, 2^13] scaled — a factor 4 below the guard (R2L_F2_RANGE) and, at the low end, still
//     2^-23 of the largest value in absolute fp16 (hi + mid) resolution — and re-centred to 2^8 when it leaves that band.
//     So nets whose gradients live where the a-priori rule expects them (default init, the trained nets measured so far) run
//     bit for bit as in round 3, and a net whose gradients drift (or start) elsewhere follows on the next step instead of
//     falling back (overflow) or silently losing bits (underflow) from then on.
// Powers of two commute with fp32 rounding: whatever the scale, the bf16x3 fallback run on it is bit-identical after unscaling.
// (A kernel, not hipMemsetAsync: a memset node captured into a hipGraph wrote a stale pattern on replay, ROCm 7.0; the history
// is device state, so captured steps adapt on replay as eager ones do.)  generic mode (mse == 0): GSCALE is r2l_gscale_kernel's.
__global__ __launch_bounds__(64) void r2l_bwd_prepare_kernel(unsigned* status, const float* __restrict__ tail_w, float grad_scale,
                                                             int mse) {
    float wm = 0.f;
    if (mse)
        for (int i = threadIdx.x; i < 3 * R2L_W; i += 64) wm = fmaxf(wm, fabsf(tail_w[i]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) wm = fmaxf(wm, __shfl_xor(wm, off));
    if (threadIdx.x != 0) return;
    const float gs = fabsf(grad_scale);
    const bool valid = status[B2S_MAGIC] == F2_MAGIC;
    const bool clean = valid && status[B2S_FLAG] == 0u;
    const float a_raw = valid ? __builtin_bit_cast(float, status[B2S_AMAX]) : 0.f;  // 0: the chain did not run (forward fell back)
    const float a = clean ? a_raw : 0.f;
    const float g_prev = valid ? __builtin_bit_cast(float, status[B2S_GSCALE]) : 0.f;
    const float gs_prev = valid ? __builtin_bit_cast(float, status[B2S_GS]) : 0.f;
    int ex_prev = 0;
    const bool g_ok = g_prev > 0.f && g_prev < 3.0e38f && frexpf(g_prev, &ex_prev) == 0.5f;  // a power of two: 2^(ex_prev - 1)
    float peak = valid ? __builtin_bit_cast(float, status[B2S_PEAK]) : 0.f;
    unsigned trips = valid ? status[B2S_TRIPS] : 0u;
    if (valid && status[B2S_FLAG] != 0u) ++trips;
    // a step whose CHAIN tripped the guard (round 5): its AMAX is the truth while every value stayed finite in fp32 — the guard
    // trips at 32768, the fp32 B values are tracked before their fp16 conversion — so the next step re-centres on it instead of
    // repeating the a-priori rule (and tripping again, step after step, on a net whose gradients grow through the body);
    // AMAX = inf / NaN: only "too large" is known — the scale drops by 2^8 and the following step refines it
    bool blind = false;
    if (clean && g_ok && a > 0.f && a < 3.0e38f) peak = a / g_prev;  // unscaled amax of the last clean step
    else if (!clean && valid && g_ok && a_raw > 0.f && a_raw < 3.0e38f) peak = a_raw / g_prev;
    else if (!clean && valid && g_ok && !(a_raw <= 0.f) && !(a_raw < 3.0e38f)) { blind = true; peak = 0.f; }
    else if (!clean) peak = 0.f;                                      // (a step whose forward fell back says nothing reliable)
    if (mse) {
        int ex = 0;
        const float est = (peak > 0.f && gs_prev > 0.f && gs > 0.f && g_ok) ? peak * (gs / gs_prev) : 0.f;
        if (blind) {
            ex = (ex_prev - 1) - 8;
        } else if (est > 0.f && est < 3.0e38f) {
            const float scaled = est * g_prev;
            if (scaled >= 0.25f && scaled <= 8192.f) {
                ex = ex_prev - 1;
            } else {
                (void)frexpf(est, &ex);
                ex = 8 - ex;
            }
        } else if (gs > 0.f) {
            int e = 0, ew = -3;
            (void)frexpf(gs, &e);
            if (wm > 0.f && wm < 3.0e38f) (void)frexpf(wm, &ew);
            ex = (8 - e) + (-3 - ew);
        }
        ex = ex > 120 ? 120 : (ex < -120 ? -120 : ex);
        status[B2S_GSCALE] = __builtin_bit_cast(unsigned, ldexpf(1.0f, ex));
        status[B2S_GINV] = __builtin_bit_cast(unsigned, ldexpf(1.0f, -ex));
        status[B2S_GS] = __builtin_bit_cast(unsigned, gs);
    } else {
        status[B2S_GS] = 0u;
    }
    status[B2S_PEAK] = __builtin_bit_cast(unsigned, peak);
    status[B2S_TRIPS] = trips;
    status[B2S_MAGIC] = F2_MAGIC;
    status[B2S_AMAX] = 0u;
    status[B2S_EXPANDED] = 0u;
    status[B2S_FLAG] = 0u;
}
extern "C" int64_t r2l_num_tiles(int64_t N) { return (N + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS; }
extern "C" int64_t r2l_padded_rows(int64_t N) { return R2L_PAD_ROWS(N); }
// (+ 16 floats of status words behind the partials)
extern "C" int64_t r2l_dw_slab_floats(void) { return DW_TAIL_SLAB_BASE + DW_TAIL_SLAB + 16; }  // (r2l_dw.h: body | head | tail regions)
extern "C" int64_t r2l_stash_slot_floats(int64_t N) { return R2L_TRIO_SLOT(R2L_PAD_ROWS(N)); }


// May the dX chain of an N-ray step be cut into block segments (R2L_BWD_CHAIN with a layer range)?  1 when the step takes the
// cooperative fp16 chains (small launches of the default trio), else 0.
extern "C" int r2l_chain_segments_ok_cfg(int64_t N, int n_block, const r2l_config* cfg) {
    R2L_CFG_QUERY(cfg);
    return (N > 0 && r2l_chain_variant(N) == R2L_VARIANT_MAIN && r2l_use_fwd3() && r2l_use_trio16() && r2l_use_coopf(N, n_block)) ? 1 : 0;
}
// Device word that the fp16 dX chain raises when a step needs the bf16x3 fallback (range guard, or the forward fell back):
// 0 after a clean step.  A host that runs steps with R2L_BWD_NOFALLBACK hands it to r2l_adam_step_guarded.
extern "C" const unsigned* r2l_backward_status_word(const float* wstream_bwd, int n_block) {
    const float* w3 = wstream_bwd + r2l_bwd32_stream_floats(n_block) + r2l_bwd16_stream_floats(n_block);
    const float* w2 = w3 + r2l_bwd3_stream_floats(n_block);
    return reinterpret_cast<const unsigned*>(w2 + r2l_bwd2_status_offset(n_block));
}

extern "C" const unsigned* r2l_backward_status_words(const float* wstream_bwd, int n_block) {
    if (wstream_bwd == nullptr || n_block < 0 || n_block > R2L_MAX_BLOCKS) return nullptr;
    return r2l_backward_status_word(wstream_bwd, n_block);
}

extern "C" int r2l_backward(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab,
                            const float* emb, const float* rgb, const float* target, const float* drgb,
                            const float* save_x, const float* save_t,
                            const float* wstream_bwd, const float* params, int n_block, float grad_scale, float* dpre,
                            float* gx, float* gt, float* sqerr_partial, float* grads, float* dw_slab, int64_t N,
                            void* stream_) {
    return r2l_backward_part_cfg(rays_o, rays_d, t_rand, ztab, emb, rgb, target, drgb, save_x, save_t, wstream_bwd, params, n_block,
                                 grad_scale, dpre, gx, gt, sqerr_partial, grads, dw_slab, N, stream_, R2L_BWD_ALL, 0, 2 * n_block,
                                 nullptr);
}

// The same backward cut into stages, so that a data-parallel host can hand finished gradient buckets to the collective
// while the remaining stages still run: parts = OR of R2L_BWD_CHAIN (dX chain; must precede everything else of a step),
// R2L_BWD_BODY (weight / bias gradients of the body layers [layer_lo, layer_hi) of the 2*n_block, complete in `grads`
// when the call's kernels have run), R2L_BWD_HEAD, R2L_BWD_TAIL.  Stages of one step go to ONE stream (they share dw_slab).
extern "C" int r2l_backward_part(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab,
                                 const float* emb, const float* rgb, const float* target, const float* drgb,
                                 const float* save_x, const float* save_t,
                                 const float* wstream_bwd, const float* params, int n_block, float grad_scale, float* dpre,
                                 float* gx, float* gt, float* sqerr_partial, float* grads, float* dw_slab, int64_t N,
                                 void* stream_, int parts, int layer_lo, int layer_hi) {
    return r2l_backward_part_cfg(rays_o, rays_d, t_rand, ztab, emb, rgb, target, drgb, save_x, save_t, wstream_bwd, params, n_block,
                                 grad_scale, dpre, gx, gt, sqerr_partial, grads, dw_slab, N, stream_, parts, layer_lo, layer_hi,
                                 nullptr);
}
// largest step whose head / tail gradients run beside the body's (R2L_DW_OVERLAP_MAX_RAYS: tuning knob, tools/run_ab.sh)
static int64_t r2l_dw_overlap_max() {
    static const int64_t v = [] {
        const char* e = getenv("R2L_DW_OVERLAP_MAX_RAYS");
        return e ? (int64_t)atoll(e) : (int64_t)R2L_COOPF_MAX_RAYS;
    }();
    return v;
}
extern "C" int r2l_backward_part_cfg(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab,
                                     const float* emb, const float* rgb, const float* target, const float* drgb,
                                     const float* save_x, const float* save_t,
                                     const float* wstream_bwd, const float* params, int n_block, float grad_scale, float* dpre,
                                     float* gx, float* gt, float* sqerr_partial, float* grads, float* dw_slab, int64_t N,
                                     void* stream_, int parts, int layer_lo, int layer_hi, const r2l_config* cfg) {
    R2L_CFG_ENTER(cfg);
    R2L_REQUIRE(N >= 0 && n_block >= 0 && n_block <= R2L_MAX_BLOCKS, "r2l_backward_part: N / n_block out of range");
    if (N <= 0) return 0;
    R2L_REQUIRE((parts & ~(R2L_BWD_ALL | R2L_BWD_NOFALLBACK)) == 0 && (parts & R2L_BWD_ALL) != 0,
                "r2l_backward_part: parts is an OR of R2L_BWD_CHAIN / BODY / HEAD / TAIL (+ R2L_BWD_NOFALLBACK)");
    R2L_REQUIRE(rgb && (target || drgb) && save_x && (save_t || n_block == 0) && wstream_bwd && params && dpre && gx &&
                    (gt || n_block == 0) && grads,
                "r2l_backward_part: a required pointer is NULL");
    R2L_REQUIRE(emb != nullptr || (rays_o && rays_d && ztab), "r2l_backward_part: needs emb, or rays_o / rays_d / ztab to recompute it");
    if (layer_lo < 0) layer_lo = 0;
    if (layer_hi > 2 * n_block) layer_hi = 2 * n_block;
    // R2L_BWD_CHAIN with a proper sub-range of the layers: ONE SEGMENT of the dX chain (include/r2l_hip.h)
    const bool chain_seg = (parts & R2L_BWD_CHAIN) && layer_hi > layer_lo && !(layer_lo == 0 && layer_hi == 2 * n_block);
    const bool no_fallback = (parts & R2L_BWD_NOFALLBACK) != 0;
    if (chain_seg && ((layer_lo | layer_hi) & 1)) {
        r2l_set_error_msg("r2l_backward_part: a chain segment covers whole blocks (even layer bounds)");
        return (int)hipErrorInvalidValue;
    }
    hipStream_t stream = (hipStream_t)stream_;
    static int n_cu_cached = 0;  // one device type per process
    if (n_cu_cached == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n_cu_cached = v;
        else
            n_cu_cached = 256;
    }
    // Data-parallel hosts overlap the gradient all-reduce with the weight-gradient stages (r2l_backward_part).  Those kernels
    // are persistent workgroups that take every register of their CU, so a collective launched beside them would wait for
    // a whole stage to finish: R2L_RESERVE_CUS=n (set by the host when world_size > 1; r2l_amd/train_step.py uses 8) keeps n
    // CUs out of the weight-gradient launches for the RCCL kernels.  Default 0.
    int reserve = 0;
    if (g_r2l_cfg.reserve_cus) reserve = g_r2l_cfg.reserve_cus;  // (-1: none)
    else if (const char* e = getenv("R2L_RESERVE_CUS")) reserve = atoi(e);
    if (reserve < 0 || reserve > n_cu_cached / 2) reserve = 0;
    const int n_cu = n_cu_cached - reserve;
    // 1. dX chain
    const int variant = r2l_chain_variant(N);
    // the bf16x3 trio (r2l_fwd3 wrote the stash): chunked stash layout, see r2l_common.h
    const bool split = r2l_stash_chunked(N, emb != nullptr);
    // MSE mode of the trio: the dX chain runs on gscale * g, gscale = 2^(8 - e) for grad_scale = m * 2^e (m in [0.5, 1)): the
    // seed gscale * dL/dpre is then <= 64 |rgb - target|, which puts the chain's values into fp16's range for the fp16
    // gradient kernels; powers of two commute with fp32 rounding, so the scaled chain is bit-identical after unscaling
    float gscale = 1.0f;
    if (split && target != nullptr && grad_scale > 0.f) {
        int e = 0;
        (void)frexpf(grad_scale, &e);
        gscale = ldexpf(1.0f, 8 - e);
    }
    // the default trio of one-wave-per-tile steps: fp16 chains + fp16 weight-gradient GEMMs on fp16 stage pieces (the forward
    // of such a step wrote that stash: r2l_forward_rays decides by the same rule)
    const bool trio16 = split && r2l_use_trio16();
    const float* w3 = wstream_bwd + r2l_bwd32_stream_floats(n_block) + r2l_bwd16_stream_floats(n_block);
    const float* w2 = w3 + r2l_bwd3_stream_floats(n_block);
    // raised by r2l_bwd2_kernel when this step has to run on the bf16x3 kernels (range guard / the forward fell back)
    unsigned* bwd_status = reinterpret_cast<unsigned*>(const_cast<float*>(w2) + r2l_bwd2_status_offset(n_block));
    // generic mode (dL/drgb from the caller, no known scale): {gscale, 1 / gscale} are chosen on the device from max |drgb| and
    // live in words 4, 5 of the status area; every kernel of the step reads them there
    // (round 4: MSE mode as well — the step's scale is chosen on the device from the previous step's largest chain value,
    // r2l_bwd_prepare_kernel; the host-side gscale below remains the bf16x3-only trio's)
    const float* scale_dev = trio16 ? reinterpret_cast<const float*>(bwd_status + B2S_GSCALE) : nullptr;
    if (!(parts & R2L_BWD_CHAIN)) {
    } else if (variant == R2L_VARIANT_COOP16) {
        const int rc = r2l_coop16_backward(rgb, target, drgb, save_x, save_t, wstream_bwd + r2l_bwd32_stream_floats(n_block),
                                           params, n_block, grad_scale, dpre, gx, gt, sqerr_partial, N, stream);
        if (rc) return rc;
    } else if (split) {
        // one-wave-per-tile dX chain.  Default trio (MSE mode: scaled chain, values in fp16's range up to the guard): two-way fp16
        // splits, 3 fp16 products per fp32 product (r2l_bwd2.hip), stashing fp16 stage pieces for r2l_dw16.hip, with the
        // bf16x3 chain behind it as fallback (returns at once unless the status word behind the bwd2 stream was raised: range
        // guard, or the forward already fell back and left an fp32 stash); otherwise the bf16x3 chain (r2l_bwd3.hip)
        if (chain_seg && !(trio16 && r2l_use_coopf(N, n_block) && no_fallback)) {
            r2l_set_error_msg("r2l_backward_part: chain segments need the cooperative fp16 chains (r2l_chain_segments_ok_cfg) and "
                              "R2L_BWD_NOFALLBACK");
            return (int)hipErrorInvalidValue;
        }
        if (trio16) {
            if (!chain_seg || layer_hi == 2 * n_block) {  // (the first segment opens the step)
                // (a kernel, not hipMemsetAsync: a memset node captured into a hipGraph wrote a stale 16-byte pattern instead
                // of zeros on replay — ROCm 7.0 —, which sent every replayed step to the fallback kernels)
                hipLaunchKernelGGL(r2l_bwd_prepare_kernel, dim3(1), dim3(64), 0, stream, bwd_status, params + b_off_tail_w(n_block),
                                   grad_scale, target != nullptr ? 1 : 0);
                R2L_CHECK(hipGetLastError());
                if (target == nullptr) {
                    hipLaunchKernelGGL(r2l_gscale_kernel, dim3(1), dim3(1024), 0, stream, drgb, 3 * N, const_cast<float*>(scale_dev));
                    R2L_CHECK(hipGetLastError());
                }
            }
            const int rc2 = r2l_bwd2_backward(rgb, target, drgb, save_x, save_t, w2, params, n_block, grad_scale, dpre, gx, gt,
                                              sqerr_partial, N, stream, gscale, bwd_status, scale_dev,
                                              chain_seg ? layer_hi / 2 - 1 : -1, chain_seg ? layer_lo / 2 : 0);
            if (rc2) return rc2;
            if (!no_fallback) {
                // the fallback's stream is packed in front of it, and only when it will run
                // (... and an fp16 forward stash is expanded for the bf16x3 kernels: a chain-only trip, r2l_bwd3.hip)
                const int rp = r2l_bwd3_pack(params, n_block, const_cast<float*>(w3), stream, bwd_status, save_x, save_t, N);
                if (rp) return rp;
            }
        }
        if (!(trio16 && no_fallback)) {
            const int rc = r2l_bwd3_backward(rgb, target, drgb, save_x, save_t, w3, params, n_block, grad_scale, dpre, gx, gt,
                                             sqerr_partial, N, stream, gscale, trio16 ? bwd_status : nullptr, scale_dev);
            if (rc) return rc;
        }
    } else {
        R2LBwdArgs a{};
        a.rgb = rgb; a.target = target; a.drgb = drgb; a.save_x = save_x; a.save_t = save_t; a.wstream = wstream_bwd;
        a.params = params; a.n_block = n_block; a.grad_scale = grad_scale; a.dpre = dpre; a.gx = gx; a.gt = gt;
        a.sqerr_partial = sqerr_partial; a.N = N;
        const int64_t tiles = (N + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS;
        hipLaunchKernelGGL(r2l_bwd_chain_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, stream, a);
        R2L_CHECK(hipGetLastError());
    }
    // Small steps: the head / tail gradients BESIDE the body's (round 5).  At 4096 rays the body's weight-gradient kernel fills 172
    // CUs for ~105 us and its reduce 19 us, then the head's (64 workgroups, VALU-bound: 54 us + 11 us of reduce) and the tail's
    // (6 + 11 us) each run alone on a mostly idle chip.  They only share the dX chain's outputs as inputs and write disjoint ranges
    // of the flat gradient and (since round 5) disjoint regions of dw_slab, so a call that does all of them puts head + tail on a
    // second stream of the library's own between two events: fork behind the chain, join before returning — for the caller's
    // stream nothing changes (capturable: the side stream joins the capture and leaves it at the join).  R2L_NO_DW_OVERLAP=1: off.
    hipStream_t hstream = stream;
    bool overlap = false;
    // The side stream and its fork / join events belong to the calling THREAD (and device): two host threads driving the same
    // device never record or wait on each other's events; calls of one thread are issued in order, so a later
    // call's re-record cannot be seen by an earlier call's already enqueued wait.
    struct R2LJoin {  // always joins once forked — on the error returns below too (an unjoined fork would invalidate a capture
        hipStream_t from = nullptr, to = nullptr;  // in progress and let the caller's later work race with head / tail kernels)
        hipEvent_t ev = nullptr;
        int join() {
            if (from == nullptr) return 0;
            const hipStream_t f = from;
            from = nullptr;
            R2L_CHECK(hipEventRecord(ev, f));
            R2L_CHECK(hipStreamWaitEvent(to, ev, 0));
            return 0;
        }
        ~R2LJoin() { (void)join(); }
    } joiner;
    {
        static thread_local hipStream_t side[16] = {nullptr};
        static thread_local hipEvent_t ev_fork[16], ev_join[16];
        static const bool overlap_off = r2l_env_on("R2L_NO_DW_OVERLAP");
        const int64_t overlap_max = r2l_dw_overlap_max();
        int dev = 0;
        if (!overlap_off && (parts & R2L_BWD_BODY) && (parts & R2L_BWD_HEAD) && layer_hi > layer_lo && dw_slab != nullptr &&
            N <= overlap_max && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16) {
            if (side[dev] == nullptr) {
                hipStream_t st = nullptr;
                R2L_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
                R2L_CHECK(hipEventCreateWithFlags(&ev_fork[dev], hipEventDisableTiming));
                R2L_CHECK(hipEventCreateWithFlags(&ev_join[dev], hipEventDisableTiming));
                side[dev] = st;  // (last: a failed creation above leaves the slot empty and the next call retries)
            }
            R2L_CHECK(hipEventRecord(ev_fork[dev], stream));
            R2L_CHECK(hipStreamWaitEvent(side[dev], ev_fork[dev], 0));
            hstream = side[dev];
            overlap = true;
            joiner.from = hstream; joiner.to = stream; joiner.ev = ev_join[dev];
        }
    }
    // 2. body weight gradients
    if ((parts & R2L_BWD_BODY) && layer_hi > layer_lo) {
        R2LDwArgs a{};
        a.save_x = save_x; a.save_t = save_t; a.gx = gx; a.gt = gt; a.grads = grads; a.n_block = n_block; a.N = N;
        a.layer0 = layer_lo; a.n_layers = layer_hi - layer_lo;
        a.units_per_layer = (N + DW_CHUNK - 1) / DW_CHUNK;
        const int64_t total = a.units_per_layer * a.n_layers;
        int64_t wgs = n_cu < DW_MAX_WGS ? n_cu : DW_MAX_WGS;
        // small steps: two workgroups per layer, none across a layer boundary (one slab flush each, half the reduce): measured
        // at 4096 rays 97 + 16 us against 111 + 22 us for 251 workgroups; at 12 288 rays the full grid wins again (229 + 22
        // against 242 + 16)
        // (that is the fp16 trio's kernel, which is bound by the slab traffic at this size; the fp32-MFMA / bf16x3 kernels are bound by
        // their MFMAs — 32 units on 172 workgroups against 22 on 251 — and keep the full grid: round 6, profiles/r06_graded_step_ab.txt E)
        if (trio16 && N <= 6144 && 2 * (int64_t)a.n_layers <= wgs) wgs = 2 * (int64_t)a.n_layers;
        // above that, up to the largest step whose head / tail gradients run beside this kernel: 11/16 of the CUs.  On the full grid the
        // head kernel (VALU-bound, 4 workgroups per ray slice) queues behind the persistent workgroups and the overlap is one in name
        // only (12 288 rays: 1.251 ms with 251 workgroups = 1.285 with the overlap off; 1.231 with 176, 1.239 with 144: round 6,
        // profiles/r06_small_step_dw_grid.txt); the kernel is HBM-bound, fewer workgroups cost it little
        if (trio16 && N > 6144 && N <= r2l_dw_overlap_max() && wgs > n_cu * 11 / 16) wgs = n_cu * 11 / 16;
        // the MFMA-bound kernels with the head / tail gradients beside them (small steps): an eighth of the CUs stays free for those,
        // or they queue behind the persistent grid (4096 rays, fp32 family: 1.358 ms with 172 workgroups, 1.367 with 251, 1.317 with 224)
        // (decided by the step size alone, not by whether THIS call overlaps: the staged form — body buckets in calls of their own —
        // must cut the same work list as the one-call form, tests: staged with one bucket == one call, bit for bit)
        if (!trio16 && N <= r2l_dw_overlap_max() && wgs > n_cu - n_cu / 8) wgs = n_cu - n_cu / 8;
        if (const char* e = getenv("R2L_DW_WGS")) {  // tuning knob (tools/small_prof.sh)
            const int64_t v = atoll(e);
            if (v >= a.n_layers && v <= wgs) wgs = v;
        }
        if (wgs > total) wgs = total;
        a.units_per_wg = (total + wgs - 1) / wgs;
        wgs = (total + a.units_per_wg - 1) / a.units_per_wg;
        // units_per_wg <= units_per_layer whenever wgs >= n_layers (always, for n_block <= 128): a range touches <= 2 layers
        a.slab = (a.units_per_wg <= a.units_per_layer) ? dw_slab : nullptr;
        // bf16 matrix pipe at fp32 accuracy: operands split once per workgroup from the chunked stash of the bf16x3 chains
        // (r2l_dw_body3c), or per wave from the row-major stash of the other chains (R2L_NO_FWD3: fp32 MFMA)
        // chunked stash: the gradient operands carry the chain's power-of-two scale
        if (split) a.unscale = 1.0f / gscale;
        a.scale_dev = scale_dev;
        // default trio: one fp16 product per fp32 product on the fp16 stage pieces the chains stashed (r2l_dw16.hip); when the
        // dX chain raised its status word (this step fell back to the bf16x3 chains and their fp32 stash) it returns at once
        // and the bf16x3 kernel behind it does the work
        if (trio16) {
            // (exact mode: the chains of this step stashed the mid halves too — the same config / environment as the forward)
            a.mid_off = r2l_dw_exact() ? (unsigned)R2L_H16_MID_BYTES(R2L_PAD_ROWS(N)) : 0u;
            a.act_scale = save_x + R2L_STASH_FMT_WORD(n_block, R2L_PAD_ROWS(N)) + 1;  // the scale the forward stashed x, relu(t) at
            const int rc = r2l_dw16_launch(a, wgs, bwd_status, stream);
            a.mid_off = 0u;
            a.act_scale = nullptr;
            if (rc) return rc;
            a.run_if = bwd_status;
            if (!no_fallback) hipLaunchKernelGGL(r2l_dw_body3c_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, a);
        } else if (split) {
            hipLaunchKernelGGL(r2l_dw_body3c_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, a);
        }
        else if (r2l_use_fwd3()) hipLaunchKernelGGL(r2l_dw_body3_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(r2l_dw_body_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, a);
        R2L_CHECK(hipGetLastError());
        if (a.slab != nullptr) {
            hipLaunchKernelGGL(r2l_dw_reduce_kernel, dim3((DW_SLAB_FLOATS / 4 + 255) / 256, a.n_layers), dim3(256), 0, stream,
                               a.slab, grads, a.units_per_layer, a.units_per_wg, wgs, a.layer0);
            R2L_CHECK(hipGetLastError());
        }
    }
    // 3. head weight gradient
    if (parts & R2L_BWD_HEAD) {
        R2LDwHeadArgs a{};
        a.rays_o = rays_o; a.rays_d = rays_d; a.t_rand = t_rand; a.ztab = ztab; a.emb = emb; a.gh = gx; a.grads = grads; a.N = N;
        int64_t slices = n_cu / 4;
        if (slices > DW_HEAD_SLAB_MAX / (R2L_W * 1024)) slices = DW_HEAD_SLAB_MAX / (R2L_W * 1024);  // (what the slab region holds)
        // small launches: >= 256 rays per slice (each slice costs a 1 MB partial).  (Round 5 tried 128 and 64 rays per slice for the
        // 4096-ray step — 32 / 64 slices instead of 16: 0.789 / 0.823 ms per step against 0.789, same box: what the wider grid gains
        // the 1 MB-per-slice reduce gives back; profiles/r05_small_step_ab.txt)
        if (slices > (N + 255) / 256) slices = (N + 255) / 256;
        if (slices < 1) slices = 1;
        int64_t per = (N + slices - 1) / slices;
        per = (per + 1) & ~(int64_t)1;  // even: a k-step pairs rays 2s, 2s+1
        if (per < 2) per = 2;
        slices = (N + per - 1) / per;
        a.rays_per_wg = per;
        // per-slice partials go to the head's region of dw_slab, else to the (by now dead) gt scratch when
        // it is large enough, else fp32 atomics
        const int64_t slab_floats = slices * (int64_t)(R2L_W * 1024);
        const int64_t gt_floats = (int64_t)n_block * R2L_PAD_ROWS(N) * R2L_W;
        if (slices > 1 && dw_slab != nullptr && slab_floats <= DW_HEAD_SLAB_MAX) a.slab = dw_slab + DW_BODY_SLAB;
        // (never `gt` beside the body kernels of an overlapped call: they are still reading it on the other stream)
        else a.slab = (slices > 1 && slab_floats <= gt_floats && !overlap) ? gt : nullptr;
        const dim3 hg((unsigned)(slices * 4)), hb(256);
        if (trio16 && emb == nullptr) {
            // default trio: the same GEMM on the fp16 matrix pipe (r2l_dw_head16.hip); the fp32 kernel behind it runs only when
            // the dX chain raised its status word (range guard / bf16x3 fallback step)
            a.gscale = gscale;
            a.unscale = 1.0f / gscale;
            a.scale_dev = scale_dev;
            a.run_unless = bwd_status;
            a.exact = r2l_dw_exact();
            const int rc = r2l_dw_head16_launch(a, slices, hstream);
            if (rc) return rc;
            a.run_unless = nullptr;
            a.run_if = bwd_status;
        }
        if (trio16 && emb == nullptr && no_fallback) {
        } else if (emb != nullptr) hipLaunchKernelGGL((r2l_dw_head_kernel<true, false>), hg, hb, 0, hstream, a);
        else if (t_rand != nullptr) hipLaunchKernelGGL((r2l_dw_head_kernel<false, true>), hg, hb, 0, hstream, a);
        else hipLaunchKernelGGL((r2l_dw_head_kernel<false, false>), hg, hb, 0, hstream, a);
        R2L_CHECK(hipGetLastError());
        if (a.slab) {
            hipLaunchKernelGGL(r2l_head_reduce_kernel, dim3(R2L_W * 1024 / 256), dim3(256), 0, hstream, a.slab,
                               (int)slices, grads);
            R2L_CHECK(hipGetLastError());
        }
    }
    // 4. tail gradients
    if (parts & R2L_BWD_TAIL) {
        int64_t wgs = 2 * n_cu;
        if (wgs > DW_TAIL_SLAB / (4 * R2L_W)) wgs = DW_TAIL_SLAB / (4 * R2L_W);  // (what the slab region holds: no switch of paths,
        int64_t per = (N + wgs - 1) / wgs;                                      //  i.e. of summation order, on a larger device)
        if (per < 1) per = 1;
        wgs = (N + per - 1) / per;
        // partials in the tail's region of dw_slab; summed in workgroup order
        float* part = (dw_slab != nullptr && wgs * (4 * R2L_W) <= DW_TAIL_SLAB) ? dw_slab + DW_TAIL_SLAB_BASE : nullptr;
        // (chunked stash: slot n of save_x holds y = x_n + x_0)
        hipLaunchKernelGGL(r2l_dw_tail_kernel, dim3((unsigned)wgs), dim3(256), 0, hstream, dpre, split ? nullptr : save_x,
                           save_x + (int64_t)n_block * (split ? R2L_TRIO_SLOT(R2L_PAD_ROWS(N)) : R2L_PAD_ROWS(N) * (int64_t)R2L_W),
                           grads, part, n_block, N, per);
        R2L_CHECK(hipGetLastError());
        if (part != nullptr) {
            hipLaunchKernelGGL(r2l_tail_reduce_kernel, dim3(R2L_W / 32, 4), dim3(256), 0, hstream, part, wgs, grads, n_block);
            R2L_CHECK(hipGetLastError());
        }
    }
    return joiner.join();  // the caller's stream continues when head + tail are done as well
}
