// r2l_coopf_bwd.hip — cooperative fp16x2 dX chain (see r2l_coopf.h; the chain itself is r2l_bwd2.hip's): dL/drgb -> sigmoid'
// -> tail^T -> the 2*n_block transposed layers, stashing the fp16 hi operands of g and of the masked u for r2l_dw16.hip and
// writing gx[0] = dL/d(head pre-activation) for the head gradient — the dX half of loss.backward()
// (/root/reference/main.py:1377-1404) for launches of a few thousand rays.  One 32-ray tile per workgroup; wave w owns output
// tiles 2w, 2w+1 of every transposed layer; the stream is r2l_pack_bwd2_kernel's (per block: zero stage, 16 stages of W2^T,
// zero stage, 16 stages of W1^T).
#include "r2l_coopf.h"
#include <type_traits>

__host__ __device__ static inline int64_t cb_off_tail_w(int n_block) {
    return (int64_t)R2L_IN * R2L_W + R2L_W + (int64_t)2 * n_block * (R2L_W * R2L_W + R2L_W);
}

struct CfBwdArgs {
    const float* rgb;
    const float* target;
    const float* drgb;
    const float* save_x;
    const float* save_t;
    const unsigned char* stream;  // bwd2 stage stream
    const float* params;
    int n_block;
    float grad_scale;
    float gscale, ginv;
    const float* scale_dev;  // generic mode: {gscale, 1 / gscale} on the device, or nullptr
    unsigned* status;        // range guard (r2l_bwd2.hip's word)
    const unsigned* fmt;     // stash format word of the forward: != 0 -> this step belongs to the bf16x3 kernels
    float* dpre;
    float* gx;
    float* gt;
    float* sqerr_partial;
    int64_t N;
    int64_t mid_units;  // != 0: mid quads stashed this many 16-byte units behind the hi pieces (exact weight gradients)
    // One launch walks the blocks b_start, b_start - 1, .., b_end (whole chain: n_block - 1 .. 0).  A data-parallel host may cut
    // the chain into several launches (r2l_backward_part: R2L_BWD_CHAIN with a layer range) so that the weight gradients — and
    // the gradient exchange — of the blocks already walked run beside the rest of the chain on the CUs a small launch leaves
    // idle: g then crosses the launch boundary as fp32 fragments in the tile's row-major area of gx slot 0 (which only the
    // last launch finally fills), every other value takes the path it takes in one launch: results are bit-identical.
    int b_start, b_end;
    int n_two;  // mixed launch: workgroups 0 .. n_two - 1 (in fc_mixed_index order) take two ray tiles, the others one
    int xcd_major;
};

// masked u: bit tt*16 + c of the forward's mask word of this wave
struct CfMask {
    unsigned w;
    __device__ __forceinline__ float operator()(float v, int tt, int c) const { return ((w >> (tt * 16 + c)) & 1u) ? v : 0.f; }
};

// The dX chain of one workgroup on the NT ray tiles tile0 .. tile0 + NT - 1; bop_lds = LDS byte address of the B-operand images
// [g | masked u][ray tile][16 stages x (hi, mid) x 1 KiB] (the kernel's allocation: the mixed launch below runs both bodies)
template <int NT, bool MID>
__device__ __forceinline__ void cf_bwd_body(const CfBwdArgs& a, const int64_t tile0, const unsigned bop_lds) {
    if (__builtin_nontemporal_load(a.fmt) != 0u) {  // the forward's stash is the bf16x3 trio's: so is this step's backward
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.status, 1u);
        return;
    }
    const float gscale = a.scale_dev != nullptr ? a.scale_dev[0] : a.gscale;
    const float ginv = a.scale_dev != nullptr ? a.scale_dev[1] : a.ginv;
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t n_tiles = (a.N + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS;
    // ray tiles of this workgroup; one past the end (odd tile count, NT = 2) recomputes the last live tile (same values to the
    // same addresses)
    int64_t tile[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
        tile[rt] = tile0 + rt;
        if (tile[rt] > n_tiles - 1) tile[rt] = n_tiles - 1;
    }
    const int64_t Np = R2L_PAD_ROWS(a.N);
    const int64_t slot = R2L_TRIO_SLOT(Np);
    const float* tw = a.params + cb_off_tail_w(a.n_block);
    const bool first = a.b_start == a.n_block - 1, final = a.b_end == 0;

    // ---- loss gradient through the sigmoid, per-tile squared error (every wave computes it; wave 0 writes) -------------------
    float dp[NT][3];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
        const int64_t ray = tile[rt] * R2L_TILE_RAYS + j;
        const bool valid = ray < a.N;
        const int64_t rc = valid ? ray : a.N - 1;
        float se = 0.f;
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
            dp[rt][c] = valid ? dl * (r * (1.0f - r)) : 0.f;
        }
        if (!valid) se = 0.f;
        if (wave == 0 && first) {
            if (valid && h == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) a.dpre[ray * 3 + c] = dp[rt][c];
            }
            if (a.sqerr_partial != nullptr) {
                float s = (h == 0) ? se : 0.f;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
                if (lane == 0) a.sqerr_partial[tile[rt]] = s;
            }
        }
    }
    // dy = Wt^T dpre (tail Linear(256,3)): register c = 4q + e of this wave's output tile tt
    auto tail_t = [&](int rt, int tt, int q, f32x4& out) {
        const int T = 2 * wave + tt;
        f32x4 wv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) wv[c] = *reinterpret_cast<const f32x4*>(tw + c * R2L_W + 32 * T + 8 * q + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = wv[0][e] * (dp[rt][0] * gscale);
            v = __builtin_fmaf(wv[1][e], dp[rt][1] * gscale, v);
            v = __builtin_fmaf(wv[2][e], dp[rt][2] * gscale, v);
            out[e] = v;
        }
    };
    f32x16 g[NT][2], u[NT][2];
    // this lane's 32 fragment values of a tile where they cross a launch boundary (gx slot 0, the tile's 32 KiB)
    auto carry_at = [&](int rt) { return a.gx + tile[rt] * (R2L_TILE_RAYS * R2L_W) + (wave * 64 + lane) * 32; };
    if (first) {
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
                    tail_t(rt, tt, q, v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[rt][tt][4 * q + e] = v[e];
                }
    } else {
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(carry_at(rt) + 16 * tt + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[rt][tt][4 * q + e] = v[e];
                }
    }

    constexpr int R = FcRingOf<NT>::value;
    FcRing<R> W;
    FcStream P;
    {
        const unsigned long long sa = (unsigned long long)a.stream;
        P.rs = u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sa),
                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sa >> 32)) & 0xffffu, 0xffffffffu, 0x00020000u};
        P.voff = (unsigned)lane * 16u + (unsigned)wave * 2048u;
#ifdef FC_LATE_MODE
        P.late = blockIdx.x >= 256;
#endif
        P.g = (unsigned)(a.n_block - 1 - a.b_start) * 34u;  // 34 stages per block: [zero, 16 of W2^T, zero, 16 of W1^T]
    }
#pragma unroll
    for (int k = 0; k < R; ++k) fc_issue(W.a[k], P);
    f16x8 ones;
#pragma unroll
    for (int k = 0; k < 8; ++k) ones[k] = (_Float16)((h == 0 && k < 2) ? 1.0f : 0.0f);
    constexpr unsigned KIND = NT * FC_BOP_BYTES;
    const unsigned bop_rd = bop_lds + (unsigned)lane * 16u;
    const unsigned bop_wr = bop_lds + (unsigned)lane * 16u + (unsigned)wave * 8192u;
    float amax = 0.f;

    // stash bases of this lane / wave (fp16 stage pieces 4w .. 4w+3 of the tile) and the forward's mask word of the wave
    u32x4* gxh[NT];
    u32x4* gth[NT];
    const unsigned* mwp[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
        const int64_t unit0 = tile[rt] * R2L_H16_TILE_UNITS + lane + 256 * wave;
        gxh[rt] = reinterpret_cast<u32x4*>(a.gx + (int64_t)(a.b_start + 1) * slot) + unit0;  // slot b + 1 of block b = b_start
        gth[rt] = reinterpret_cast<u32x4*>(a.gt + (int64_t)a.b_start * slot) + unit0;
        mwp[rt] = reinterpret_cast<const unsigned*>(a.save_t + (int64_t)a.b_start * slot + R2L_MASK_OFFSET(Np) +
                                                    tile[rt] * 256 + lane * 4) + wave;
    }

    // g B operands in the kind-0 images, masked-u B operands in the kind-1 images; a barrier after each production
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) fc_produce<false, true, false, FcIdentity, MID>(g[rt], bop_wr + (unsigned)rt * FC_BOP_BYTES, gxh[rt], nullptr, amax, FcIdentity(), a.mid_units);
    fc_barrier();
    auto block = [&](auto ph_tag, bool last) {
        constexpr int PH = decltype(ph_tag)::value;
        // the forward's mask words: untracked loads like the ring's (a compiler-tracked one would drain vmcnt, i.e. the ring, at
        // its first use).  They are older than every load of GEMM A (68), whose last fc_wait leaves only the 4 (R - 1) youngest in flight.
        unsigned mw[NT];
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) asm volatile("global_load_dword %0, %1, off" : "=&v"(mw[rt]) : "v"(mwp[rt]) : "memory");
        // GEMM A: u = W2^T g   (zero stage first: u is initialised by C = 0)
        fc_layer<PH, true>(u, W, P, bop_rd, ones);
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
            asm volatile("" : "+v"(mw[rt]));  // (uses of mw stay behind GEMM A)
            fc_produce<false, true, false, CfMask, MID>(u[rt], bop_wr + KIND + (unsigned)rt * FC_BOP_BYTES, gth[rt], nullptr, amax, CfMask{mw[rt]}, a.mid_units);
        }
        fc_barrier();
        // GEMM B: g += W1^T (u . mask)
        fc_layer<(PH + 1) % R, false>(g, W, P, bop_rd + KIND, ones);
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
            gxh[rt] -= slot / 4;
            gth[rt] -= slot / 4;
            mwp[rt] -= slot;
        }
        if (!last) {
#pragma unroll
            for (int rt = 0; rt < NT; ++rt)
                fc_produce<false, true, false, FcIdentity, MID>(g[rt], bop_wr + (unsigned)rt * FC_BOP_BYTES, gxh[rt], nullptr, amax, FcIdentity(), a.mid_units);
            fc_barrier();
        }
    };
    // the k-th block of the walk starts in phase 2 k mod R: unrolled over R / 2 blocks
#pragma unroll 1
    for (int b = a.b_start; b >= a.b_end; b -= R / 2) {
        block(std::integral_constant<int, 0>{}, b == a.b_end);
        if constexpr (R >= 4) {
            if (b - 1 >= a.b_end) block(std::integral_constant<int, 2 % R>{}, b - 1 == a.b_end);
        }
        if constexpr (R >= 8) {
            if (b - 2 >= a.b_end) block(std::integral_constant<int, 4 % R>{}, b - 2 == a.b_end);
            if (b - 3 >= a.b_end) block(std::integral_constant<int, 6 % R>{}, b - 3 == a.b_end);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    f2_report_amax<B2S_AMAX>(a.status, amax, lane);  // AMAX (the next step's scale is chosen from it), FLAG if out of range

    if (!final) {  // the chain continues in the next launch
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4*>(carry_at(rt) + 16 * tt + 4 * q) =
                        f32x4{g[rt][tt][4 * q], g[rt][tt][4 * q + 1], g[rt][tt][4 * q + 2], g[rt][tt][4 * q + 3]};
        return;
    }

    // ---- head: dL/d(head pre-activation) = (g + dy) * (x_0 > 0) -> gx[0], row-major fp32 (the head weight gradient reads rows) -----
    // (dy recomputed: the same operations on the same values as the chain's seed)
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
        // x_0's fp16 stage pieces (slot 0 of save_x): stage kb = 2T + r holds fragment registers c = 8r .. 8r+7 of tile T
        const u32x4* r = reinterpret_cast<const u32x4*>(a.save_x) + tile[rt] * R2L_H16_TILE_UNITS + lane;
        float* o = a.gx + (tile[rt] * R2L_TILE_RAYS + j) * R2L_W + 4 * h;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int T = 2 * wave + tt;
                const f16x8 xv = __builtin_bit_cast(f16x8, r[64 * (2 * T + rr)]);
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    f32x4 dyv, ov;
                    tail_t(rt, tt, 2 * rr + q2, dyv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = 8 * rr + 4 * q2 + e;
                        ov[e] = (float)xv[4 * q2 + e] > 0.f ? (g[rt][tt][c] + dyv[e]) * ginv : 0.f;
                    }
                    *reinterpret_cast<f32x4*>(o + 32 * T + 8 * (2 * rr + q2)) = ov;
                }
            }
    }
}

template <int NT, bool MID = false>
__global__ __launch_bounds__(256, NT == 1 ? 2 : 1) void r2l_coopf_bwd_kernel(const CfBwdArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char bop[2][NT][FC_BOP_BYTES];
    cf_bwd_body<NT, MID>(a, (int64_t)blockIdx.x * NT, (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)&bop[0][0][0]);
}

// MIXED launch (r2l_coopf_fwd.hip r2l_coopf_fwd_mixed_kernel: same tile -> workgroup map, so a tile's stash is read by the
// role that wrote it — not that it matters: every tile takes the same path in either role)
template <bool MID>
__global__ __launch_bounds__(256, 1) void r2l_coopf_bwd_mixed_kernel(const CfBwdArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char bop[2][2][FC_BOP_BYTES];
    // (128 KiB: at most one workgroup per CU — the one-tile body must not share a SIMD with a second wave, r2l_coopf.h)
    const unsigned bop_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)&bop[0][0][0];
    const int b = fc_mixed_index(a.xcd_major);
    if (b < a.n_two) cf_bwd_body<2, MID>(a, (int64_t)2 * b, bop_lds);
    else cf_bwd_body<1, MID>(a, (int64_t)a.n_two + b, bop_lds);
}

int r2l_coopf_backward(const float* rgb, const float* target, const float* drgb, const float* save_x, const float* save_t,
                       const float* wstream_bwd2, const float* params, int n_block, float grad_scale, float* dpre, float* gx,
                       float* gt, float* sqerr_partial, int64_t N, hipStream_t stream, float gscale, unsigned* status,
                       const float* scale_dev, int b_start, int b_end) {
    CfBwdArgs a{};
    a.b_start = b_start < 0 ? n_block - 1 : b_start;
    a.b_end = b_end;
    a.status = status;
    a.scale_dev = scale_dev;
    a.fmt = reinterpret_cast<const unsigned*>(save_x) + R2L_STASH_FMT_WORD(n_block, R2L_PAD_ROWS(N));
    a.gscale = gscale; a.ginv = 1.0f / gscale;
    a.rgb = rgb; a.target = target; a.drgb = drgb; a.save_x = save_x; a.save_t = save_t;
    a.stream = reinterpret_cast<const unsigned char*>(wstream_bwd2); a.params = params; a.n_block = n_block;
    a.grad_scale = grad_scale; a.dpre = dpre; a.gx = gx; a.gt = gt; a.sqerr_partial = sqerr_partial; a.N = N;
    a.mid_units = r2l_dw_exact() ? R2L_H16_MID_BYTES(R2L_PAD_ROWS(N)) / 16 : 0;
    const int64_t tiles = (N + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS;
    static int solo_ok[2] = {0, 0};
    if (const int n_two = r2l_coopf_mixed_two(tiles); n_two > 0) {  // between one and two tiles per CU: one workgroup on every CU
        a.n_two = n_two;
        a.xcd_major = r2l_coopf_mixed_xcd_major();
        const dim3 grid((unsigned)(tiles - n_two));
        if (a.mid_units != 0) hipLaunchKernelGGL((r2l_coopf_bwd_mixed_kernel<true>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((r2l_coopf_bwd_mixed_kernel<false>), grid, dim3(256), 0, stream, a);
    } else if (a.mid_units != 0) {  // exact weight gradients: the mid halves of g / masked u are stashed too
        if (r2l_coopf_two_tiles(tiles))
            hipLaunchKernelGGL((r2l_coopf_bwd_kernel<2, true>), dim3((unsigned)((tiles + 1) / 2)), dim3(256), 0, stream, a);
        else {
            if (int e = fc_check_solo(r2l_coopf_bwd_kernel<1, true>, "r2l_coopf_bwd_kernel<1, mid>", &solo_ok[1])) return e;
            hipLaunchKernelGGL((r2l_coopf_bwd_kernel<1, true>), dim3((unsigned)tiles), dim3(256), FC_SOLO_LDS_BYTES, stream, a);
        }
    } else if (r2l_coopf_two_tiles(tiles)) {
        hipLaunchKernelGGL((r2l_coopf_bwd_kernel<2>), dim3((unsigned)((tiles + 1) / 2)), dim3(256), 0, stream, a);
    } else {
        if (int e = fc_check_solo(r2l_coopf_bwd_kernel<1>, "r2l_coopf_bwd_kernel<1>", &solo_ok[0])) return e;
        hipLaunchKernelGGL((r2l_coopf_bwd_kernel<1>), dim3((unsigned)tiles), dim3(256), FC_SOLO_LDS_BYTES, stream, a);
    }
    R2L_CHECK(hipGetLastError());
    return 0;
}
