// r2l_bwd2.hip — the dX chain of r2l_bwd3.hip on the fp16 matrix pipe with two-way operand splits (machinery: r2l_f2.h):
// three fp16 products per fp32 product, ~2^-21 relative.  The chain runs on gscale * g (a power of two chosen from the MSE
// gradient scale, r2l_backward), which puts its values into fp16's range; a range guard raises a status word and the bf16x3
// kernel launched behind it redoes the launch.  Stage order, mask ring and stash stores are r2l_bwd3.hip's; six 16 KiB
// weight buffers fit beside the 32 KiB mask ring.
#include "r2l_f2.h"
#include "r2l_coopf.h"

#define B3_RING 2  // mask pieces (one per block) per wave

__host__ __device__ static inline int64_t b2_off_body_w(int layer) {
    return (int64_t)R2L_IN * R2L_W + R2L_W + (int64_t)layer * (R2L_W * R2L_W + R2L_W);
}
__host__ __device__ static inline int64_t b2_off_tail_w(int n_block) { return b2_off_body_w(2 * n_block); }

// =================================================================================================================
// pack: stage order and element numbering of r2l_pack_bwd2_kernel, two fp16 splits per value, 16 KiB stages
// =================================================================================================================
__global__ void r2l_pack_bwd2_kernel(const float* __restrict__ params, unsigned short* __restrict__ out, int n_block) {
    const int64_t stages = r2l_bwd3_stages(n_block);
    const int64_t total = (stages + R2L_F3_PAD_STAGES) * 8 * 64 * 8;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(idx & 7), lane = (int)((idx >> 3) & 63), tile = (int)((idx >> 9) & 7);
        const int64_t g = idx >> 12;
        const int i = lane & 31, h = lane >> 5, o = 32 * tile + i;
        unsigned short* st = out + g * (F2_STAGE_BYTES / 2);
        unsigned short v0 = 0, v1 = 0;
        if (g < stages) {
            const int slot = (int)(g / 34), r = (int)(g % 34), b = n_block - 1 - slot;
            if (r != 0 && r != 17) {
                const int layer = r < 17 ? 2 * b + 1 : 2 * b, kb = r < 17 ? r - 1 : r - 18;
                const int T = kb >> 1, rr = kb & 1;
                const int in = 32 * T + 8 * (2 * rr + (s >> 2)) + 4 * h + (s & 3);
                const float w = params[b2_off_body_w(layer) + (int64_t)in * R2L_W + o];  // (W^T)[o][in] = W[in][o]
                const _Float16 hi = (_Float16)w;
                const _Float16 mid = (_Float16)(w - (float)hi);
                v0 = __builtin_bit_cast(unsigned short, hi);
                v1 = __builtin_bit_cast(unsigned short, mid);
            }
        }
        const int64_t e = ((int64_t)tile * 64 + lane) * 8 + s;
        st[e] = v0;
        st[8 * 64 * 8 + e] = v1;
    }
}

// =================================================================================================================
// kernel
// =================================================================================================================
struct B2Args {
    const float* rgb;
    const float* target;
    const float* drgb;
    const float* save_x;
    const float* save_t;
    const unsigned char* stream;
    const float* params;
    int n_block;
    float grad_scale;
    float gscale, ginv;  // the chain runs on gscale * g (a power of two; what it stashes is scaled), gx[0] is scaled back
    const float* scale_dev;  // generic mode: {gscale, 1 / gscale} chosen on the device from max |drgb| (r2l_gscale_kernel), or nullptr
    unsigned* status;    // range guard: raised when a chain value leaves fp16's safe range (the bf16x3 kernel then redoes it)
    const unsigned* fmt; // stash format word of the forward (r2l_common.h): != 0 -> chunked fp32 stash (the forward fell back to
                         // the bf16x3 kernel): this launch raises *status and leaves the work to the bf16x3 chain as well
    float* dpre;
    float* gx;
    float* gt;
    float* sqerr_partial;
    int64_t N;
    unsigned stash_mid;  // != 0: also stash the mid halves of g / masked u, this many bytes behind the hi pieces (r2l_f2.h)
};

// gatherers: four B values of the next stage (+ their stash store)
// (what the weight-gradient GEMMs read of g / the masked u is the fp16 hi operand these values are converted to: stashed by
// the stage that assembles it, r2l_f2.h F2Hst — gx[b+1] and gt[b] hold fp16 stage pieces, scaled by gscale)
struct B3TakeG {  // g values (identity)
    const f32x16& frag;
    int c0;
    __device__ __forceinline__ void operator()(float (&v)[4]) const {
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = frag[c0 + s];
    }
};
struct B3TakeU {  // u values masked by relu'(t_b) (mask words of the forward: bit (T&1)*16 + c of word T>>1)
    const f32x16& frag;
    int c0;
    int T;
    const u32x4& mb;
    __device__ __forceinline__ void operator()(float (&v)[4]) const {
        const unsigned w = mb[T >> 1] >> ((T & 1) * 16 + c0);
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = ((w >> s) & 1u) ? frag[c0 + s] : 0.f;
    }
};

// MID: the mid halves of g / masked u are stashed too (exact weight gradients; template parameter: see r2l_fwd2.hip)
template <bool MID>
__global__ __launch_bounds__(256, 1) void r2l_bwd2_kernel(const B2Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[F2_NBUF][F2_STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char mring[4][B3_RING][1024];

    if (__builtin_nontemporal_load(a.fmt) != 0u) {  // the forward's stash is the bf16x3 trio's: so is this step's backward
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.status, 1u);
        return;
    }
    const float gscale = a.scale_dev != nullptr ? a.scale_dev[0] : a.gscale;
    const float ginv = a.scale_dev != nullptr ? a.scale_dev[1] : a.ginv;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // a wave whose tile lies past the end recomputes the last live tile (identical values to identical addresses): nothing in
    // the chain is conditional
    const int64_t n_tiles = (a.N + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile > n_tiles - 1) tile = n_tiles - 1;
    const int64_t ray = tile * R2L_TILE_RAYS + (lane & 31);
    const bool valid = ray < a.N;
    const int64_t rc = valid ? ray : a.N - 1;
    const int64_t Np = R2L_PAD_ROWS(a.N);

    // ---- loss gradient through the sigmoid, per-tile squared error (as r2l_bwd_chain_kernel) ---------------------------
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
    if (a.sqerr_partial != nullptr) {
        float s = (h == 0) ? se : 0.f;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) a.sqerr_partial[tile] = s;
    }
    // g = dy = Wt^T dpre   (tail Linear(256,3))
    f32x16 g[R2L_NT], u[R2L_NT], dy[R2L_NT];
    {
        const float* tw = a.params + b2_off_tail_w(a.n_block);
#pragma unroll
        for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 wv[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) wv[c] = *reinterpret_cast<const f32x4*>(tw + c * R2L_W + 32 * T + 8 * q + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = wv[0][j] * (dp[0] * gscale);
                    v = __builtin_fmaf(wv[1][j], dp[1] * gscale, v);
                    v = __builtin_fmaf(wv[2][j], dp[2] * gscale, v);
                    g[T][4 * q + j] = v;
                    dy[T][4 * q + j] = v;
                }
            }
    }

    // ---- weight staging (6 buffers of 16 KiB beside the 32 KiB mask ring) ------------------------------------------------------
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

    // mask words of this wave's tile: one 1 KiB piece per block (64 lanes x 16 B), fetched by DMA into a two-slot LDS ring at
    // the start of the block and read when its second GEMM starts, 17 stages later
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)&mring[0][0][0] +
                              (unsigned)wave * (B3_RING * 1024u);
    const unsigned char* ring_lane = &mring[0][0][0] + wave * (B3_RING * 1024) + lane * 16;
    const int64_t slot = R2L_TRIO_SLOT(Np);
    const int64_t lane_unit = tile * R2L_H16_TILE_UNITS + lane;  // this lane's 16-byte unit of stage piece 0 in a slot
    const unsigned mvoff = (unsigned)((R2L_MASK_OFFSET(Np) + tile * 256 + lane * 4) * 4);  // byte offset of the lane's mask words

#pragma unroll 1
    for (int b = a.n_block - 1; b >= 0; --b) {
        // descriptor of save_t[b] (per block: 32-bit offsets inside the slot)
        const unsigned long long ta = (unsigned long long)(a.save_t + (int64_t)b * slot);
        const u32x4 trs = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ta),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ta >> 32)) & 0xffffu, 0xffffffffu,
                           0x00020000u};
        float* const gxr = a.gx + (int64_t)(b + 1) * slot;
        float* const gtr = a.gt + (int64_t)b * slot;
        const unsigned hvoff = (unsigned)(lane_unit * 16);
        const F3Dma no_dma{false, u32x4{0u, 0u, 0u, 0u}, 0u, 0u, 0u};
        constexpr bool mid = MID;
        const F3Dma mask_dma{true, trs, mvoff, 0u, ring_lds + (unsigned)(b & 1) * 1024u};
        // GEMM A: u = W2^T g.  stage 0 (zero stage, zero-initialises u) gathers g block 0; stage 1+kb gathers g block kb+1
        f2_stage<true, true, false>(u, P, B3TakeG{g[0], 0}, B3TakeG{g[0], 4}, mask_dma, no_dma, F2Hst{true, gxr, hvoff, 0u});
#pragma unroll
        for (int kb = 0; kb < 15; ++kb)
            f2_stage<false, false, false>(u, P, B3TakeG{g[(kb + 1) >> 1], 8 * ((kb + 1) & 1)},
                                          B3TakeG{g[(kb + 1) >> 1], 8 * ((kb + 1) & 1) + 4}, no_dma, no_dma,
                                          F2Hst{true, gxr, hvoff, 1024u * (unsigned)(kb + 1)},
                                          F2Hst{mid, gxr, hvoff, 1024u * (unsigned)kb + a.stash_mid});
        // (the mask piece was requested 16 stages ago: every stage wait since has retired all but the newest loads)
        const u32x4 mb = *reinterpret_cast<const u32x4*>(ring_lane + (b & 1) * 1024);
        f2_stage<false, false, true>(u, P, F3None{}, F3None{}, no_dma, no_dma, F2Hst{false, nullptr, 0u, 0u},
                                     F2Hst{mid, gxr, hvoff, 1024u * 15u + a.stash_mid});
        // GEMM B: g += W1^T (u . mask).  stage 17 (zero stage) gathers masked-u block 0; stage 18+kb gathers block kb+1
        f2_stage<true, false, false>(g, P, B3TakeU{u[0], 0, 0, mb}, B3TakeU{u[0], 4, 0, mb}, no_dma, no_dma, F2Hst{true, gtr, hvoff, 0u});
#pragma unroll
        for (int kb = 0; kb < 15; ++kb)
            f2_stage<false, false, false>(g, P, B3TakeU{u[(kb + 1) >> 1], 8 * ((kb + 1) & 1), (kb + 1) >> 1, mb},
                                          B3TakeU{u[(kb + 1) >> 1], 8 * ((kb + 1) & 1) + 4, (kb + 1) >> 1, mb}, no_dma, no_dma,
                                          F2Hst{true, gtr, hvoff, 1024u * (unsigned)(kb + 1)},
                                          F2Hst{mid, gtr, hvoff, 1024u * (unsigned)kb + a.stash_mid});
        // next: the zero stage of the next block (or the padding)
        f2_stage<false, false, true>(g, P, F3None{}, F3None{}, no_dma, no_dma, F2Hst{false, nullptr, 0u, 0u},
                                     F2Hst{mid, gtr, hvoff, 1024u * 15u + a.stash_mid});
    }

    f2_report_amax<B2S_AMAX>(a.status, P.amax, lane);  // AMAX (the next step's scale is chosen from it), FLAG if out of range

    // ---- head: dL/d(head pre-activation) = (g + dy) * (x_0 > 0) -> gx[0] ---------------------------------------------------------
    {
        // x_0 = relu(head): its fp16 stage pieces (slot 0 of save_x); stage kb = 2T + r holds fragment registers c = 8r .. 8r+7
        // of tile T.  Only the sign matters; fp16 rounds x_0 < 3e-8 to zero (such a unit counts as inactive).
        const u32x4* r = reinterpret_cast<const u32x4*>(a.save_x) + lane_unit;
        float* o = a.gx + ray * R2L_W + 4 * h;  // row-major fp32: the head weight gradient reads rows
#pragma unroll
        for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const f16x8 xv = __builtin_bit_cast(f16x8, r[64 * (2 * T + rr)]);
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    f32x4 ov;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = 8 * rr + 4 * q2 + j;
                        ov[j] = (float)xv[4 * q2 + j] > 0.f ? (g[T][c] + dy[T][c]) * ginv : 0.f;
                    }
                    *reinterpret_cast<f32x4*>(o + 32 * T + 8 * (2 * rr + q2)) = ov;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
int r2l_bwd2_pack(const float* params, int n_block, float* wstream2, hipStream_t stream) {
    hipLaunchKernelGGL(r2l_pack_bwd2_kernel, dim3(2048), dim3(256), 0, stream, params,
                       reinterpret_cast<unsigned short*>(wstream2), n_block);
    R2L_CHECK(hipGetLastError());
    return 0;
}

int r2l_bwd2_backward(const float* rgb, const float* target, const float* drgb, const float* save_x, const float* save_t,
                      const float* wstream_bwd2, const float* params, int n_block, float grad_scale, float* dpre, float* gx,
                      float* gt, float* sqerr_partial, int64_t N, hipStream_t stream, float gscale, unsigned* status,
                      const float* scale_dev, int b_start, int b_end) {
    if (r2l_use_coopf(N, n_block))  // small launches: the cooperative chain (r2l_coopf_bwd.hip), same stream / stash / status word
        return r2l_coopf_backward(rgb, target, drgb, save_x, save_t, wstream_bwd2, params, n_block, grad_scale, dpre, gx, gt,
                                  sqerr_partial, N, stream, gscale, status, scale_dev, b_start, b_end);
    if (b_start >= 0 && !(b_start == n_block - 1 && b_end == 0)) {
        r2l_set_error_msg("r2l_bwd2_backward: only the cooperative chains can be cut into block ranges");
        return (int)hipErrorInvalidValue;
    }
    B2Args a{};
    a.status = status;
    a.scale_dev = scale_dev;
    a.fmt = reinterpret_cast<const unsigned*>(save_x) + R2L_STASH_FMT_WORD(n_block, R2L_PAD_ROWS(N));
    a.gscale = gscale; a.ginv = 1.0f / gscale;
    a.rgb = rgb; a.target = target; a.drgb = drgb; a.save_x = save_x; a.save_t = save_t;
    a.stream = reinterpret_cast<const unsigned char*>(wstream_bwd2); a.params = params; a.n_block = n_block;
    a.grad_scale = grad_scale; a.dpre = dpre; a.gx = gx; a.gt = gt; a.sqerr_partial = sqerr_partial; a.N = N;
    a.stash_mid = r2l_dw_exact() ? (unsigned)R2L_H16_MID_BYTES(R2L_PAD_ROWS(N)) : 0u;
    const int64_t tiles = (N + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS;
    if (a.stash_mid != 0u) hipLaunchKernelGGL(r2l_bwd2_kernel<true>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(r2l_bwd2_kernel<false>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, stream, a);
    R2L_CHECK(hipGetLastError());
    return 0;
}
