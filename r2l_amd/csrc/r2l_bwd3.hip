// r2l_bwd3.hip — the dX chain of the R2L student backward at fp32 accuracy on the bf16 matrix pipe (same scheme as
// r2l_fwd3.hip, machinery in r2l_f3.h): one wave = 32 rays, per block  u = (W2^T g) * [t_b > 0],  g += W1^T u,  writing
// g (-> gx[b+1]) and the masked u (-> gt[b]) for the weight-gradient GEMMs.
//   * stage stream: per block (last block first) [zero stage, 16 k-blocks of W2^T, zero stage, 16 k-blocks of W1^T].  The
//     all-zero "bias" stages cost 8 MFMAs each and buy what the bias stages give the forward: u is zero-initialised by the
//     matrix pipe, and the B values of a GEMM's first k-block are gathered AFTER the previous GEMM has finished;
//   * the B values of every stage are exactly what has to be stashed (g, masked u): the stores ride along the gathers;
//   * the ReLU mask of the block comes as 128 bits per lane that the forward chain wrote beside its stash (one 1 KiB piece
//     per tile and block): DMA'd into a two-slot per-wave LDS ring at the start of the block, read when the second GEMM
//     starts (by then the staging pipeline's own `vmcnt` waits have guaranteed its arrival) and applied when u is gathered as
//     the B operand of the second GEMM;
//   * dy (the outer-residual branch) waits in scratch for the head.
#include "r2l_f3.h"

#define B3_NBUF 6
#define B3_RING 2  // mask pieces (one per block) per wave

__host__ __device__ static inline int64_t b3_off_body_w(int layer) {
    return (int64_t)R2L_IN * R2L_W + R2L_W + (int64_t)layer * (R2L_W * R2L_W + R2L_W);
}
__host__ __device__ static inline int64_t b3_off_tail_w(int n_block) { return b3_off_body_w(2 * n_block); }

// =================================================================================================================
// pack: stage g = 34 * slot + r, slot = n_block-1-b;  r = 0 / 17: zero stages;  r = 1..16: k-block r-1 of W2^T (layer 2b+1);
// r = 18..33: k-block r-18 of W1^T (layer 2b).  Element (split, tile t, lane (i,h), slot s): (W^T)[32t+i][feature(kb,h,s)]
// =================================================================================================================
// When this pack runs as the FALLBACK of the fp16 dX chain (run_if = the step's status words, raised) and the forward of the step
// stayed on the fp16 kernels (format word 0: only the gradient chain left fp16's range), the stash still holds fp16 stage pieces
// of x / act_s and relu(t) / act_s (r2l_f2.h), which neither the bf16x3 chain behind this launch (x_0's ReLU mask) nor its
// weight-gradient kernel (both operands chunked fp32) can read (round 4 read them as fp32 — garbage body dW, applied
// silently).  The first 2 n_block workgroups therefore EXPAND one slot each to the chunked fp32 layout, in place and unscaled:
// per 32-ray tile 16 KiB of hi halves at byte 16384 T become 32 KiB at 32768 T, which covers the pieces of tiles 2T and 2T + 1 —
// so a slot's tiles are walked from the last to the first by ONE workgroup (a rare path: no parallelism spent on it).  The hi
// halves are all the default trio stashes (its weight gradients take one fp16 product anyway); the mid halves of the exact-dW
// mode lie where the expanded tiles go and are dropped: such a step's weight gradients are the default trio's.  The last
// workgroup to finish flips the format word to 1, so that a second backward over the same stash goes straight to the bf16x3 kernels.
__device__ __forceinline__ void b3_expand_h16_slot(float* __restrict__ base, int64_t n_tiles, float act_scale) {
    const int t = (int)threadIdx.x;
    typedef _Float16 b3_f16x8 __attribute__((ext_vector_type(8)));
    const b3_f16x8* in = reinterpret_cast<const b3_f16x8*>(base);
    for (int64_t T = n_tiles - 1; T >= 0; --T) {
        f32x4 o[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const b3_f16x8 v = in[T * 1024 + t + 256 * j];  // (R2L_H16_TILE_UNITS of r2l_f2.h: 1024 16-byte units per tile)
            o[j][0] = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]} * act_scale;
            o[j][1] = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]} * act_scale;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this lane's loads of the tile have landed ...
        __syncthreads();  // ... every lane's, before any lane overwrites them (T = 0: the output covers the tile's own pieces)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = t + 256 * j, kb = u >> 6, lane = u & 63, i = lane & 31, h = lane >> 5;
            float* dst = base + T * R2L_CHUNK_TILE + (int64_t)(2 * kb) * R2L_CHUNK_PIECE + i * 8 + 4 * h;
            *reinterpret_cast<f32x4*>(dst) = o[j][0];                    // features 16 kb + 4 h .. + 3      (slots s = 0 .. 3)
            *reinterpret_cast<f32x4*>(dst + R2L_CHUNK_PIECE) = o[j][1];  // features 16 kb + 8 + 4 h .. + 3  (slots s = 4 .. 7)
        }
    }
}
__global__ void r2l_pack_bwd3_kernel(const float* __restrict__ params, unsigned short* __restrict__ out, int n_block,
                                     unsigned* __restrict__ run_if, float* save_x, float* save_t, int64_t Np) {
    if (run_if != nullptr && __builtin_nontemporal_load(run_if) == 0u) return;  // fallback stream: only packed when needed
    if (run_if != nullptr && save_x != nullptr && (int)blockIdx.x < 2 * n_block) {
        unsigned* fmt = reinterpret_cast<unsigned*>(save_x) + R2L_STASH_FMT_WORD(n_block, Np);
        if (__builtin_nontemporal_load(fmt) == 0u) {  // (uniform: every workgroup reads it before the last one can flip it)
            const float act_scale = reinterpret_cast<const float*>(fmt)[1];
            const int b = (int)blockIdx.x >> 1;
            b3_expand_h16_slot(((blockIdx.x & 1) ? save_t : save_x) + (int64_t)b * R2L_TRIO_SLOT(Np), Np / R2L_TILE_RAYS, act_scale);
            __threadfence();
            __syncthreads();
            if (threadIdx.x == 0 && atomicAdd(run_if + B2S_EXPANDED, 1u) == 2u * (unsigned)n_block - 1u) {
                __threadfence();
                *fmt = 1u;
            }
        }
    }
    const int64_t stages = r2l_bwd3_stages(n_block);
    const int64_t total = (stages + R2L_F3_PAD_STAGES) * 8 * 64 * 8;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(idx & 7), lane = (int)((idx >> 3) & 63), tile = (int)((idx >> 9) & 7);
        const int64_t g = idx >> 12;
        const int i = lane & 31, h = lane >> 5, o = 32 * tile + i;
        unsigned short* st = out + g * (F3_STAGE_BYTES / 2);
        unsigned short v0 = 0, v1 = 0, v2 = 0;
        if (g < stages) {
            const int slot = (int)(g / 34), r = (int)(g % 34), b = n_block - 1 - slot;
            if (r != 0 && r != 17) {
                const int layer = r < 17 ? 2 * b + 1 : 2 * b, kb = r < 17 ? r - 1 : r - 18;
                const int T = kb >> 1, rr = kb & 1;
                const int in = 32 * T + 8 * (2 * rr + (s >> 2)) + 4 * h + (s & 3);
                const float w = params[b3_off_body_w(layer) + (int64_t)in * R2L_W + o];  // (W^T)[o][in] = W[in][o]
                v0 = f3_bf16_rne(w);
                const float r1 = w - f3_bf16_to_f(v0);
                v1 = f3_bf16_rne(r1);
                v2 = f3_bf16_rne(r1 - f3_bf16_to_f(v1));
            }
        }
        const int64_t e = ((int64_t)tile * 64 + lane) * 8 + s;
        st[e] = v0;
        st[8 * 64 * 8 + e] = v1;
        st[2 * 8 * 64 * 8 + e] = v2;
    }
}

// =================================================================================================================
// kernel
// =================================================================================================================
struct B3Args {
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
    const float* scale_dev;  // {gscale, 1 / gscale} chosen on the device (generic mode behind the fp16 chain), or nullptr
    const unsigned* run_if;  // nullptr, or: return at once while this word is 0 (range-guard fallback of r2l_bwd2.hip)
    float* dpre;
    float* gx;
    float* gt;
    float* sqerr_partial;
    int64_t N;
};

// gatherers: four B values of the next stage (+ their stash store)
struct B3TakeG {  // g values (identity), stored to gx[b+1]
    const f32x16& frag;
    int c0;
    float* stash;  // lane base (r2l_chunk_lane) in the gx slot: chunked layout
    int T;
    __device__ __forceinline__ void operator()(float (&v)[4]) const {
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = frag[c0 + s];
        r2l_chunk_store(stash + R2L_CHUNK_PIECE * (4 * T + (c0 >> 2)), f32x4{v[0], v[1], v[2], v[3]});
    }
};
struct B3TakeU {  // u values masked by relu'(t_b) (mask words of the forward: bit (T&1)*16 + c of word T>>1), stored to gt[b]
    const f32x16& frag;
    int c0;
    float* stash;
    int T;
    const u32x4& mb;
    __device__ __forceinline__ void operator()(float (&v)[4]) const {
        const unsigned w = mb[T >> 1] >> ((T & 1) * 16 + c0);
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = ((w >> s) & 1u) ? frag[c0 + s] : 0.f;
        r2l_chunk_store(stash + R2L_CHUNK_PIECE * (4 * T + (c0 >> 2)), f32x4{v[0], v[1], v[2], v[3]});
    }
};

__global__ __launch_bounds__(256, 1) void r2l_bwd3_kernel(const B3Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[B3_NBUF][F3_STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char mring[4][B3_RING][1024];
    if (a.run_if != nullptr && __builtin_nontemporal_load(a.run_if) == 0u) return;
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
        const float* tw = a.params + b3_off_tail_w(a.n_block);
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

    // ---- weight staging (6 buffers beside the 8 KiB mask ring) ----------------------------------------------------------------
    typedef F3PipeT<B3_NBUF> Pipe;
    Pipe P;
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

    // mask words of this wave's tile: one 1 KiB piece per block (64 lanes x 16 B), fetched by DMA into a two-slot LDS ring at
    // the start of the block and read when its second GEMM starts, 17 stages later
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)&mring[0][0][0] +
                              (unsigned)wave * (B3_RING * 1024u);
    const unsigned char* ring_lane = &mring[0][0][0] + wave * (B3_RING * 1024) + lane * 16;
    const int64_t lane_off = r2l_chunk_lane(tile, lane & 31, h);  // this lane's base in a (chunked) slot
    const int64_t slot = R2L_TRIO_SLOT(Np);
    const unsigned mvoff = (unsigned)((R2L_MASK_OFFSET(Np) + tile * 256 + lane * 4) * 4);  // byte offset of the lane's mask words

#pragma unroll 1
    for (int b = a.n_block - 1; b >= 0; --b) {
        // descriptor of save_t[b] (per block: 32-bit offsets inside the slot)
        const unsigned long long ta = (unsigned long long)(a.save_t + (int64_t)b * slot);
        const u32x4 trs = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ta),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ta >> 32)) & 0xffffu, 0xffffffffu,
                           0x00020000u};
        float* gxs = a.gx + (int64_t)(b + 1) * slot + lane_off;
        float* gts = a.gt + (int64_t)b * slot + lane_off;
        const F3Dma no_dma{false, u32x4{0u, 0u, 0u, 0u}, 0u, 0u, 0u};
        const F3Dma mask_dma{true, trs, mvoff, 0u, ring_lds + (unsigned)(b & 1) * 1024u};
        // GEMM A: u = W2^T g.  stage 0 (zero stage, zero-initialises u) gathers g block 0; stage 1+kb gathers g block kb+1
        f3_stage<true, true, false>(u, P, B3TakeG{g[0], 0, gxs, 0}, B3TakeG{g[0], 4, gxs, 0}, mask_dma, no_dma);
#pragma unroll
        for (int kb = 0; kb < 15; ++kb)
            f3_stage<false, false, false>(u, P, B3TakeG{g[(kb + 1) >> 1], 8 * ((kb + 1) & 1), gxs, (kb + 1) >> 1},
                                          B3TakeG{g[(kb + 1) >> 1], 8 * ((kb + 1) & 1) + 4, gxs, (kb + 1) >> 1});
        // (the mask piece was requested 16 stages ago: every stage wait since has retired all but the newest loads)
        const u32x4 mb = *reinterpret_cast<const u32x4*>(ring_lane + (b & 1) * 1024);
        f3_stage<false, false, true>(u, P, F3None{}, F3None{});
        // GEMM B: g += W1^T (u . mask).  stage 17 (zero stage) gathers masked-u block 0; stage 18+kb gathers block kb+1
        f3_stage<true, false, false>(g, P, B3TakeU{u[0], 0, gts, 0, mb}, B3TakeU{u[0], 4, gts, 0, mb});
#pragma unroll
        for (int kb = 0; kb < 15; ++kb)
            f3_stage<false, false, false>(g, P, B3TakeU{u[(kb + 1) >> 1], 8 * ((kb + 1) & 1), gts, (kb + 1) >> 1, mb},
                                          B3TakeU{u[(kb + 1) >> 1], 8 * ((kb + 1) & 1) + 4, gts, (kb + 1) >> 1, mb});
        f3_stage<false, false, true>(g, P, F3None{}, F3None{});  // next: the zero stage of the next block (or the padding)
    }

    // ---- head: dL/d(head pre-activation) = (g + dy) * (x_0 > 0) -> gx[0] ---------------------------------------------------------
    {
        const float* r = a.save_x + lane_off;  // x_0: chunked like every slot the forward chain stashes
        float* o = a.gx + ray * R2L_W + 4 * h;  // row-major: the head weight gradient reads rows
#pragma unroll
        for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(r + R2L_CHUNK_PIECE * (4 * T + q));
                f32x4 ov;
#pragma unroll
                for (int j = 0; j < 4; ++j) ov[j] = xv[j] > 0.f ? (g[T][4 * q + j] + dy[T][4 * q + j]) * ginv : 0.f;
                *reinterpret_cast<f32x4*>(o + 32 * T + 8 * q) = ov;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
int r2l_bwd3_pack(const float* params, int n_block, float* wstream3, hipStream_t stream, const unsigned* run_if,
                  const float* save_x, const float* save_t, int64_t N) {
    // (the status words and the stash are library-private contents of caller-owned buffers: written through, hence the casts)
    hipLaunchKernelGGL(r2l_pack_bwd3_kernel, dim3(2048), dim3(256), 0, stream, params,
                       reinterpret_cast<unsigned short*>(wstream3), n_block, const_cast<unsigned*>(run_if),
                       const_cast<float*>(save_x), const_cast<float*>(save_t), R2L_PAD_ROWS(N));
    R2L_CHECK(hipGetLastError());
    return 0;
}

int r2l_bwd3_backward(const float* rgb, const float* target, const float* drgb, const float* save_x, const float* save_t,
                      const float* wstream_bwd3, const float* params, int n_block, float grad_scale, float* dpre, float* gx,
                      float* gt, float* sqerr_partial, int64_t N, hipStream_t stream, float gscale, const unsigned* run_if,
                      const float* scale_dev) {
    B3Args a{};
    a.run_if = run_if;
    a.gscale = gscale; a.ginv = 1.0f / gscale;
    a.scale_dev = scale_dev;
    a.rgb = rgb; a.target = target; a.drgb = drgb; a.save_x = save_x; a.save_t = save_t;
    a.stream = reinterpret_cast<const unsigned char*>(wstream_bwd3); a.params = params; a.n_block = n_block;
    a.grad_scale = grad_scale; a.dpre = dpre; a.gx = gx; a.gt = gt; a.sqerr_partial = sqerr_partial; a.N = N;
    const int64_t tiles = (N + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS;
    hipLaunchKernelGGL(r2l_bwd3_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, stream, a);
    R2L_CHECK(hipGetLastError());
    return 0;
}
