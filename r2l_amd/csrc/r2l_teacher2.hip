// r2l_teacher2.hip — the NeRF teacher's point network of r2l_teacher3.hip on the fp16 matrix pipe with two-way operand splits
// (machinery: r2l_f2.h): three fp16 products per fp32 product, ~2^-21 relative.  Stage order, gatherers and epilogue are
// r2l_teacher3.hip's; stages are 16 KiB.  A range guard raises a status word behind the stream and the bf16x3 kernel launched
// behind this one redoes the launch — and the stream is re-packed for a power-of-two activation scale (r2l_f2.h "range
// control": layer 0, the two embedding column blocks and every bias divided by s, the alpha / rgb heads multiplied by s: exact),
// so the launches after it run here again: the teacher's weights do not change, a sticky guard would be for good.
#include "r2l_f2.h"

#define T2_W 256
#define T2_XYZ 63
#define T2_DIR 27
// 145 stages of the 256-wide layers + the views layer (128 outputs = four of a stage's eight tile slots): its bias stage and
// nine DOUBLE stages, each holding TWO k-blocks (tile slots 0-3: tiles 0-3 of k-block 2j, slots 4-7: tiles 0-3 of k-block
// 2j + 1; eight feature pairs, then the two direction blocks) — both halves of such a stage accumulate into tiles 0-3, the
// second with its own B operand.  (Round 2 ran the layer as 19 ordinary stages whose slots 4-7 held zeros: 5.5 % of the
// launch's matrix work.)
#define T2_STAGES 155

struct T2Off {
    int64_t w[8], b[8], views_w, views_b, feat_w, feat_b, alpha_w, alpha_b, rgb_w, rgb_b, total;
};
__host__ __device__ static inline T2Off t2_offsets() {  // state_dict order of NeRF(D=8, W=256, 63, 27, use_viewdirs)
    T2Off o;
    int64_t p = 0;
    for (int i = 0; i < 8; ++i) {
        const int fin = i == 0 ? T2_XYZ : (i == 5 ? T2_W + T2_XYZ : T2_W);
        o.w[i] = p; p += (int64_t)T2_W * fin;
        o.b[i] = p; p += T2_W;
    }
    o.views_w = p; p += (int64_t)128 * (T2_W + T2_DIR);
    o.views_b = p; p += 128;
    o.feat_w = p; p += (int64_t)T2_W * T2_W;
    o.feat_b = p; p += T2_W;
    o.alpha_w = p; p += T2_W;
    o.alpha_b = p; p += 1;
    o.rgb_w = p; p += 3 * 128;
    o.rgb_b = p; p += 3;
    o.total = p;
    return o;
}

// embedding column of value v of half h (or -1 = zero padding); `nfreq_half` frequencies per half (5 xyz / 2 direction)
__host__ __device__ static inline int t2_emb_col(int v, int h, int nfreq_half) {
    const int ntrig = 6 * nfreq_half;
    if (v < ntrig) {
        const int q = v >> 1, fl = q / 3, ax = q % 3;
        return 3 + (nfreq_half * h + fl) * 6 + ((v & 1) ? 3 + ax : ax);
    }
    if (v == ntrig) return h ? 2 : 0;
    if (v == ntrig + 1) return h ? -1 : 1;
    return -1;
}

// =================================================================================================================
// pack
// =================================================================================================================
// inv_s: 1 / activation scale (a power of two).  The hidden activations, the feature vector and the views layer's output are
// all divided by s when layer 0's OUTPUT is (the kernel multiplies its fp32 accumulators: layer 0's weights and bias are packed
// as they are — dividing weights of O(0.1) by s before their fp16 split costs them their mid halves), and the
// embedding column blocks of layer 5 and of the views layer and every other bias are: kinds 0 (layer > 0), 1 (layer 5), 4 below.
// (Those two column blocks share their accumulators with the products of the hidden activations, so they stay weight-scaled:
// their terms are O(|W| |pe|) / s against hidden terms of O(amax / s) >= 2^10 whenever s > 1 — what they lose below 2^-25
// absolute is below 2^-35 of the layer's output.)
__device__ __forceinline__ void t2_pack_elements(const float* __restrict__ params, unsigned short* __restrict__ out, float inv_s) {
    const T2Off off = t2_offsets();
    const int64_t total = (int64_t)(T2_STAGES + R2L_F3_PAD_STAGES) * 8 * 64 * 8;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(idx & 7), lane = (int)((idx >> 3) & 63), tile = (int)((idx >> 9) & 7);
        const int g = (int)(idx >> 12);
        const int i = lane & 31, h = lane >> 5, o = 32 * tile + i;
        unsigned short* st = out + (int64_t)g * (F2_STAGE_BYTES / 2);
        // decode the stage: kind 0 bias (offset boff), 1 xyz block (layer 0 or 5), 2 256->256 block, 3 views feature block,
        // 4 views direction block
        int kind = -1, layer = 0, kb = 0;
        int64_t boff = 0;
        bool views = false;
        if (g == 0) { kind = 0; boff = off.b[0]; }
        else if (g < 5) { kind = 1; layer = 0; kb = g - 1; }
        else if (g < 145) {
            int r = g - 5, k = 0;
            const int psz[4] = {34, 34, 38, 34};
            while (r >= psz[k]) { r -= psz[k]; ++k; }
            const int lt = 1 + 2 * k, lx = 2 + 2 * k;  // lx == 8: feature_linear
            const int tl = (k == 2) ? 21 : 17;        // stages of the t-layer
            if (r < tl) {
                layer = lt;
                if (r == 0) { kind = 0; boff = off.b[lt]; }
                else if (k == 2 && r <= 4) { kind = 1; kb = r - 1; }
                else { kind = 2; kb = r - 1 - (k == 2 ? 4 : 0); }
            } else {
                r -= tl;
                layer = lx;
                if (r == 0) { kind = 0; boff = lx == 8 ? off.feat_b : off.b[lx]; }
                else { kind = 2; kb = r - 1; }
            }
        } else if (g < T2_STAGES) {
            const int r = g - 145;
            views = true;
            if (r == 0) { kind = 0; boff = off.views_b; }
            else if (r <= 8) { kind = 3; kb = 2 * (r - 1) + (tile >> 2); }  // double stage: k-blocks 2j (slots 0-3), 2j+1 (4-7)
            else { kind = 4; kb = tile >> 2; }
        }
        const int ov = 32 * (tile & 3) + i;  // output row of a views-layer double stage
        float w = 0.f;
        bool have = false;
        if (kind == 0) {
            if (!views || tile < 4) { w = params[boff + o]; have = true; }
        } else if (kind == 1) {
            const int col = t2_emb_col(8 * kb + s, h, 5);
            if (col >= 0) { w = layer == 5 ? params[off.w[5] + (int64_t)o * (T2_W + T2_XYZ) + col] : params[off.w[0] + (int64_t)o * T2_XYZ + col]; have = true; }
        } else if (kind == 2) {
            const int T = kb >> 1, r = kb & 1;
            const int in = 32 * T + 8 * (2 * r + (s >> 2)) + 4 * h + (s & 3);
            if (layer == 8) w = params[off.feat_w + (int64_t)o * T2_W + in];
            else if (layer == 5) w = params[off.w[5] + (int64_t)o * (T2_W + T2_XYZ) + T2_XYZ + in];
            else w = params[off.w[layer] + (int64_t)o * T2_W + in];
            have = true;
        } else if (kind == 3) {
            const int T = kb >> 1, r = kb & 1;
            const int in = 32 * T + 8 * (2 * r + (s >> 2)) + 4 * h + (s & 3);
            w = params[off.views_w + (int64_t)ov * (T2_W + T2_DIR) + in];
            have = true;
        } else if (kind == 4) {
            const int col = t2_emb_col(8 * kb + s, h, 2);
            if (col >= 0) { w = params[off.views_w + (int64_t)ov * (T2_W + T2_DIR) + T2_W + col]; have = true; }
        }
        unsigned short v0 = 0, v1 = 0;
        if (have) {
            if ((kind == 0 && g != 0) || (kind == 1 && layer == 5) || kind == 4) w *= inv_s;
            const _Float16 hi = (_Float16)w;
            const _Float16 mid = (_Float16)(w - (float)hi);
            const unsigned short hb = __builtin_bit_cast(unsigned short, hi), mb = __builtin_bit_cast(unsigned short, mid);
            if (kind == 0) {
                v0 = (h == 0) ? (s == 0 ? hb : (s == 1 ? mb : (unsigned short)0)) : (unsigned short)0;
            } else {
                v0 = hb; v1 = mb;
            }
        }
        const int64_t e = ((int64_t)tile * 64 + lane) * 8 + s;
        st[e] = v0;
        st[8 * 64 * 8 + e] = v1;
    }
}
// pack for the scale the status words ask for, then (one thread) commit it: r2l_f2.h range control, as r2l_fwd2.hip
__global__ void r2l_pack_teacher2_kernel(const float* __restrict__ params, unsigned short* __restrict__ out,
                                         const unsigned* __restrict__ status) {
    t2_pack_elements(params, out, f2_next_scale(status).inv);
}
__global__ void r2l_teacher2_commit_kernel(unsigned* status) {
    if (threadIdx.x == 0) f2_commit_scale(status, f2_next_scale(status), false);
}
// Behind every r2l_teacher2_kernel launch.  FLAG == 0 (the launch stayed in range): GO = 0, done.  Else the stream is re-packed
// for the next scale and the last workgroup commits it: FLAG cleared, GO = 1 — the r2l_teacher3 launch behind (run_if = GO)
// redoes this launch on its own (unscaled, always packed) stream, the next launch runs on the fp16 kernel again.
__global__ void r2l_teacher2_rescale_kernel(const float* __restrict__ params, unsigned short* __restrict__ out, unsigned* status) {
    bool tripped;
    if (!f2_rescale_due(status, tripped)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) status[F2S_GO] = 0u;
        return;
    }
    const F2Next nx = f2_next_scale(status);
    t2_pack_elements(params, out, nx.inv);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(status + F2S_DONE, 1u) == gridDim.x - 1u) f2_commit_scale(status, nx, tripped);
}

// =================================================================================================================
// kernel
// =================================================================================================================
struct T2Args {
    const float* rays_o;
    const float* rays_d;
    const float* viewdirs;
    const float* z;
    const unsigned char* stream;
    unsigned* status;  // range-guard word behind the stream: != 0 -> this launch is left to the bf16x3 kernel
    const float* params;
    float* raw;
    int64_t n_pts;
    int S;
};

// four embedding values v0 .. v0+3 of this half-wave: (sin, cos) pairs of c[axis] * 2^(nf*h + fl), then the identity
template <int NF>
struct T2Emb4 {
    const float (&c)[3];
    int h;
    int v0;
    __device__ __forceinline__ void operator()(float (&out)[4]) const {
        const float base = h ? (float)(1 << NF) : 1.0f;
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            const int v = v0 + j;
            if (v < 6 * NF) {
                const int q = v >> 1, fl = q / 3, ax = q % 3;
#ifdef T2_TIME_NOSINCOS  // (timing experiment only: results are wrong)
                out[j] = c[ax] * (base * (float)(1 << fl)); out[j + 1] = out[j] + 1.0f;
#else
                r2l_sincos(c[ax] * (base * (float)(1 << fl)), out[j], out[j + 1]);
#endif
            } else if (v == 6 * NF) {
                out[j] = h ? c[2] : c[0];
                out[j + 1] = h ? 0.f : c[1];
            } else {
                out[j] = 0.f;
                out[j + 1] = 0.f;
            }
        }
    }
};
// relu(frag[c0 .. c0+3]) as B values (F3Take4<true>), and — riding along — four terms of a dot product with them: the alpha head
// (alpha_linear on relu(layer 7)) is accumulated by the 32 gathers of the feature layer's GEMM, in the shadow of its MFMAs, from
// the values they produce anyway; as a VALU burst after the layer it cost 1043 instructions per tile with the matrix pipe idle
// (3 v_max per ReLU, AGPR reads, serialized LDS waits: 3 - 4 % of the frame).  w: this lane's 4 weights (LDS; a zero page for
// the layer pairs that have no head).
struct T2ReluDot4 {
    const f32x16& frag;
    int c0;
    const float* w;
    float& acc;
    __device__ __forceinline__ void operator()(float (&v)[4]) const {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w);
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = fmaxf(frag[c0 + s], 0.f);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_fmaf(wv[s], v[s], acc);
    }
};
// either of two gatherers, chosen at run time (the stage in front of a conditionally inserted group of stages)
template <class GA, class GB>
struct T2Select {
    bool first;
    GA ga;
    GB gb;
    __device__ __forceinline__ void operator()(float (&v)[4]) const {
#ifdef T2_SELECT_BOTH  // (rounds 2 - 3: both gatherers evaluated, values selected — two sin / cos pairs per call for nothing in three
        float w[4];    //  of the four layer pairs, and in a BIAS stage, whose side work is not hidden under MFMAs)
        ga(v);
        gb(w);
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = first ? v[s] : w[s];
#else
        if (first) ga(v);  // (wave-uniform)
        else gb(v);
#endif
    }
};

// ---- double stages (views layer) ---------------------------------------------------------------------------------------
// F2Side (r2l_f2.h) with TWO gatherers: the four B values of the next stage's k-block A in steps 0-3 and of its k-block B in
// steps 2-5 (gather, hi, residual, mid each).
template <bool BIAS_A, class GA, class GB>
struct T2SideV {
    F2A4& a;
    const unsigned char* lb;
    int half;
    GA ga;
    GB gb;
    bool want_b;
    F3Dma dma;
    float& amax;
    float xa[4], xb[4];
    unsigned uha[2], uma[2], uhb[2], umb[2];
    typedef F2Side<false, F3None> S;
    __device__ __forceinline__ void loads(int i) {
#pragma unroll
        for (int k = 2 * i; k < 2 * i + 2; ++k) {
            if (k >= (BIAS_A ? 4 : 8)) continue;
            const int tt = BIAS_A ? k : k / 2, sp = BIAS_A ? 0 : k % 2;
            const f16x8 v = *reinterpret_cast<const f16x8*>(lb + (sp * 8 + 4 * half + tt) * 1024);
            if (sp == 0) a.h[tt] = v;
            else a.m[tt] = v;
        }
    }
    static __device__ __forceinline__ void track(float& amax, const float (&x)[4]) {
        amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(x[0])), __builtin_fabsf(x[1]));
        amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(x[2])), __builtin_fabsf(x[3]));
    }
    static __device__ __forceinline__ void residual(float (&x)[4], const unsigned (&u)[2]) {
        x[0] = f2_res_lo(u[0], x[0]); x[1] = f2_res_hi(u[0], x[1]);
        x[2] = f2_res_lo(u[1], x[2]); x[3] = f2_res_hi(u[1], x[3]);
    }
    __device__ __forceinline__ void step(int i) {
        loads(i);
        if (dma.on && i < 4) {
            if (i == 0) f2_dma_base(dma.la);
            f2_dma_piece_i(i, dma.rs, dma.voff, dma.so);
        }
        if (!want_b) return;
        if (i == 0) { ga(xa); track(amax, xa); }
        else if (i == 1) { uha[0] = S::pk(xa[0], xa[1]); uha[1] = S::pk(xa[2], xa[3]); }
        else if (i == 2) residual(xa, uha);
        else if (i == 3) { uma[0] = S::pk(xa[0], xa[1]); uma[1] = S::pk(xa[2], xa[3]); }
        if (i == 2) { gb(xb); track(amax, xb); }
        else if (i == 3) { uhb[0] = S::pk(xb[0], xb[1]); uhb[1] = S::pk(xb[2], xb[3]); }
        else if (i == 4) residual(xb, uhb);
        else if (i == 5) { umb[0] = S::pk(xb[0], xb[1]); umb[1] = S::pk(xb[2], xb[3]); }
    }
};
// One stage whose successor is a double stage (or the padding, BIAS_NEXT): acc (+)= stage k, and the B operands of BOTH
// k-blocks of stage k+1 are gathered (P.sb <- k-block A from galo / gahi, sb2 <- k-block B from gblo / gbhi).
// VCUR: stage k is itself a double stage — its second half multiplies tile slots 4-7 by sb2 into tiles 0-3.
template <bool BIAS_K, bool ZERO_K, bool VCUR, bool BIAS_NEXT, class GAlo, class GBlo, class GAhi, class GBhi>
__device__ __forceinline__ void t2_vstage(f32x16 (&acc)[R2L_NT], F2Pipe& P, F2Split& sb2, GAlo galo, GBlo gblo, GAhi gahi,
                                          GBhi gbhi) {
    static_assert(!(BIAS_K && VCUR), "a bias stage is an ordinary stage");
    T2SideV<BIAS_K, GAlo, GBlo> sa{P.a2, P.lb, 1, galo, gblo, !BIAS_NEXT, F3Dma{false, P.rs, 0u, 0u, 0u}, P.amax};
    f2_mfma_half<BIAS_K, ZERO_K>(acc, 0, P.a1, P.sb, sa);
    __builtin_amdgcn_sched_barrier(0);
    P.sync_next();
    T2SideV<BIAS_NEXT, GAhi, GBhi> sb{P.a1, P.lb, 0, gahi, gbhi, !BIAS_NEXT, P.request(), P.amax};
    f2_mfma_half<BIAS_K, ZERO_K>(acc, VCUR ? 0 : 1, P.a2, VCUR ? sb2 : P.sb, sb);
    __builtin_amdgcn_sched_barrier(0);
    if (BIAS_NEXT) {
        P.sb = P.ones;
    } else {
        P.sb.h = __builtin_bit_cast(f16x8, u32x4{sa.uha[0], sa.uha[1], sb.uha[0], sb.uha[1]});
        P.sb.m = __builtin_bit_cast(f16x8, u32x4{sa.uma[0], sa.uma[1], sb.uma[0], sb.uma[1]});
        sb2.h = __builtin_bit_cast(f16x8, u32x4{sa.uhb[0], sa.uhb[1], sb.uhb[0], sb.uhb[1]});
        sb2.m = __builtin_bit_cast(f16x8, u32x4{sa.umb[0], sa.umb[1], sb.umb[0], sb.umb[1]});
    }
}

__global__ __launch_bounds__(256, 1) void r2l_teacher2_kernel(const T2Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[F2_NBUF][F2_STAGE_BYTES];
    // weights of the two VALU heads (alpha_linear [256], rgb_linear [3][128]): staged in LDS once per workgroup.  Read from global
    // inside the tile — 80 16-byte loads per lane with no registers left to keep them in flight — the heads cost 7 % of the frame
    // (timing build without them: 117.8 -> 109.7 ms, profiles/r04_teacher_heads_ab.txt); as broadcast LDS reads they cost ~2 %.
    __shared__ __attribute__((aligned(16))) float head_a[2][T2_W];  // [0]: zeros, [1]: alpha_linear.weight
    __shared__ __attribute__((aligned(16))) float head_rgb[3 * 128];
    // the heads' four biases as well: a VMEM load at the end of a tile makes hipcc wait for vmcnt(0), i.e. for the twenty weight-DMA
    // loads the pipeline has in flight for the stages behind the tile (1 - 2 us per 60 us tile: most of what the heads "cost")
    __shared__ __attribute__((aligned(16))) float head_b[4];  // rgb_linear.bias[0..2], alpha_linear.bias
    if (__builtin_nontemporal_load(a.status) != 0u) return;
    const T2Off off = t2_offsets();
    for (int i = threadIdx.x; i < T2_W; i += 256) { head_a[0][i] = 0.f; head_a[1][i] = a.params[off.alpha_w + i]; }
    for (int i = threadIdx.x; i < 3 * 128; i += 256) head_rgb[i] = a.params[off.rgb_w + i];  // (published by the prologue's barrier)
    if (threadIdx.x < 4) head_b[threadIdx.x] = threadIdx.x < 3 ? a.params[off.rgb_b + threadIdx.x] : a.params[off.alpha_b];
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    const int64_t pt = tile * R2L_TILE_RAYS + (lane & 31);
    const bool valid = pt < a.n_pts;
    const int64_t pc = valid ? pt : a.n_pts - 1;
    const int64_t ray = pc / a.S;

    float p[3], vd[3];
    {
        const float z = a.z[pc];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            p[k] = a.rays_o[ray * 3 + k] + a.rays_d[ray * 3 + k] * z;  // create_data.py:484, mul/add rounded separately
            vd[k] = a.viewdirs[ray * 3 + k];
        }
    }
    const float* P0 = a.params;
    const float act_s = f2_act_scale(a.status);  // the activation scale the stream is packed for

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

    f32x16 x[R2L_NT], t[R2L_NT];
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

    typedef T2Emb4<5> Xyz4;
    typedef T2Emb4<2> Dir4;
    typedef F3Take4<true> Relu4;
    // ---- layer 0: x = W0 pe + b0 (pre-activation; every consumer applies the ReLU to its B values) ---------------------
    f2_stage<true, true, false>(x, P, Xyz4{p, h, 0}, Xyz4{p, h, 4});
    f2_stage<false, false, false>(x, P, Xyz4{p, h, 8}, Xyz4{p, h, 12});
    f2_stage<false, false, false>(x, P, Xyz4{p, h, 16}, Xyz4{p, h, 20});
    f2_stage<false, false, false>(x, P, Xyz4{p, h, 24}, Xyz4{p, h, 28});
    f2_stage<false, false, true>(x, P, F3None{}, F3None{});
    {   // layer 0 ran on unscaled weights: the chain continues on x / act_s (a power of two, 1.0 for every teacher in range)
        const float act_inv = f2_act_inv(a.status);
#pragma unroll
        for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
            for (int c = 0; c < 16; ++c) x[T][c] *= act_inv;
    }

    // ---- (L1,L2) (L3,L4) (L5,L6) (L7,feature): t = W_odd relu(x) [+ W5pe pe] + b ; x = W_even relu(t) + b -----------------
    float alpha = 0.f;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        {
            // opaque copies: keeps hipcc from hoisting the (loop-invariant) sin/cos values of the k == 2 branch out of the loop
            float pp[3] = {p[0], p[1], p[2]};
            int hh = h;
            asm volatile("" : "+v"(pp[0]), "+v"(pp[1]), "+v"(pp[2]), "+v"(hh));
            f2_stage<true, true, false>(t, P, T2Select<Xyz4, Relu4>{k == 2, Xyz4{pp, hh, 0}, Relu4{x[0], 0, nullptr, 0}},
                                        T2Select<Xyz4, Relu4>{k == 2, Xyz4{pp, hh, 4}, Relu4{x[0], 4, nullptr, 0}});
            if (k == 2) {
                f2_stage<false, false, false>(t, P, Xyz4{pp, hh, 8}, Xyz4{pp, hh, 12});
                f2_stage<false, false, false>(t, P, Xyz4{pp, hh, 16}, Xyz4{pp, hh, 20});
                f2_stage<false, false, false>(t, P, Xyz4{pp, hh, 24}, Xyz4{pp, hh, 28});
                f2_stage<false, false, false>(t, P, Relu4{x[0], 0, nullptr, 0}, Relu4{x[0], 4, nullptr, 0});
            }
        }
#pragma unroll
        for (int kb = 0; kb < 15; ++kb)
            f2_stage<false, false, false>(t, P, Relu4{x[(kb + 1) >> 1], 8 * ((kb + 1) & 1), nullptr, 0},
                                          Relu4{x[(kb + 1) >> 1], 8 * ((kb + 1) & 1) + 4, nullptr, 0});
        f2_stage<false, false, true>(t, P, F3None{}, F3None{});
        // x = W_even relu(t) + b; k == 3 (feature_linear): its gathers of relu(layer 7) also accumulate the alpha head
        const float* wa = &head_a[k == 3 ? 1 : 0][4 * h];
        f2_stage<true, true, false>(x, P, T2ReluDot4{t[0], 0, wa, alpha}, T2ReluDot4{t[0], 4, wa + 8, alpha});
#pragma unroll
        for (int kb = 0; kb < 15; ++kb)
            f2_stage<false, false, false>(x, P,
                                          T2ReluDot4{t[(kb + 1) >> 1], 8 * ((kb + 1) & 1), wa + 32 * ((kb + 1) >> 1) + 16 * ((kb + 1) & 1), alpha},
                                          T2ReluDot4{t[(kb + 1) >> 1], 8 * ((kb + 1) & 1) + 4, wa + 32 * ((kb + 1) >> 1) + 16 * ((kb + 1) & 1) + 8, alpha});
        f2_stage<false, false, true>(x, P, F3None{}, F3None{});  // next: the next pair's (or the views layer's) bias stage
    }

    // ---- views layer: v[128] = Wv [feature, dir-embedding] + bv in tiles 0-3 of t (ReLU applied by the rgb head) ---------
    // bias stage, then nine double stages (two k-blocks each: feature pairs (2j, 2j+1) = fragment registers 0-7 / 8-15 of
    // tile j, then the two direction blocks); every stage gathers both B operands of its successor
    typedef F3Take4<false> Id4;
    F2Split sb2 = P.ones;
    t2_vstage<true, true, false, false>(t, P, sb2, Id4{x[0], 0, nullptr, 0}, Id4{x[0], 8, nullptr, 0}, Id4{x[0], 4, nullptr, 0},
                                        Id4{x[0], 12, nullptr, 0});
#pragma unroll
    for (int j = 1; j < 8; ++j)
        t2_vstage<false, false, true, false>(t, P, sb2, Id4{x[j], 0, nullptr, 0}, Id4{x[j], 8, nullptr, 0},
                                             Id4{x[j], 4, nullptr, 0}, Id4{x[j], 12, nullptr, 0});
    t2_vstage<false, false, true, false>(t, P, sb2, Dir4{vd, h, 0}, Dir4{vd, h, 8}, Dir4{vd, h, 4}, Dir4{vd, h, 12});
    t2_vstage<false, false, true, true>(t, P, sb2, F3None{}, F3None{}, F3None{}, F3None{});  // next: stream padding

    // rgb = Wrgb relu(v) + b
    float acc3[3] = {0.f, 0.f, 0.f};
#ifdef T2_TIME_NOHEADS
    acc3[0] = t[0][0]; acc3[1] = t[1][1]; acc3[2] = t[2][2];
#else
    {
        // Nothing gathers v any more, so this head stays a VALU burst with the matrix pipe idle — written for its instruction
        // count: one v_max per ReLU (fmaxf costs canonicalise + max), the products as packed FMAs on operand PAIRS that already
        // sit in adjacent registers (two weights of a 16-byte LDS read x two neighbouring values: plain v_pk_fma_f32, no op_sel),
        // even / odd partial sums per channel.  hipcc's own vectorisation of the scalar loop: 972 instructions; this: ~330.
        typedef float t2_f32x2 __attribute__((ext_vector_type(2)));
        t2_f32x2 a2[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        const float* wr = head_rgb + 4 * h;
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 wv[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) wv[c] = *reinterpret_cast<const f32x4*>(wr + c * 128 + 32 * T + 8 * q);
                float y[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) asm("v_max_f32_e32 %0, 0, %1" : "=v"(y[j]) : "v"(t[T][4 * q + j]));
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    a2[c] = __builtin_elementwise_fma(t2_f32x2{wv[c][0], wv[c][1]}, t2_f32x2{y[0], y[1]}, a2[c]);
                    a2[c] = __builtin_elementwise_fma(t2_f32x2{wv[c][2], wv[c][3]}, t2_f32x2{y[2], y[3]}, a2[c]);
                }
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) acc3[c] = a2[c][0] + a2[c][1];
    }
#endif
#pragma unroll
    for (int c = 0; c < 3; ++c) acc3[c] = (acc3[c] + __shfl_xor(acc3[c], 32)) * act_s + head_b[c];
    f2_report_amax(a.status, P.amax, lane);  // AMAX, and FLAG if this launch belongs to the bf16x3 kernel
    alpha = (alpha + __shfl_xor(alpha, 32)) * act_s + head_b[3];  // (the chain holds activations / act_s: exact)
    if (valid && h == 0) {
        const f32x4 o4 = {acc3[0], acc3[1], acc3[2], alpha};
        *reinterpret_cast<f32x4*>(a.raw + pt * 4) = o4;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side (called from r2l_teacher_mlp.hip's C ABI entry points)
// ------------------------------------------------------------------------------------------------------------------
static inline int64_t t2_status_offset() { return (int64_t)(T2_STAGES + R2L_F3_PAD_STAGES) * (F2_STAGE_BYTES / 4); }
int64_t r2l_teacher2_stream_floats(void) { return t2_status_offset() + 16; }
const unsigned* r2l_teacher2_status(const float* wstream2) { return reinterpret_cast<const unsigned*>(wstream2 + t2_status_offset()); }

int r2l_teacher2_pack(const float* tparams, float* wstream2, hipStream_t stream) {
    unsigned* status = reinterpret_cast<unsigned*>(wstream2 + t2_status_offset());
    hipLaunchKernelGGL(r2l_pack_teacher2_kernel, dim3(512), dim3(256), 0, stream, tparams,
                       reinterpret_cast<unsigned short*>(wstream2), status);
    R2L_CHECK(hipGetLastError());
    hipLaunchKernelGGL(r2l_teacher2_commit_kernel, dim3(1), dim3(64), 0, stream, status);
    R2L_CHECK(hipGetLastError());
    return 0;
}

int r2l_teacher2_mlp(const float* rays_o, const float* rays_d, const float* viewdirs, const float* z,
                     const float* wstream2, const float* tparams, float* raw, int64_t n_pts, int S, hipStream_t stream) {
    T2Args a{};
    a.rays_o = rays_o; a.rays_d = rays_d; a.viewdirs = viewdirs; a.z = z;
    a.stream = reinterpret_cast<const unsigned char*>(wstream2); a.params = tparams; a.raw = raw; a.n_pts = n_pts; a.S = S;
    // the status word lives in the caller's stream buffer (library-private contents): written through, hence the cast
    a.status = reinterpret_cast<unsigned*>(const_cast<float*>(wstream2) + t2_status_offset());
    const int64_t tiles = (n_pts + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS;
    hipLaunchKernelGGL(r2l_teacher2_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, stream, a);
    R2L_CHECK(hipGetLastError());
    hipLaunchKernelGGL(r2l_teacher2_rescale_kernel, dim3(512), dim3(256), 0, stream, tparams,
                       reinterpret_cast<unsigned short*>(const_cast<float*>(wstream2)), a.status);
    R2L_CHECK(hipGetLastError());
    return 0;
}
