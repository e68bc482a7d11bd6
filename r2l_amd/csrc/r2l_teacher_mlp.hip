// r2l_teacher_mlp.hip — fused NeRF-teacher point network for gfx950, replacing run_network + NeRF.forward
// (/root/reference/utils/create_data.py:55-77, model/nerf_raybased.py:377-401; embedder helpers:24-74):
//   pts = o + d*z  ->  embed xyz (63) , embed viewdir (27)  ->  8 x Linear(.,256)+ReLU with the input re-concatenated
//   after layer 4 (K = 319)  ->  alpha = Linear(256,1), feature = Linear(256,256)
//   ->  Linear(283,128)+ReLU on [feature, dir-embedding]  ->  rgb = Linear(128,3)   ->  raw[point] = (r,g,b,sigma)
// Same register-resident chain as the student (r2l_common.h): one wavefront = 32 consecutive sample points, exact-fp32
// MFMA, weights pre-packed into ONE stream in consumption order (r2l_pack_teacher below).
#include "r2l_common.h"

#define T_W 256
#define T_XYZ 63
#define T_DIR 27
#define T_PE_STEPS 36   // xyz embedding k-steps per half-wave (30 trig + 3 identity + 3 pad)  -> 9 groups
#define T_PE_GROUPS 9
#define T_DIR_STEPS 16  // dir embedding k-steps per half-wave (12 trig + 3 identity + 1 pad)  -> 4 groups of 4 tiles

// ---- flat parameter offsets, state_dict order of NeRF(D=8,W=256,63,27,use_viewdirs=True) ------------------------------
struct TOff {
    int64_t w[8], b[8], views_w, views_b, feat_w, feat_b, alpha_w, alpha_b, rgb_w, rgb_b, total;
};
__host__ __device__ static inline TOff t_offsets() {
    TOff o;
    int64_t p = 0;
    for (int i = 0; i < 8; ++i) {
        const int fin = i == 0 ? T_XYZ : (i == 5 ? T_W + T_XYZ : T_W);
        o.w[i] = p; p += (int64_t)T_W * fin;
        o.b[i] = p; p += T_W;
    }
    o.views_w = p; p += (int64_t)128 * (T_W + T_DIR);
    o.views_b = p; p += 128;
    o.feat_w = p; p += (int64_t)T_W * T_W;
    o.feat_b = p; p += T_W;
    o.alpha_w = p; p += T_W;
    o.alpha_b = p; p += 1;
    o.rgb_w = p; p += 3 * 128;
    o.rgb_b = p; p += 3;
    o.total = p;
    return o;
}

// xyz-embedding column fed by k-step s of half-wave h (or -1 = zero padding).  The k order is ours to choose (the pack
// kernel follows it): (sin, cos) PAIRS, so that a group of 4 k-steps needs exactly two sincos evaluations and the
// kernel can generate them group by group instead of holding all 36 values (which spilled to scratch):
//   s < 30 : pair q = s/2 -> frequency 5h + q/3, axis q%3; even s = sin, odd s = cos;  s = 30..32 : identity (half 0 only)
__host__ __device__ static inline int t_xyz_col(int s, int h) {
    if (s < 30) {
        const int q = s >> 1, fl = q / 3, ax = q % 3;
        return 3 + (5 * h + fl) * 6 + ((s & 1) ? 3 + ax : ax);
    }
    if (s < 33) return h == 0 ? s - 30 : -1;
    return -1;
}
__host__ __device__ static inline int t_dir_col(int s, int h) {
    if (s < 12) return 3 + (2 * h + s / 6) * 6 + (s % 6);
    if (s < 15) return h == 0 ? s - 12 : -1;
    return -1;
}

// stream layout (units: load-groups of 8 float4 per lane = 2048 floats); "B*" = bias group (r2l_common.h)
//   [B0][L0 pe x9] {[B_i][L_i x32]} i=1..4  [B5][L5 pe x9 + zero group][L5 h x32] [B6][L6 x32] [B7][L7 x32] [BF][feature x32]
//   [BV][views(feature part) x16][views(dir part) x2]
#define TG_B0 0
#define TG_L0 1
#define TG_BODY (TG_L0 + T_PE_GROUPS)                 // 10: layers 1..4, 33 groups each (bias + 32)
#define TG_B5 (TG_BODY + 4 * 33)                      // 142
#define TG_L5PE (TG_B5 + 1)
#define T_PE5_GROUPS 10                               // layer 5's embedding part: 9 groups + 1 all-zero group, so that every
                                                      // trip of the layer-pair loop has an even number of groups (ring of 2)
#define TG_L5H (TG_L5PE + T_PE5_GROUPS)               // 153
#define TG_L67F (TG_L5H + 32)                         // 185: layers 6, 7, feature: 33 groups each
#define TG_BV (TG_L67F + 3 * 33)                      // 284
#define TG_VIEWS (TG_BV + 1)                          // 32 k-groups x 4 tiles = 16 load-groups
#define TG_VDIR (TG_VIEWS + 16)                       // 4 k-groups x 4 tiles = 2 load-groups
#define TG_TOTAL (TG_VDIR + 2)                        // 303

__global__ void r2l_pack_teacher_kernel(const float* __restrict__ params, float* __restrict__ out) {
    const TOff off = t_offsets();
    const int64_t total = (int64_t)TG_TOTAL * R2L_GROUP_FLOATS;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total + R2L_STREAM_PAD;
         i += (int64_t)gridDim.x * blockDim.x) {
        if (i >= total) { out[i] = 0.f; continue; }
        const int gidx = (int)(i / R2L_GROUP_FLOATS);
        const int rem = (int)(i % R2L_GROUP_FLOATS);
        const int slot = rem >> 8, lane = (rem & 255) >> 2, j = rem & 3;
        const int h = lane >> 5, jl = lane & 31;
        const int o = 32 * slot + jl;
        // decode the group: kind 0 bias (bias offset in `boff`), 1 xyz-embedding part (layer 0 or 5), 2 256->256 layer,
        // 3 views feature part, 4 views direction part
        int kind, layer = 0, G = 0;
        int64_t boff = 0;
        if (gidx == TG_B0) { kind = 0; boff = off.b[0]; }
        else if (gidx < TG_BODY) { kind = 1; layer = 0; G = gidx - TG_L0; }
        else if (gidx < TG_B5) {
            layer = 1 + (gidx - TG_BODY) / 33; G = (gidx - TG_BODY) % 33 - 1;
            kind = G < 0 ? 0 : 2; boff = off.b[layer];
        }
        else if (gidx == TG_B5) { kind = 0; boff = off.b[5]; }
        else if (gidx < TG_L5H) { kind = 1; layer = 5; G = gidx - TG_L5PE; }
        else if (gidx < TG_L67F) { kind = 2; layer = 5; G = gidx - TG_L5H; }
        else if (gidx < TG_BV) {
            layer = 6 + (gidx - TG_L67F) / 33; G = (gidx - TG_L67F) % 33 - 1;  // 6, 7, 8 (= feature_linear)
            kind = G < 0 ? 0 : 2; boff = layer == 8 ? off.feat_b : off.b[layer];
        }
        else if (gidx == TG_BV) { kind = 0; boff = off.views_b; }
        else if (gidx < TG_VDIR) { kind = 3; G = gidx - TG_VIEWS; }
        else { kind = 4; G = gidx - TG_VDIR; }
        float v = 0.f;
        if (kind == 0) {
            const bool in_range = gidx == TG_BV ? slot < 4 : true;  // the views layer has 4 output tiles
            if (j == 0 && h == 0 && in_range) v = params[boff + o];
        } else if (kind == 1) {
            const int col = t_xyz_col(4 * G + j, h);
            if (col >= 0) v = layer == 5 ? params[off.w[5] + (int64_t)o * (T_W + T_XYZ) + col]
                                         : params[off.w[0] + (int64_t)o * T_XYZ + col];
        } else if (kind == 2) {
            const int in = 32 * (G >> 2) + 8 * (G & 3) + 4 * h + j;
            if (layer == 8) v = params[off.feat_w + (int64_t)o * T_W + in];
            else if (layer == 5) v = params[off.w[5] + (int64_t)o * (T_W + T_XYZ) + T_XYZ + in];
            else v = params[off.w[layer] + (int64_t)o * T_W + in];
        } else if (kind == 3) {  // slot = (k-group parity)*4 + tile
            const int Gk = 2 * G + (slot >> 2), tile = slot & 3;
            const int in = 32 * (Gk >> 2) + 8 * (Gk & 3) + 4 * h + j;
            v = params[off.views_w + (int64_t)(32 * tile + jl) * (T_W + T_DIR) + in];
        } else {
            const int g = 2 * G + (slot >> 2), tile = slot & 3;
            const int col = t_dir_col(4 * g + j, h);
            if (col >= 0) v = params[off.views_w + (int64_t)(32 * tile + jl) * (T_W + T_DIR) + T_W + col];
        }
        out[i] = v;
    }
}

struct TeacherArgs {
    const float* rays_o;    // [R,3]
    const float* rays_d;    // [R,3]
    const float* viewdirs;  // [R,3] unit view directions (render(): d / |d|, create_data.py:154-157)
    const float* z;         // [R,S]
    const float* wstream;
    const float* params;
    float* raw;             // [R,S,4]
    int64_t n_pts;          // R*S
    int S;
};

#ifndef T_RING
#define T_RING 1  // depth of the weight ring (1 or 2; same-box A/B: 2 is 0.6 % slower here); slot of group i = i & (T_RING - 1)
#endif
typedef WRingT<T_RING> TRing;
#define T_SLOT(i) ((i) & (T_RING - 1))

// the 4 B-operand values of embedding group g (k-steps 4g .. 4g+3) for this lane's half-wave: two sincos per group
template <int g>
__device__ __forceinline__ f32x4 t_xyz_group(const float (&p)[3], int h) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const float base = h ? 32.0f : 1.0f;
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
        constexpr int s0 = 4 * g;
        const int s = s0 + j;
        if (s < 30) {
            const int q = s >> 1, fl = q / 3, ax = q % 3;
            float sn, cs;
            r2l_sincos(p[ax] * (base * (float)(1 << fl)), sn, cs);
            v[j] = sn;
            v[j + 1] = cs;
        } else if (s < 33) {
            v[j] = h ? 0.f : p[s - 30];
            if (s + 1 < 33) v[j + 1] = h ? 0.f : p[s + 1 - 30];
        }
    }
    return v;
}

// NG groups of embedding k-steps, the first one in ring slot T_SLOT(BASE).  Software pipeline: the values of group g+1
// are evaluated (VALU) under the MFMAs of group g.
template <int BASE, int g, int NG>
__device__ __forceinline__ void t_pe_steps(f32x16 (&acc)[R2L_NT], const float (&p)[3], int h, TRing& ws, f32x4 cur) {
    if constexpr (g < NG) {
        f32x4 nxt = {0.f, 0.f, 0.f, 0.f};
        if constexpr (g + 1 < NG && g + 1 < T_PE_GROUPS) nxt = t_xyz_group<g + 1>(p, h);
        mfma_group<T_SLOT(BASE + g), 0, 0, 12>(acc, ws, cur[0], cur[1], cur[2], cur[3]);
        t_pe_steps<BASE, g + 1, NG>(acc, p, h, ws, nxt);
    }
}
template <int BASE, int NG>
__device__ __forceinline__ void t_pe_gemm(f32x16 (&acc)[R2L_NT], const float (&p)[3], int h, TRing& ws) {
    t_pe_steps<BASE, 0, NG>(acc, p, h, ws, t_xyz_group<0>(p, h));
}

__global__ __launch_bounds__(256, 1) void r2l_teacher_mlp_kernel(const TeacherArgs a) {
    const TOff off = t_offsets();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile * R2L_TILE_RAYS >= a.n_pts) return;
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
    TRing ws;
    ws.init(a.wstream, lane);
    const float* P = a.params;

    f32x16 x[R2L_NT], t[R2L_NT];
    const float one_h0 = h ? 0.f : 1.f;  // B operand of the bias k-steps
    // layer 0 (x holds PRE-activations from here on: every consumer applies the ReLU to its B operands on the fly)
    {
        mfma_bias_group<true, 0>(x, ws, one_h0);
        t_pe_gemm<1, T_PE_GROUPS>(x, p, h, ws);  // 1 + 9 groups: the loop below starts at an even group index
    }
    // (L1,L2) (L3,L4) (L5,L6) (L7,feature): t = W_odd relu(x) [+ W5pe pe] + b ; x = W_even relu(t) + b
    float alpha = 0.f;
    NoHook nh;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        mfma_bias_group<true, 0>(t, ws, one_h0);
        if (k == 2) {  // 10 groups (the last one all zero): slot parity unchanged
            // opaque copy: hipcc would otherwise hoist the (loop-invariant) sin/cos values out of the layer loop and
            // keep 33 of them in scratch across it; their reloads then queue behind the weight prefetches (in-order vmcnt)
            float pp[3] = {p[0], p[1], p[2]};
            int hh = h;  // (the per-half frequency scales too)
            asm volatile("" : "+v"(pp[0]), "+v"(pp[1]), "+v"(pp[2]), "+v"(hh));
            t_pe_gemm<1, T_PE5_GROUPS>(t, pp, hh, ws);
        }
        gemm256x<true, T_SLOT(1)>(t, x, ws, nh);
        if (k == 3) {  // alpha_linear on relu(layer 7)
            float acc = 0.f;
#pragma unroll
            for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(P + off.alpha_w + 32 * T + 8 * q + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = __builtin_fmaf(wv[j], fmaxf(t[T][4 * q + j], 0.f), acc);
                }
            acc += __shfl_xor(acc, 32);
            alpha = acc + P[off.alpha_b];
        }
        mfma_bias_group<true, T_SLOT(1)>(x, ws, one_h0);
        gemm256x<true, 0>(x, t, ws, nh);  // k == 3: x = feature_linear(relu(layer 7)), consumed WITHOUT a ReLU below
    }
    // views layer: v[128] = Wv [feature, dir-embedding] + bv   (4 output tiles; ReLU applied by the rgb head)
    f32x16 v[4];
    {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        ws.p.opaque();
#pragma unroll
        for (int t4 = 0; t4 < R2L_NT; ++t4) {
            if (t4 < 4) v[t4] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws.w[0][t4][0], one_h0, zero, 0, 0, 0);
            ws.w[0][t4] = ws.p[t4 * 64];
        }
        ws.p += R2L_NT * 64;
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int G2 = 0; G2 < 16; ++G2) {
        const int Ga = 2 * G2, Gb = 2 * G2 + 1;
        const float ba[4] = {x[Ga >> 2][(Ga & 3) * 4 + 0], x[Ga >> 2][(Ga & 3) * 4 + 1], x[Ga >> 2][(Ga & 3) * 4 + 2],
                             x[Ga >> 2][(Ga & 3) * 4 + 3]};
        const float bb[4] = {x[Gb >> 2][(Gb & 3) * 4 + 0], x[Gb >> 2][(Gb & 3) * 4 + 1], x[Gb >> 2][(Gb & 3) * 4 + 2],
                             x[Gb >> 2][(Gb & 3) * 4 + 3]};
        if (G2 & 1) mfma_group4x2<0>(v, ws, ba, bb);  // the views bias group sat in slot 0
        else mfma_group4x2<T_SLOT(1)>(v, ws, ba, bb);
    }
    {
        float fd[T_DIR_STEPS];
        const float base = h ? 4.0f : 1.0f;
#pragma unroll
        for (int fl = 0; fl < 2; ++fl)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax)
                r2l_sincos(vd[ax] * (base * (float)(1 << fl)), fd[fl * 6 + ax], fd[fl * 6 + 3 + ax]);
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) fd[12 + ax] = h ? 0.f : vd[ax];
        fd[15] = 0.f;
#pragma unroll
        for (int G2 = 0; G2 < 2; ++G2) {
            const float ba[4] = {fd[8 * G2 + 0], fd[8 * G2 + 1], fd[8 * G2 + 2], fd[8 * G2 + 3]};
            const float bb[4] = {fd[8 * G2 + 4], fd[8 * G2 + 5], fd[8 * G2 + 6], fd[8 * G2 + 7]};
            if (G2 & 1) mfma_group4x2<0>(v, ws, ba, bb);
            else mfma_group4x2<T_SLOT(1)>(v, ws, ba, bb);
        }
    }
    // rgb = Wrgb relu(v) + b
    float acc3[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 wv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) wv[c] = *reinterpret_cast<const f32x4*>(P + off.rgb_w + c * 128 + 32 * T + 8 * q + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = fmaxf(v[T][4 * q + j], 0.f);
#pragma unroll
                for (int c = 0; c < 3; ++c) acc3[c] = __builtin_fmaf(wv[c][j], y, acc3[c]);
            }
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) acc3[c] = acc3[c] + __shfl_xor(acc3[c], 32) + P[off.rgb_b + c];
    if (valid && h == 0) {
        const f32x4 o4 = {acc3[0], acc3[1], acc3[2], alpha};
        *reinterpret_cast<f32x4*>(a.raw + pt * 4) = o4;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------------
extern "C" int64_t r2l_teacher_param_count(void) { return t_offsets().total; }
// r2l_teacher3.hip: the same network on the bf16 matrix pipe (fp32-accurate); its stage stream follows the fp32 one
int64_t r2l_teacher3_stream_floats(void);
int r2l_teacher3_pack(const float* tparams, float* wstream3, hipStream_t stream);
int r2l_teacher3_mlp(const float* rays_o, const float* rays_d, const float* viewdirs, const float* z,
                     const float* wstream3, const float* tparams, float* raw, int64_t n_pts, int S, hipStream_t stream,
                     const unsigned* run_if);
// r2l_teacher2.hip: three fp16 products per fp32 product (default), range-guarded; its stream follows the bf16x3 one
int64_t r2l_teacher2_stream_floats(void);
const unsigned* r2l_teacher2_status(const float* wstream2);
int r2l_teacher2_pack(const float* tparams, float* wstream2, hipStream_t stream);
int r2l_teacher2_mlp(const float* rays_o, const float* rays_d, const float* viewdirs, const float* z,
                     const float* wstream2, const float* tparams, float* raw, int64_t n_pts, int S, hipStream_t stream);
static inline int64_t t_stream32_floats() { return (int64_t)TG_TOTAL * R2L_GROUP_FLOATS + R2L_STREAM_PAD; }

extern "C" int64_t r2l_teacher_stream_floats(void) {
    return t_stream32_floats() + r2l_teacher3_stream_floats() + r2l_teacher2_stream_floats();
}

// the 16 status words of the fp16x2 teacher stream inside `wstream` (include/r2l_hip.h: range control, telemetry)
extern "C" const unsigned* r2l_teacher_status_words(const float* wstream) {
    if (wstream == nullptr) return nullptr;
    return r2l_teacher2_status(wstream + t_stream32_floats() + r2l_teacher3_stream_floats());
}

extern "C" int r2l_pack_teacher(const float* params, float* wstream, void* stream) {
    R2L_REQUIRE(params && wstream, "r2l_pack_teacher: tparams / wstream is NULL");
    hipLaunchKernelGGL(r2l_pack_teacher_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, params, wstream);
    R2L_CHECK(hipGetLastError());
    const int rc = r2l_teacher3_pack(params, wstream + t_stream32_floats(), (hipStream_t)stream);
    if (rc) return rc;
    return r2l_teacher2_pack(params, wstream + t_stream32_floats() + r2l_teacher3_stream_floats(), (hipStream_t)stream);
}

extern "C" int r2l_teacher_mlp(const float* rays_o, const float* rays_d, const float* viewdirs, const float* z,
                               const float* wstream, const float* params, float* raw, int64_t R, int S, void* stream) {
    return r2l_teacher_mlp_cfg(rays_o, rays_d, viewdirs, z, wstream, params, raw, R, S, stream, nullptr);
}
extern "C" int r2l_teacher_mlp_cfg(const float* rays_o, const float* rays_d, const float* viewdirs, const float* z,
                                   const float* wstream, const float* params, float* raw, int64_t R, int S, void* stream,
                                   const r2l_config* cfg) {
    R2L_CFG_ENTER(cfg);
    R2L_REQUIRE(R >= 0 && S >= 0, "r2l_teacher_mlp: negative R / S");
    if (R == 0 || S == 0) return 0;
    R2L_REQUIRE(rays_o && rays_d && viewdirs && z && wstream && params && raw, "r2l_teacher_mlp: a required pointer is NULL");
    TeacherArgs a{};
    a.rays_o = rays_o; a.rays_d = rays_d; a.viewdirs = viewdirs; a.z = z; a.wstream = wstream; a.params = params;
    a.raw = raw; a.n_pts = R * (int64_t)S; a.S = S;
    if (a.n_pts <= 0) return 0;
    if (r2l_use_fwd2()) {  // default: 3 fp16 products per fp32 product, the bf16x3 kernel behind it as range-guard fallback
        const float* w3 = wstream + t_stream32_floats();
        const float* w2 = w3 + r2l_teacher3_stream_floats();
        const int rc = r2l_teacher2_mlp(rays_o, rays_d, viewdirs, z, w2, params, raw, a.n_pts, S, (hipStream_t)stream);
        if (rc) return rc;
        return r2l_teacher3_mlp(rays_o, rays_d, viewdirs, z, w3, params, raw, a.n_pts, S, (hipStream_t)stream,
                                r2l_teacher2_status(w2) + F2S_GO);
    }
    if (r2l_use_fwd3())  // R2L_NO_FWD2=1: fp32-exact products on the bf16 matrix pipe (R2L_NO_FWD3=1: fp32 MFMA)
        return r2l_teacher3_mlp(rays_o, rays_d, viewdirs, z, wstream + t_stream32_floats(), params, raw, a.n_pts, S,
                                (hipStream_t)stream, nullptr);
    const int64_t tiles = (a.n_pts + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS;
    hipLaunchKernelGGL(r2l_teacher_mlp_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    R2L_CHECK(hipGetLastError());
    return 0;
}
