// r2l_dw.h — what the weight-gradient kernels of the body layers share (r2l_backward.hip: fp32-MFMA and bf16x3 kernels, the
// slab reduce; r2l_dw16.hip: the fp16 kernel of the default training trio): flat-parameter offsets, the (layer, ray-chunk)
// work list and the per-workgroup partial-sum slab.
#pragma once
#include "r2l_common.h"

// ---- flat parameter offsets (same as r2l_forward.hip) ---------------------------------------------------------
__host__ __device__ static inline int64_t b_off_head_b() { return (int64_t)R2L_IN * R2L_W; }
__host__ __device__ static inline int64_t b_off_body_w(int layer) {
    return (int64_t)R2L_IN * R2L_W + R2L_W + (int64_t)layer * (R2L_W * R2L_W + R2L_W);
}
__host__ __device__ static inline int64_t b_off_body_b(int layer) { return b_off_body_w(layer) + R2L_W * R2L_W; }
__host__ __device__ static inline int64_t b_off_tail_w(int n_block) { return b_off_body_w(2 * n_block); }
__host__ __device__ static inline int64_t b_off_tail_b(int n_block) { return b_off_tail_w(n_block) + 3 * R2L_W; }

#define DW_CHUNK 64  // rays per work unit
#ifndef DW_DEPTH
#define DW_DEPTH 4  // rotating operand buffers = k-steps of load latency covered (must divide 32)
#endif
#ifndef DW_LONG_TRIP
#define DW_LONG_TRIP 128  // k-steps per trip of the main loop
#endif

struct R2LDwArgs {
    const float* save_x;
    const float* save_t;
    const float* gx;
    const float* gt;
    float* grads;  // flat gradient buffer (state_dict order)
    int n_block;
    int layer0;    // this launch covers the body layers [layer0, layer0 + n_layers) of the 2*n_block (gradient buckets, in
    int n_layers;  // backward order, for the overlapped all-reduce: r2l_backward_part)
    int64_t N;
    int64_t units_per_layer;  // ceil(N / DW_CHUNK)
    int64_t units_per_wg;
    float* slab;  // [wgs][2][DW_SLAB_FLOATS] per-workgroup partial (dW, db) of the <= 2 layers its range touches, or
                  // nullptr -> fp32 atomics straight into grads
    float unscale = 1.0f;  // the gradient operands carry a power-of-two scale (r2l_bwd3): dW, db are multiplied by its inverse
    const float* scale_dev = nullptr;  // generic mode of the fp16 trio: {gscale, 1 / gscale} chosen on the device; overrides unscale
    // fallback chaining of the default trio: r2l_dw16_kernel raises *status (if given) when it hands the launch over; a kernel
    // launched with run_if returns at once while that word is 0
    unsigned* status = nullptr;
    const unsigned* run_if = nullptr;
    // exact weight gradients of the fp16 trio (r2l_dw16.hip): != 0 -> the slots also hold the operands' mid halves, this many
    // bytes behind the hi stage pieces (r2l_f2.h R2L_H16_MID_BYTES), and every fp32 product is taken as hi*hi + hi*mid + mid*hi
    unsigned mid_off = 0u;
    // fp16 trio: the forward chain ran on x / s (range control, r2l_f2.h) and stashed that; *act_scale = s (the word behind the
    // stash format word): r2l_dw16 multiplies dW — not db, which never sees an activation — by it at the flush
    const float* act_scale = nullptr;
};

#define DW_SLAB_FLOATS (R2L_W * R2L_W + R2L_W)  // one layer: dW[256][256] then db[256], as in the flat gradient
#define DW_MAX_WGS 256
#define DW_HEAD_SLAB_MAX ((int64_t)64 * R2L_W * 1024)  // head partials: up to 64 ray slices of [256][1024]
// Regions of dw_slab (round 5: disjoint, so that the head / tail gradients of a small step can run BESIDE the body's on a second
// stream; rounds 2 - 4 let the head reuse the body's region once its reduce had consumed it):
//   [0, DW_BODY_SLAB)                      body partials, [workgroup][2][DW_SLAB_FLOATS]
//   [DW_BODY_SLAB, + DW_HEAD_SLAB_MAX)     head partials, [slice][256][1024]
//   [DW_TAIL_SLAB_BASE, + DW_TAIL_SLAB)    tail partials, [workgroup <= 512][4][256]
#define DW_BODY_SLAB ((int64_t)DW_MAX_WGS * 2 * DW_SLAB_FLOATS)
#define DW_TAIL_SLAB_BASE (DW_BODY_SLAB + DW_HEAD_SLAB_MAX)
#define DW_TAIL_SLAB ((int64_t)512 * 4 * R2L_W)

// ---- head weight gradient (r2l_backward.hip: fp32 MFMA; r2l_dw_head16.hip: fp16 MFMA of the default trio) ----------------------
struct R2LDwHeadArgs {
    const float* rays_o;
    const float* rays_d;
    const float* t_rand;
    const float* ztab;
    const float* emb;  // [N,1008] given encoding (module-boundary path) or nullptr -> recompute from the rays
    const float* gh;   // [N,256] = gx[0]
    float* slab;       // [n_slices][256][1024] per-slice partial dW (plain stores, reduced in fixed order) or nullptr
    float* grads;
    int64_t N;
    int64_t rays_per_wg;
    // default fp16 trio (r2l_dw_head16.hip): gh is multiplied by gscale (the dX chain's power-of-two scale) before it is rounded
    // to fp16 and the result by unscale = 1 / gscale; run_unless / run_if: the dX chain's status word (the fp16 kernel returns
    // at once when it is raised, the fp32 kernel launched behind it when it is not)
    float gscale = 1.0f, unscale = 1.0f;
    const float* scale_dev = nullptr;  // {gscale, 1 / gscale} on the device (generic mode); overrides the two above
    const unsigned* run_unless = nullptr;
    const unsigned* run_if = nullptr;
    bool exact = false;  // r2l_dw_head16: both operands as hi + mid, three products (exact weight gradients)
};

// Per-lane description of one encoding column k (fixed for the whole kernel): which sample / axis it reads and what it
// applies.  column k of PositionalEmbedder's output: coord c = k/21 (sample c/3, axis c%3), slot f = k%21
// (f < 10: sin(2^f x), 10 <= f < 20: cos(2^(f-10) x), f == 20: x).
struct PECol {
    int smp, ax;
    float scale;  // 2^freq (trig columns)
    int kind;     // 0 sin, 1 cos, 2 identity, 3 padding (k >= 1008)
};
__device__ __forceinline__ PECol pe_col(int k) {
    PECol c;
    if (k >= R2L_IN) { c.smp = 0; c.ax = 0; c.scale = 0.f; c.kind = 3; return c; }
    const int co = k / 21, f = k - 21 * co;
    c.smp = co / 3;
    c.ax = co - 3 * c.smp;
    c.kind = f == 20 ? 2 : (f < 10 ? 0 : 1);
    c.scale = f == 20 ? 1.f : (float)(1 << (f < 10 ? f : f - 10));
    return c;
}
// value of the column for the point x = o + d*z of one ray
__device__ __forceinline__ float pe_eval(const PECol& c, float x) {
    float s, co;
    r2l_sincos(x * c.scale, s, co);
    const float t = c.kind == 0 ? s : co;
    return c.kind == 2 ? x : (c.kind == 3 ? 0.f : t);
}

int r2l_dw_head16_launch(const R2LDwHeadArgs& a, int64_t slices, hipStream_t stream);

// fp16 weight-gradient GEMMs of the default trio (r2l_dw16.hip): operands are the fp16 stage pieces the chains stashed
// (r2l_f2.h).  run_unless: the dX chain's status word — when the step fell back to the bf16x3 chains (fp32 stash) this launch
// raises *status and returns, and the bf16x3 weight-gradient kernel launched behind it (run_if = status) does the work.
int r2l_dw16_launch(const R2LDwArgs& a, int64_t wgs, const unsigned* run_unless, hipStream_t stream);
