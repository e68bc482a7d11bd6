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
    // range guard of the fp16 variant (r2l_dw_body3c_kernel<3, true>): it raises *status when an operand value leaves fp16's
    // safe range; the bf16 variant launched behind it with run_if = status then redoes the launch (else returns at once)
    unsigned* status = nullptr;
    const unsigned* run_if = nullptr;
};

#define DW_SLAB_FLOATS (R2L_W * R2L_W + R2L_W)  // one layer: dW[256][256] then db[256], as in the flat gradient
#define DW_MAX_WGS 256
#define DW_HEAD_SLAB_MAX ((int64_t)64 * R2L_W * 1024)  // head partials: up to 64 ray slices of [256][1024] at the slab start

// fp16 weight-gradient GEMMs of the default trio (r2l_dw16.hip): operands are the fp16 stage pieces the chains stashed
// (r2l_f2.h).  run_unless: the dX chain's status word — when the step fell back to the bf16x3 chains (fp32 stash) this launch
// raises *status and returns, and the bf16x3 weight-gradient kernel launched behind it (run_if = status) does the work.
int r2l_dw16_launch(const R2LDwArgs& a, int64_t wgs, const unsigned* run_unless, hipStream_t stream);
