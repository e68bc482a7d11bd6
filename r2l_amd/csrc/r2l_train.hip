// r2l_train.hip — optimizer and loss plumbing of the distillation step on flat fp32 buffers.
//   r2l_adam_step   : torch.optim.Adam (lr, betas (0.9,0.999), eps 1e-8, no weight decay; /root/reference/main.py:465-467,
//                     1406) as ONE elementwise kernel over the flat parameter / gradient / moment buffers.
//   r2l_loss_finish : sum of the per-tile squared-error partials -> mse, psnr  (helpers:19-20, main.py:1377-1378).
//   r2l_adam_step_packed (round 5): the same update with the re-pack of the fp16x2 weight streams folded in — the optimizer
//                     kernel already holds every new body weight in registers, so it writes their (hi, mid) stage pieces of the
//                     forward AND the transposed backward stream itself (through a 32 x 32 LDS tile: both orders leave as
//                     contiguous 1 KiB pieces); one small kernel behind it packs the head / bias stages for the activation
//                     scale and commits it.  Two launches instead of four (adam, pack_fwd2, commit, pack_bwd2) per step.
#include "r2l_f2.h"

struct R2LAdamK {
    float step_size, b1, b2, eps, sqrt_bc2, gscale;
};
// one parameter: torch.optim.Adam's op sequence (exp_avg.lerp_(grad, 1-b1); exp_avg_sq.mul_(b2).addcmul_(grad, grad, 1-b2))
__device__ __forceinline__ float r2l_adam_one(float p, float g, float& m, float& v, const R2LAdamK& k) {
    const float gi = g * k.gscale;
    const float mi = m + (gi - m) * (1.0f - k.b1);
    const float vi = v * k.b2 + (gi * gi) * (1.0f - k.b2);
    m = mi;
    v = vi;
    const float denom = sqrtf(vi) / k.sqrt_bc2 + k.eps;
    return p - k.step_size * (mi / denom);
}

__global__ void r2l_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float bc1,
                                float sqrt_bc2, float gscale, const unsigned* __restrict__ skip_if) {
    if (skip_if != nullptr && __builtin_nontemporal_load(skip_if) != 0u) return;  // (r2l_adam_step_guarded)
    const R2LAdamK k{lr / bc1, b1, b2, eps, sqrt_bc2, gscale};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float mi = m[i], vi = v[i];
        p[i] = r2l_adam_one(p[i], g[i], mi, vi, k);
        m[i] = mi;
        v[i] = vi;
    }
}

// ---- Adam + re-pack of the fp16x2 streams ----------------------------------------------------------------------------------
// Workgroups [0, 128 n_block): one 32 x 32 tile (output rows 32 To .., input columns 32 Ti ..) of one body weight matrix each:
// update (16-byte loads / stores: 128-byte row segments), new values -> LDS, then the tile's two stage pieces of the forward
// stream (stage 64 + 17 layer + 1 + kb, kb = 2 Ti + r: element (tile To, lane (i, h), s) = W[32 To + i][32 Ti + 16 r + 8 (s >> 2) +
// 4 h + (s & 3)], r2l_f2.h f2_pack_fwd_element) and of the transposed stream of the dX chain (r2l_bwd2.hip r2l_pack_bwd2_kernel:
// block b's W2^T stages 34 (n_block - 1 - b) + 1 + kb, W1^T stages + 18 + kb, kb = 2 To + r: element (tile Ti, lane (i, h), s) =
// W[32 To + 16 r + 8 (s >> 2) + 4 h + (s & 3)][32 Ti + i]) as (hi, mid) fp16 halves, 8 bytes per thread and half: every piece
// leaves as one contiguous KiB.  The remaining workgroups: every other parameter (head, biases, tail), plain update.
#define AP_REST_WGS 1024
__global__ __launch_bounds__(256) void r2l_adam_pack_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                            float* __restrict__ v, int n_block, R2LAdamK k,
                                                            const unsigned* __restrict__ skip_if,
                                                            unsigned short* __restrict__ out_f, unsigned short* __restrict__ out_b,
                                                            unsigned* status) {
    if (skip_if != nullptr && __builtin_nontemporal_load(skip_if) != 0u) return;
    __shared__ float tl[32][33];
    const int t = (int)threadIdx.x;
    const int n_body = 128 * n_block;  // 2 n_block layers x 64 tiles
    if ((int)blockIdx.x < n_body) {
        const int layer = (int)blockIdx.x >> 6, To = ((int)blockIdx.x >> 3) & 7, Ti = (int)blockIdx.x & 7;
        const int r = t >> 3, c4 = (t & 7) * 4;
        const int64_t at = f2_off_body_w(layer) + (int64_t)(32 * To + r) * R2L_W + 32 * Ti + c4;
        f32x4 pv = *reinterpret_cast<const f32x4*>(p + at);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + at);
        f32x4 mv = *reinterpret_cast<const f32x4*>(m + at), vv = *reinterpret_cast<const f32x4*>(v + at);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float mi = mv[e], vi = vv[e];
            pv[e] = r2l_adam_one(pv[e], gv[e], mi, vi, k);
            mv[e] = mi; vv[e] = vi;
            tl[r][c4 + e] = pv[e];
        }
        *reinterpret_cast<f32x4*>(p + at) = pv;
        *reinterpret_cast<f32x4*>(m + at) = mv;
        *reinterpret_cast<f32x4*>(v + at) = vv;
        __syncthreads();
        typedef unsigned short ap_u16x4 __attribute__((ext_vector_type(4)));
        const int r2 = t >> 7, lane = (t >> 1) & 63, sh = t & 1, i = lane & 31, h = lane >> 5;
        const int in_local = 16 * r2 + 8 * sh + 4 * h;
        ap_u16x4 fh, fm, bh, bm;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float wf = tl[i][in_local + e], wb = tl[in_local + e][i];
            const _Float16 fhi = (_Float16)wf, bhi = (_Float16)wb;
            fh[e] = f2_bits(fhi); fm[e] = f2_bits((_Float16)(wf - (float)fhi));
            bh[e] = f2_bits(bhi); bm[e] = f2_bits((_Float16)(wb - (float)bhi));
        }
        unsigned short* sf = out_f + (int64_t)(64 + 17 * layer + 1 + 2 * Ti + r2) * (F2_STAGE_BYTES / 2) + (To * 64 + lane) * 8 + 4 * sh;
        *reinterpret_cast<ap_u16x4*>(sf) = fh;
        *reinterpret_cast<ap_u16x4*>(sf + 8 * 64 * 8) = fm;
        const int b = layer >> 1, kbb = 2 * To + r2;
        const int64_t gb = 34 * (int64_t)(n_block - 1 - b) + ((layer & 1) ? 1 + kbb : 18 + kbb);
        unsigned short* sb = out_b + gb * (F2_STAGE_BYTES / 2) + (Ti * 64 + lane) * 8 + 4 * sh;
        *reinterpret_cast<ap_u16x4*>(sb) = bh;
        *reinterpret_cast<ap_u16x4*>(sb + 8 * 64 * 8) = bm;
        return;
    }
    // range control (r2l_f2.h): ONE thread of this launch closes the amax epoch and commits the activation scale the head / bias
    // stages are packed for by the kernel behind this one (nothing in this launch reads the status words: a kernel boundary
    // instead of the device-wide fence a "last workgroup commits" costs — measured: that fence made a 2400-workgroup pack 156 us)
    if ((int)blockIdx.x == n_body && t == 0) f2_commit_scale(status, f2_next_scale(status), false);
    // everything that is not a body weight: [0, head) ++ the body biases ++ the tail
    const int64_t head = f2_off_body_w(0), n_bias = (int64_t)2 * n_block * R2L_W, tail = 3 * R2L_W + 3;
    const int64_t total = head + n_bias + tail;
    for (int64_t j = (int64_t)((int)blockIdx.x - n_body) * 256 + t; j < total; j += (int64_t)AP_REST_WGS * 256) {
        int64_t at = j;
        if (j >= head + n_bias) at = f2_off_tail_w(n_block) + (j - head - n_bias);
        else if (j >= head) at = f2_off_body_b((int)((j - head) >> 8)) + ((j - head) & 255);
        float mi = m[at], vi = v[at];
        p[at] = r2l_adam_one(p[at], g[at], mi, vi, k);
        m[at] = mi;
        v[at] = vi;
    }
}
// behind it: the forward stream's head and bias stages for the activation scale the optimizer kernel just committed (r2l_f2.h:
// range control — the body's weight stages do not depend on it)
__global__ void r2l_pack_fwd2_nonbody_kernel(const float* __restrict__ params, unsigned short* __restrict__ out, int n_block,
                                             const unsigned* __restrict__ status, const unsigned* __restrict__ skip_if) {
    if (skip_if != nullptr && __builtin_nontemporal_load(skip_if) != 0u) return;  // (parameters untouched: the stream stands)
    const float inv = status[F2S_MAGIC] == F2_MAGIC ? __builtin_bit_cast(float, status[F2S_INV]) : 1.0f;
    f2_pack_fwd_nonbody(params, out, n_block, inv, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}
extern "C" int r2l_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, int step, float grad_scale, void* stream) {
    return r2l_adam_step_guarded(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, grad_scale, nullptr, stream);
}
extern "C" int r2l_adam_step_guarded(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                     float beta1, float beta2, float eps, int step, float grad_scale, const unsigned* skip_if,
                                     void* stream) {
    if (n <= 0) return 0;
    R2L_REQUIRE(params && grads && exp_avg && exp_avg_sq, "r2l_adam_step: a buffer is NULL");
    R2L_REQUIRE(step >= 1, "r2l_adam_step: step counts from 1");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(r2l_adam_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), grad_scale, skip_if);
    R2L_CHECK(hipGetLastError());
    return 0;
}

extern "C" int r2l_adam_step_packed(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n_block, float lr,
                                    float beta1, float beta2, float eps, int step, float grad_scale, const unsigned* skip_if,
                                    float* wstream_fwd, float* wstream_bwd, void* stream) {
    R2L_REQUIRE(params && grads && exp_avg && exp_avg_sq && wstream_fwd && wstream_bwd, "r2l_adam_step_packed: a buffer is NULL");
    R2L_REQUIRE(step >= 1 && n_block >= 0 && n_block <= R2L_MAX_BLOCKS, "r2l_adam_step_packed: step counts from 1; n_block out of range");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const R2LAdamK k{lr / (float)bc1, beta1, beta2, eps, (float)sqrt(bc2), grad_scale};
    float* w2f = wstream_fwd + r2l_fwd32_stream_floats(n_block) + r2l_fwd16_stream_floats(n_block) + r2l_fwd3_stream_floats(n_block);
    float* w2b = wstream_bwd + r2l_bwd32_stream_floats(n_block) + r2l_bwd16_stream_floats(n_block) + r2l_bwd3_stream_floats(n_block);
    unsigned* status = reinterpret_cast<unsigned*>(w2f + r2l_fwd2_status_offset(n_block));
    hipLaunchKernelGGL(r2l_adam_pack_kernel, dim3((unsigned)(128 * n_block + AP_REST_WGS)), dim3(256), 0, (hipStream_t)stream, params,
                       grads, exp_avg, exp_avg_sq, n_block, k, skip_if, reinterpret_cast<unsigned short*>(w2f),
                       reinterpret_cast<unsigned short*>(w2b), status);
    R2L_CHECK(hipGetLastError());
    // (one element per thread: the head's gather loads are latency-bound)
    hipLaunchKernelGGL(r2l_pack_fwd2_nonbody_kernel, dim3((unsigned)((64 + 2 * n_block) * 16)), dim3(256), 0, (hipStream_t)stream, params,
                       reinterpret_cast<unsigned short*>(w2f), n_block, status, skip_if);
    R2L_CHECK(hipGetLastError());
    return 0;
}

// out[0] = sum(partials) / denom (mse*lw), out[1] = psnr = -10 log10(out[0]); single block, deterministic order
__global__ void r2l_loss_finish_kernel(const float* __restrict__ partial, int64_t n, float inv_denom,
                                       float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mse = (float)(red[0] * (double)inv_denom);
        out[0] = mse;
        out[1] = -10.0f * logf(mse) / logf(10.0f);
    }
}

extern "C" int r2l_loss_finish(const float* sqerr_partial, int64_t n_partial, float inv_denom, float* out2,
                               void* stream) {
    R2L_REQUIRE(out2 && (sqerr_partial || n_partial <= 0) && n_partial >= 0, "r2l_loss_finish: NULL buffer / negative count");
    hipLaunchKernelGGL(r2l_loss_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sqerr_partial, n_partial,
                       inv_denom, out2);
    R2L_CHECK(hipGetLastError());
    return 0;
}
