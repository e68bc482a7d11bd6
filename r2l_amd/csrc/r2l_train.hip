// r2l_train.hip — optimizer and loss plumbing of the distillation step on flat fp32 buffers.
//   r2l_adam_step   : torch.optim.Adam (lr, betas (0.9,0.999), eps 1e-8, no weight decay; /root/reference/main.py:465-467,
//                     1406) as ONE elementwise kernel over the flat parameter / gradient / moment buffers.
//   r2l_loss_finish : sum of the per-tile squared-error partials -> mse, psnr  (helpers:19-20, main.py:1377-1378).
#include "r2l_common.h"

__global__ void r2l_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float bc1,
                                float sqrt_bc2, float gscale, const unsigned* __restrict__ skip_if) {
    if (skip_if != nullptr && __builtin_nontemporal_load(skip_if) != 0u) return;  // (r2l_adam_step_guarded)
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        // torch: exp_avg.lerp_(grad, 1-b1); exp_avg_sq.mul_(b2).addcmul_(grad, grad, 1-b2)
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
        const float vi = v[i] * b2 + (gi * gi) * (1.0f - b2);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
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
