// r2l_ssim.hip — SSIM of a rendered frame against its ground truth, one fused kernel.
//
// Test-set evaluation (main.py:254,334,384-391) calls utils/ssim_torch.py:28-56,86-94 per frame: five depthwise 11x11
// Gaussian conv2d's (zero padding 5) of img1, img2, img1^2, img2^2, img1*img2, a dozen elementwise kernels and a mean.
// Here: one launch.  A 16x16-pixel tile and its 5-pixel halo of both images sit in LDS; every thread accumulates the
// five windowed moments of its pixel over the 121 taps (same 2-D window values as the reference: g[i]*g[j] in fp32),
// forms the SSIM map value and the block reduces it; a second tiny kernel sums the per-block partials in a fixed order.
// HBM-bound by construction (2 x H x W x C x 4 B read once, halo re-reads hit L2): 3.84 MB per 400x400 frame.
#include "r2l_common.h"

#include <math.h>

namespace {

constexpr int SS_WIN = 11, SS_R = 5, SS_T = 16, SS_HALO = SS_T + 2 * SS_R;  // 26

struct SsimWindow {
    float w[SS_WIN * SS_WIN];
};

__global__ __launch_bounds__(SS_T* SS_T) void r2l_ssim_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              int H, int W, int C, SsimWindow win,
                                                              float* __restrict__ partial) {
    __shared__ float sa[SS_HALO][SS_HALO + 1], sb[SS_HALO][SS_HALO + 1];
    __shared__ float red[SS_T * SS_T / 64];
    const int tx = threadIdx.x % SS_T, ty = threadIdx.x / SS_T;
    const int x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T, c = blockIdx.z;
    for (int i = threadIdx.x; i < SS_HALO * SS_HALO; i += SS_T * SS_T) {
        const int hy = i / SS_HALO, hx = i % SS_HALO;
        const int y = y0 + hy - SS_R, x = x0 + hx - SS_R;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        const int64_t off = ((int64_t)y * W + x) * C + c;  // images are [H, W, C] as render_path holds them
        sa[hy][hx] = in ? a[off] : 0.f;
        sb[hy][hx] = in ? b[off] : 0.f;
    }
    __syncthreads();
    float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
    for (int i = 0; i < SS_WIN; ++i) {
#pragma unroll
        for (int j = 0; j < SS_WIN; ++j) {
            const float w = win.w[i * SS_WIN + j];
            const float p = sa[ty + i][tx + j], q = sb[ty + i][tx + j];
            m1 = fmaf(w, p, m1);
            m2 = fmaf(w, q, m2);
            s11 = fmaf(w, p * p, s11);
            s22 = fmaf(w, q * q, s22);
            s12 = fmaf(w, p * q, s12);
        }
    }
    const float mu1_sq = m1 * m1, mu2_sq = m2 * m2, mu12 = m1 * m2;
    const float sig1 = s11 - mu1_sq, sig2 = s22 - mu2_sq, sig12 = s12 - mu12;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float v = ((2.f * mu12 + C1) * (2.f * sig12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sig1 + sig2 + C2));
    if (y0 + ty >= H || x0 + tx >= W) v = 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0)
        partial[((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void r2l_ssim_finish_kernel(const float* __restrict__ partial, int64_t n,
                                                              float inv_count, float* __restrict__ out) {
    __shared__ float red[4];
    float v = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) v += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_count;
}

}  // namespace

extern "C" {

int64_t r2l_ssim_partial_count(int H, int W, int C) {
    return (int64_t)((H + SS_T - 1) / SS_T) * ((W + SS_T - 1) / SS_T) * C;
}

int r2l_ssim(const float* img1, const float* img2, int H, int W, int C, const float* window_host, float* partial,
             float* out, void* stream) {
    if (H <= 0 || W <= 0 || C <= 0 || C > 65535) {
        r2l_set_error_msg("r2l_ssim: bad image shape");
        return 1;
    }
    R2L_REQUIRE(img1 && img2 && partial && out, "r2l_ssim: a required pointer is NULL");
    SsimWindow win;
    if (window_host) {
        for (int i = 0; i < SS_WIN * SS_WIN; ++i) win.w[i] = window_host[i];
    } else {  // ssim_torch.py:11-25: fp32 Gaussian (sigma 1.5) normalised in fp32, outer product in fp32
        float g[SS_WIN], sum = 0.f;
        for (int x = 0; x < SS_WIN; ++x) {
            g[x] = (float)exp(-(double)((x - SS_R) * (x - SS_R)) / (2.0 * 1.5 * 1.5));
            sum += g[x];
        }
        for (int x = 0; x < SS_WIN; ++x) g[x] /= sum;
        for (int i = 0; i < SS_WIN; ++i)
            for (int j = 0; j < SS_WIN; ++j) win.w[i * SS_WIN + j] = g[i] * g[j];
    }
    dim3 grid((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, C);
    hipLaunchKernelGGL(r2l_ssim_kernel, grid, dim3(SS_T * SS_T), 0, (hipStream_t)stream, img1, img2, H, W, C, win, partial);
    R2L_CHECK(hipGetLastError());
    hipLaunchKernelGGL(r2l_ssim_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial,
                       r2l_ssim_partial_count(H, W, C), 1.f / ((float)H * (float)W * (float)C), out);
    R2L_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"
