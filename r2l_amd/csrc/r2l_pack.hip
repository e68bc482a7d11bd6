// r2l_pack.hip — re-pack the nn.Linear weights of NeRF_v3_2 (flat state_dict-order fp32 buffer; layer shapes from
// /root/reference/model/nerf_raybased.py:500-537) into the per-lane MFMA A-operand weight streams that the chain
// kernels consume sequentially (layout contract: r2l_common.h).  Runs once per optimizer step (23.7 MB gather).
#include "r2l_common.h"

__host__ __device__ static inline int64_t pk_off_body_w(int layer) {
    return (int64_t)R2L_IN * R2L_W + R2L_W + (int64_t)layer * (R2L_W * R2L_W + R2L_W);
}

// forward stream: [head bias group | head trig groups (sample it, axis, g) | head identity groups] then per body layer
// [bias group | 32 weight groups] in execution order.  Bias group: float4 component 0 of lane (l<32) of tile t holds
// bias[32t + l]; everything else is 0 (the kernel multiplies it by the B operand [1, 0]).
__global__ void r2l_pack_fwd_kernel(const float* __restrict__ params, float* __restrict__ out, int n_block) {
    const int64_t total = (int64_t)(R2L_FWD_HEAD_GROUPS + 2 * n_block * R2L_FWD_LAYER_GROUPS) * R2L_GROUP_FLOATS;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total + R2L_STREAM_PAD;
         i += (int64_t)gridDim.x * blockDim.x) {
        if (i >= total) { out[i] = 0.f; continue; }
        const int64_t gidx = i / R2L_GROUP_FLOATS;
        const int rem = (int)(i % R2L_GROUP_FLOATS);
        const int tile = rem >> 8, lane = (rem & 255) >> 2, j = rem & 3;
        const int h = lane >> 5, o = 32 * tile + (lane & 31);
        float v;
        if (gidx < R2L_FWD_HEAD_GROUPS) {
            const int g1 = (int)gidx - 1;
            if (g1 < 0) {
                v = (j == 0 && h == 0) ? params[(int64_t)R2L_IN * R2L_W + o] : 0.f;
            } else if (g1 < R2L_HEAD_TRIG_GROUPS) {
                const int it = g1 / 15, ax = (g1 % 15) / 5, g = g1 % 5;
                const int k = ((8 * h + it) * 3 + ax) * 21 + 4 * g + j;
                v = params[(int64_t)o * R2L_IN + k];
            } else {
                const int e = 4 * (g1 - R2L_HEAD_TRIG_GROUPS) + j;
                const int k = (24 * h + e) * 21 + 20;
                v = params[(int64_t)o * R2L_IN + k];
            }
        } else {
            const int64_t gb = gidx - R2L_FWD_HEAD_GROUPS;
            const int layer = (int)(gb / R2L_FWD_LAYER_GROUPS), G = (int)(gb % R2L_FWD_LAYER_GROUPS) - 1;
            if (G < 0) {
                v = (j == 0 && h == 0) ? params[pk_off_body_w(layer) + R2L_W * R2L_W + o] : 0.f;
            } else {
                const int in = 32 * (G >> 2) + 8 * (G & 3) + 4 * h + j;
                v = params[pk_off_body_w(layer) + (int64_t)o * R2L_W + in];
            }
        }
        out[i] = v;
    }
}

// backward (dX) stream: transposed body layers in reverse execution order: (n-1,2)^T, (n-1,0)^T, ..., (0,2)^T, (0,0)^T
__global__ void r2l_pack_bwd_kernel(const float* __restrict__ params, float* __restrict__ out, int n_block) {
    const int64_t total = (int64_t)2 * n_block * R2L_LAYER_FLOATS;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total + R2L_STREAM_PAD;
         i += (int64_t)gridDim.x * blockDim.x) {
        if (i >= total) { out[i] = 0.f; continue; }
        const int64_t gidx = i / R2L_GROUP_FLOATS;
        const int rem = (int)(i % R2L_GROUP_FLOATS);
        const int tile = rem >> 8, lane = (rem & 255) >> 2, j = rem & 3;
        const int h = lane >> 5, o = 32 * tile + (lane & 31);
        const int slot = (int)(gidx / R2L_LAYER_GROUPS), G = (int)(gidx % R2L_LAYER_GROUPS);
        const int layer = 2 * n_block - 1 - slot;
        const int in = 32 * (G >> 2) + 8 * (G & 3) + 4 * h + j;
        out[i] = params[pk_off_body_w(layer) + (int64_t)in * R2L_W + o];  // (W^T)[o][in] = W[in][o]
    }
}

// ---- 16-ray cooperative layout (r2l_common.h: group16) -------------------------------------------------------------------
// forward: 63 head groups in the permuted k' order, then 16 groups per body layer (no bias groups: r2l_coop16.hip reads
// the biases from the flat parameters).  k' < 960: block B = k'/80 of 4 coordinates, coordinate ci = (k'%80)/20 of it,
// slot (k'%20): frequency f = slot/2, sin (even) or cos (odd) -> column 21*(4B+ci) + f (+10 for cos) of head.0.weight;
// k' >= 960: the identity column 21*(k'-960) + 20.
__global__ void r2l_pack_fwd16_kernel(const float* __restrict__ params, float* __restrict__ out, int n_block) {
    const int64_t total = (int64_t)(R2L_C16_HEAD_GROUPS + 2 * n_block * R2L_C16_LAYER_GROUPS) * R2L_C16_GROUP_FLOATS;
    const int64_t padded = r2l_fwd16_stream_floats(n_block);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < padded; i += (int64_t)gridDim.x * blockDim.x) {
        if (i >= total) { out[i] = 0.f; continue; }
        const int64_t gidx = i / R2L_C16_GROUP_FLOATS;
        const int rem = (int)(i % R2L_C16_GROUP_FLOATS);
        const int tile = rem >> 8, lane = (rem & 255) >> 2, e = rem & 3;
        const int kk = lane >> 4, o = 16 * tile + (lane & 15);
        float v;
        if (gidx < R2L_C16_HEAD_GROUPS) {
            const int kp = 16 * (int)gidx + 4 * kk + e;
            int col;
            if (kp < 960) {
                const int B = kp / 80, off = kp % 80, ci = off / 20, slot = off % 20, f = slot >> 1;
                col = 21 * (4 * B + ci) + ((slot & 1) ? 10 + f : f);
            } else {
                col = 21 * (kp - 960) + 20;
            }
            v = params[(int64_t)o * R2L_IN + col];
        } else {
            const int64_t gb = gidx - R2L_C16_HEAD_GROUPS;
            const int layer = (int)(gb / R2L_C16_LAYER_GROUPS), G = (int)(gb % R2L_C16_LAYER_GROUPS);
            const int in = 16 * G + 4 * kk + e;
            v = params[pk_off_body_w(layer) + (int64_t)o * R2L_W + in];
        }
        out[i] = v;
    }
}

// backward (dX): transposed body layers in reverse execution order, 16 groups each
__global__ void r2l_pack_bwd16_kernel(const float* __restrict__ params, float* __restrict__ out, int n_block) {
    const int64_t total = (int64_t)2 * n_block * R2L_C16_LAYER_GROUPS * R2L_C16_GROUP_FLOATS;
    const int64_t padded = r2l_bwd16_stream_floats(n_block);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < padded; i += (int64_t)gridDim.x * blockDim.x) {
        if (i >= total) { out[i] = 0.f; continue; }
        const int64_t gidx = i / R2L_C16_GROUP_FLOATS;
        const int rem = (int)(i % R2L_C16_GROUP_FLOATS);
        const int tile = rem >> 8, lane = (rem & 255) >> 2, e = rem & 3;
        const int kk = lane >> 4, o = 16 * tile + (lane & 15);
        const int slot = (int)(gidx / R2L_C16_LAYER_GROUPS), G = (int)(gidx % R2L_C16_LAYER_GROUPS);
        const int layer = 2 * n_block - 1 - slot;
        const int in = 16 * G + 4 * kk + e;
        out[i] = params[pk_off_body_w(layer) + (int64_t)in * R2L_W + o];  // (W^T)[o][in] = W[in][o]
    }
}

extern "C" int64_t r2l_param_count(int n_block) {
    return (int64_t)R2L_IN * R2L_W + R2L_W + (int64_t)2 * n_block * (R2L_W * R2L_W + R2L_W) + 3 * R2L_W + 3;
}

// a stream buffer = [32-ray-tile layout | 16-ray-tile layout | bf16x3 stages | fp16x2 stages (forward stream only)]; every
// kernel finds its part by offset
extern "C" int64_t r2l_fwd_stream_floats(int n_block) {
    return r2l_fwd32_stream_floats(n_block) + r2l_fwd16_stream_floats(n_block) + r2l_fwd3_stream_floats(n_block) +
           r2l_fwd2_stream_floats(n_block);
}

extern "C" int64_t r2l_bwd_stream_floats(int n_block) {
    return r2l_bwd32_stream_floats(n_block) + r2l_bwd16_stream_floats(n_block) + r2l_bwd3_stream_floats(n_block) +
           r2l_bwd2_stream_floats(n_block);
}

// layout: 32 (main + coop kernels), 16 (coop16 kernels) or 0 (both).  A caller that knows which chain variant its next
// launches use (r2l_variant_for) can skip the other half of the stream: 10 us each, 3 % of a 4096-ray step.
extern "C" int r2l_variant_for_cfg(int64_t N, const r2l_config* cfg) {
    R2L_CFG_QUERY(cfg);
    return r2l_chain_variant(N);
}
extern "C" int r2l_variant_for(int64_t N) { return r2l_variant_for_cfg(N, nullptr); }
// stream layout a forward launch with N rays reads: 16 / 32 (cooperative variants / fp32-MFMA kernels), 3 (the bf16x3
// kernel, r2l_fwd3.hip: every one-wave-per-tile forward, with or without the training stash) or 2 (fp16x2 kernel,
// r2l_fwd2.hip, with the bf16x3 stream as its fallback: forward-only launches)
extern "C" int r2l_forward_layout_for(int64_t N, int with_stash) { return r2l_forward_layout_for_cfg(N, with_stash, nullptr); }
extern "C" int r2l_forward_layout_for_cfg(int64_t N, int with_stash, const r2l_config* cfg) {
    R2L_CFG_QUERY(cfg);
    const int v = r2l_chain_variant(N);
    if (v == R2L_VARIANT_COOP16) return 16;
    if (v == R2L_VARIANT_MAIN && (with_stash ? r2l_use_trio16() : r2l_use_fwd2())) return 2;
    if (v == R2L_VARIANT_MAIN && r2l_use_fwd3()) return 3;
    return 32;
}
// transposed-stream layout the backward of an N-ray launch reads: 16 / 32 as the forward, 3 (bf16x3 dX chain) or 2 (fp16x2 dX
// chain with the bf16x3 stream behind it as range-guard fallback: r2l_pack_backward_layout(2) fills both)
extern "C" int r2l_backward_layout_for(int64_t N) { return r2l_backward_layout_for_cfg(N, nullptr); }
extern "C" int r2l_backward_layout_for_cfg(int64_t N, const r2l_config* cfg) {
    R2L_CFG_QUERY(cfg);
    const int v = r2l_chain_variant(N);
    if (v == R2L_VARIANT_COOP16) return 16;
    if (v == R2L_VARIANT_MAIN && r2l_use_fwd3()) return r2l_use_trio16() ? 2 : 3;
    return 32;
}
extern "C" int r2l_pack_forward_layout(const float* params, int n_block, float* wstream, int layout, void* stream) {
    R2L_REQUIRE(params && wstream, "r2l_pack_forward_layout: params / wstream is NULL");
    R2L_REQUIRE(n_block >= 0 && n_block <= R2L_MAX_BLOCKS, "r2l_pack_forward_layout: n_block out of range");
    R2L_REQUIRE(layout == 0 || layout == 32 || layout == 16 || layout == 3 || layout == 2, "r2l_pack_forward_layout: layout is 0, 32, 16, 3 or 2");
    if (layout == 0 || layout == 32) {
        hipLaunchKernelGGL(r2l_pack_fwd_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, params, wstream, n_block);
        R2L_CHECK(hipGetLastError());
    }
    if (layout == 0 || layout == 16) {
        hipLaunchKernelGGL(r2l_pack_fwd16_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, params,
                           wstream + r2l_fwd32_stream_floats(n_block), n_block);
        R2L_CHECK(hipGetLastError());
    }
    if (layout == 0 || layout == 3) {  // (layout 2's bf16x3 fallback stream is packed by the fallback launch itself, when it runs)
        const int rc = r2l_fwd3_pack(params, n_block, wstream + r2l_fwd32_stream_floats(n_block) + r2l_fwd16_stream_floats(n_block),
                                     (hipStream_t)stream);
        if (rc) return rc;
    }
    if (layout == 0 || layout == 2) {
        const int rc = r2l_fwd2_pack(params, n_block,
                                     wstream + r2l_fwd32_stream_floats(n_block) + r2l_fwd16_stream_floats(n_block) +
                                         r2l_fwd3_stream_floats(n_block),
                                     (hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
}
extern "C" int r2l_pack_backward_layout(const float* params, int n_block, float* wstream, int layout, void* stream) {
    R2L_REQUIRE(params && wstream, "r2l_pack_backward_layout: params / wstream is NULL");
    R2L_REQUIRE(n_block >= 0 && n_block <= R2L_MAX_BLOCKS, "r2l_pack_backward_layout: n_block out of range");
    R2L_REQUIRE(layout == 0 || layout == 32 || layout == 16 || layout == 3 || layout == 2, "r2l_pack_backward_layout: layout is 0, 32, 16, 3 or 2");
    if (layout == 0 || layout == 32) {
        hipLaunchKernelGGL(r2l_pack_bwd_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, params, wstream, n_block);
        R2L_CHECK(hipGetLastError());
    }
    if (layout == 0 || layout == 16) {
        hipLaunchKernelGGL(r2l_pack_bwd16_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, params,
                           wstream + r2l_bwd32_stream_floats(n_block), n_block);
        R2L_CHECK(hipGetLastError());
    }
    if (layout == 0 || layout == 3) {  // (layout 2's bf16x3 fallback stream is packed by the fallback launch itself, when it runs)
        const int rc = r2l_bwd3_pack(params, n_block, wstream + r2l_bwd32_stream_floats(n_block) + r2l_bwd16_stream_floats(n_block),
                                     (hipStream_t)stream);
        if (rc) return rc;
    }
    if (layout == 0 || layout == 2) {
        const int rc = r2l_bwd2_pack(params, n_block,
                                     wstream + r2l_bwd32_stream_floats(n_block) + r2l_bwd16_stream_floats(n_block) +
                                         r2l_bwd3_stream_floats(n_block),
                                     (hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int r2l_pack_forward(const float* params, int n_block, float* wstream, void* stream) {
    R2L_REQUIRE(params && wstream, "r2l_pack_forward: params / wstream is NULL");
    R2L_REQUIRE(n_block >= 0 && n_block <= R2L_MAX_BLOCKS, "r2l_pack_forward: n_block out of range");
    hipLaunchKernelGGL(r2l_pack_fwd_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, params, wstream, n_block);
    R2L_CHECK(hipGetLastError());
    hipLaunchKernelGGL(r2l_pack_fwd16_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, params,
                       wstream + r2l_fwd32_stream_floats(n_block), n_block);
    R2L_CHECK(hipGetLastError());
    const int rc = r2l_fwd3_pack(params, n_block, wstream + r2l_fwd32_stream_floats(n_block) + r2l_fwd16_stream_floats(n_block),
                                 (hipStream_t)stream);
    if (rc) return rc;
    return r2l_fwd2_pack(params, n_block,
                         wstream + r2l_fwd32_stream_floats(n_block) + r2l_fwd16_stream_floats(n_block) + r2l_fwd3_stream_floats(n_block),
                         (hipStream_t)stream);
}

extern "C" int r2l_pack_backward(const float* params, int n_block, float* wstream, void* stream) {
    R2L_REQUIRE(params && wstream, "r2l_pack_backward: params / wstream is NULL");
    R2L_REQUIRE(n_block >= 0 && n_block <= R2L_MAX_BLOCKS, "r2l_pack_backward: n_block out of range");
    hipLaunchKernelGGL(r2l_pack_bwd_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, params, wstream, n_block);
    R2L_CHECK(hipGetLastError());
    hipLaunchKernelGGL(r2l_pack_bwd16_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, params,
                       wstream + r2l_bwd32_stream_floats(n_block), n_block);
    R2L_CHECK(hipGetLastError());
    const int rc = r2l_bwd3_pack(params, n_block, wstream + r2l_bwd32_stream_floats(n_block) + r2l_bwd16_stream_floats(n_block),
                                 (hipStream_t)stream);
    if (rc) return rc;
    return r2l_bwd2_pack(params, n_block,
                         wstream + r2l_bwd32_stream_floats(n_block) + r2l_bwd16_stream_floats(n_block) + r2l_bwd3_stream_floats(n_block),
                         (hipStream_t)stream);
}
