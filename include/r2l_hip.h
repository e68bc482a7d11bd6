/* r2l_hip.h — C ABI of libr2l_hip.so, the MI355X (gfx950) implementation of the R2L hot path.
 *
 * The reference (snap-research/R2L) has no FFI layer: its hot path is a chain of PyTorch ops.  Each entry point
 * below replaces the reference op sequence cited beside it (paths are into /root/reference).  All pointers are
 * DEVICE pointers to contiguous row-major fp32 unless marked "host".  The library never allocates, frees or
 * retains caller memory; kernels are enqueued on the caller's HIP stream (`stream` = hipStream_t, 0 = default) with
 * no implicit synchronisation.  Every function returns 0 on success or a hipError_t code; r2l_last_error() gives
 * the text.  Nothing throws across this boundary.  INTEGRATION.md shows the ctypes stub a maintainer binds.
 */
#ifndef R2L_HIP_H
#define R2L_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* r2l_last_error(void);

/* ---- parameter layout ------------------------------------------------------------------------------------------
 * `params` is ONE flat fp32 buffer holding NeRF_v3_2's tensors in state_dict order (model/nerf_raybased.py:500-537):
 *   head.0.weight[256,1008] head.0.bias[256] { body.b.body.0.weight[256,256] .bias[256] body.b.body.2.weight .bias }
 *   x n_block, tail.0.weight[3,256] tail.0.bias[3].          W = 256, 16 samples/ray, L = 10 are compiled in.   */
int64_t r2l_param_count(int n_block);        /* 5 917 187 for n_block = 43 (D = 88) */
int64_t r2l_fwd_stream_floats(int n_block);  /* size of the packed forward weight stream, incl. prefetch padding */
int64_t r2l_bwd_stream_floats(int n_block);  /* size of the packed transposed (dX) weight stream                */

/* Re-pack params into the MFMA A-operand weight streams the chain kernels read (call after every weight update). */
int r2l_pack_forward(const float* params, int n_block, float* wstream, void* stream);
int r2l_pack_backward(const float* params, int n_block, float* wstream_bwd, void* stream);

/* ---- student forward -------------------------------------------------------------------------------------------
 * rgb[N,3] = NeRF_v3_2.forward(PositionalEmbedder(10)(PointSampler.sample_train(rays_o, rays_d, perturb)))
 *   replaces model/nerf_raybased.py:114-126 (sample_train), :198-208 (PositionalEmbedder.__call__),
 *   :461-465 (ResMLP.forward), :539-544 (NeRF_v3_2.forward); call site main.py:1371-1374 / 220-230.
 * ztab[32] = z_lower[16] ++ z_span[16]; depth of sample s = z_lower[s] + z_span[s]*t_rand[ray,s], or z_lower[s]
 * when t_rand == NULL (perturb == 0; then z_lower = PointSampler.z_vals).
 * save_x [(n_block+1),N,256] / save_t [n_block,N,256]: optional activation stash for r2l_backward (both or none). */
int r2l_forward_rays(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab,
                     const float* wstream, const float* params, int n_block, float* rgb, float* save_x,
                     float* save_t, int64_t N, void* stream);

/* rgb[H*W,3] for a whole frame from a camera pose: PointSampler.sample_test (model/nerf_raybased.py:80-102) fused in
 * front of the same chain; call sites main.py:300-309 (render_path) and main.py:401-404 (render_func, --benchmark).
 * c2w_host12: HOST pointer to the row-major [3,4] camera-to-world matrix. */
int r2l_forward_pose(const float* c2w_host12, int H, int W, float focal, const float* ztab, const float* wstream,
                     const float* params, int n_block, float* rgb, void* stream);

/* rgb[N,3] = NeRF_v3_2.forward(emb[N,1008])  — the module-boundary form (model/nerf_raybased.py:539-544) for callers
 * that still run their own sampler/embedder. */
int r2l_forward_emb(const float* emb, const float* wstream, const float* params, int n_block, float* rgb,
                    float* save_x, float* save_t, int64_t N, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* R2L_HIP_H */
