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

/* ---- explicit dispatch ------------------------------------------------------------------------------------------
 * Which kernel family serves a call is the library's choice (by ray count) unless the host says otherwise.  Hosts say so
 * with an r2l_config passed to the *_cfg form of an entry point (cfg == NULL and all-zero fields = AUTO = the plain entry
 * point).  AUTO fields still honour the R2L_* environment switches of README.md (test / A-B overrides); a non-zero field
 * wins over the environment.  A forward with a stash and the backward that consumes it must be given the same config, and
 * so must the r2l_*_layout_for_cfg queries that decide which weight-stream layout to pack for them.
 * A field outside its enum / range (or a non-zero reserved word) is an error: launch entry points return
 * hipErrorInvalidValue (r2l_last_error says which field), the host-side *_for_cfg / *_ok_cfg queries return -1.
 * (The reference has no counterpart: its dtype / device choices are torch globals; this replaces `setenv` for hosts that
 * are not this repo's Python.) */
enum { R2L_PRECISION_AUTO = 0,
       R2L_PRECISION_FP16X2 = 1,     /* 3 fp16 MFMA products per fp32 product (~2^-21), range-guarded; dW per dw_mode     */
       R2L_PRECISION_BF16X3 = 2,     /* 6 bf16 products per fp32 product (fp32-exact products) everywhere                  */
       R2L_PRECISION_FP32_MFMA = 3   /* v_mfma_f32_32x32x2_f32 everywhere                                                  */ };
enum { R2L_TILING_AUTO = 0,
       R2L_TILING_WAVE_PER_TILE = 1, /* one wavefront owns a 32-ray tile ("main")                                          */
       R2L_TILING_COOP_RETIRED = 2,  /* (rounds 1 - 4: fp32-MFMA cooperative kernels, 32-ray tiles; retired: hipErrorInvalidValue) */
       R2L_TILING_COOP16 = 3,        /* fp32-MFMA cooperative kernels, 16-ray tile per workgroup                           */
       R2L_TILING_COOPF = 4          /* fp16x2 cooperative kernels (r2l_coopf_*), coop_tiles ray tiles per workgroup       */ };
enum { R2L_DW_AUTO = 0,
       R2L_DW_FP16 = 1,              /* fp16 trio: weight-gradient GEMMs on the operands' fp16 hi halves, 1 product        */
       R2L_DW_EXACT = 2              /* fp16 trio: both operands as hi + mid (22 bits), 3 products: fp32-grade dW          */ };
typedef struct r2l_config {
    int precision;    /* R2L_PRECISION_*                                                                                  */
    int tiling;       /* R2L_TILING_*                                                                                     */
    int coop_tiles;   /* 0 auto, 1 or 2: 32-ray tiles per workgroup of the fp16x2 cooperative kernels; 3 (opt-in, never auto):
                         mixed grid (two-tile and one-tile workgroups, one per CU) for tile counts between one and two per CU */
    int reserve_cus;  /* 0 auto (R2L_RESERVE_CUS or none), n > 0: CUs the persistent weight-gradient kernels leave free
                         for collectives running beside them, -1: none                                                    */
    int dw_mode;      /* R2L_DW_*                                                                                         */
    int reserved[3];  /* must be 0                                                                                        */
} r2l_config;

/* Errors: every entry point returning int gives 0 on success, else a hipError_t value (hipErrorInvalidValue = 1 for a NULL
 * required pointer, a size / n_block (0 .. 1024) / layout / parts argument out of range, or a bad r2l_config — checked before
 * anything is launched) or R2L_ERR_RCCL_BASE + ncclResult_t; r2l_last_error() holds the text for the calling thread.  N == 0
 * (R == 0, K == 0) is a successful no-op. */

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
/* Per-layout form: layout = 32 (fp32-MFMA one-wave-per-tile kernels), 16 (16-ray cooperative
 * kernels), 3 (bf16 (hi, mid, lo) stages: r2l_fwd3.hip / r2l_bwd3.hip), 2 (fp16 (hi, mid) stages: r2l_fwd2.hip /
 * r2l_bwd2.hip and their cooperative forms r2l_coopf_*.hip; the bf16 stream behind them is their range-guard fallback and
 * is packed by the fallback launch itself when — and only when — it runs) or 0 (all parts, = r2l_pack_forward/backward).
 * r2l_variant_for(N) tells which chain variant a call with N rays will take (0 main — incl. the cooperative fp16x2 kernels
 * of small launches —, 2 coop16: layout 16; 1 named the retired 32-ray cooperative family), honouring R2L_FORCE_VARIANT
 * (main | coopf | coop16). */
int r2l_variant_for(int64_t N);
/* Within the fp16 trio (layout 2): 0 = the one-wave-per-tile chains serve a launch of N rays, 1 / 2 = the cooperative chains
 * with that many 32-ray tiles per workgroup, 3 = their MIXED grid (only when pinned; tile counts between one and two per CU:
 * tiles - n_cu two-tile workgroups + 2 n_cu - tiles one-tile ones, one workgroup on every CU) (<= 16 384 rays, and 32 769 .. 49 152 rays:
 * csrc/r2l_common.h r2l_use_coopf; R2L_FORCE_VARIANT=main|coopf and R2L_COOPF_TILES=1|2|3 pin it).  1 / 2 / 3 give
 * bit-identical results (every tile takes the same path); 0 agrees with them within rounding. */
int r2l_coop_tiles_for(int64_t N, int n_block);
int r2l_forward_layout_for(int64_t N, int with_stash); /* 16, 32, 3 = bf16x3 stage stream, 2 = fp16x2 stage stream (+ the
                                                         * bf16x3 one behind it as range-guard fallback) */
/* Same for the transposed stream r2l_backward reads for N rays (r2l_pack_backward_layout takes the value). */
int r2l_backward_layout_for(int64_t N);
/* The same four queries for calls that will be made with an r2l_config; cfg == NULL: the plain forms. */
int r2l_variant_for_cfg(int64_t N, const r2l_config* cfg);
int r2l_coop_tiles_for_cfg(int64_t N, int n_block, const r2l_config* cfg);
int r2l_forward_layout_for_cfg(int64_t N, int with_stash, const r2l_config* cfg);
int r2l_backward_layout_for_cfg(int64_t N, const r2l_config* cfg);
int r2l_pack_forward_layout(const float* params, int n_block, float* wstream, int layout, void* stream);
int r2l_pack_backward_layout(const float* params, int n_block, float* wstream, int layout, void* stream);

/* ---- student forward -------------------------------------------------------------------------------------------
 * rgb[N,3] = NeRF_v3_2.forward(PositionalEmbedder(10)(PointSampler.sample_train(rays_o, rays_d, perturb)))
 *   replaces model/nerf_raybased.py:114-126 (sample_train), :198-208 (PositionalEmbedder.__call__),
 *   :461-465 (ResMLP.forward), :539-544 (NeRF_v3_2.forward); call site main.py:1371-1374 / 220-230.
 * ztab[32] = z_lower[16] ++ z_span[16]; depth of sample s = z_lower[s] + z_span[s]*t_rand[ray,s], or z_lower[s]
 * when t_rand == NULL (perturb == 0; then z_lower = PointSampler.z_vals).
 * save_x [(n_block+1) slots] / save_t [n_block slots] of r2l_stash_slot_floats(N) floats each: optional activation stash
 * for r2l_backward (both or none); opaque to the caller (row-major fp32 [Np,256], chunked fp32 or fp16 stage pieces per slot,
 * depending on the kernel family the library picks for N and the R2L_* environment — which must be the same for this call
 * and the r2l_backward that consumes the stash; Np = r2l_padded_rows(N) = N rounded up to 32). */
int r2l_forward_rays(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab,
                     const float* wstream, const float* params, int n_block, float* rgb, float* save_x,
                     float* save_t, int64_t N, void* stream);

/* rgb[H*W,3] for a whole frame from a camera pose: PointSampler.sample_test (model/nerf_raybased.py:80-102) fused in
 * front of the same chain; call sites main.py:300-309 (render_path) and main.py:401-404 (render_func, --benchmark).
 * c2w_host12: HOST pointer to the row-major [3,4] camera-to-world matrix. */
int r2l_forward_pose(const float* c2w_host12, int H, int W, float focal, const float* ztab, const float* wstream,
                     const float* params, int n_block, float* rgb, void* stream);
/* r2l_forward_rays / r2l_forward_pose with explicit dispatch (r2l_config above; cfg == NULL: as the plain forms). */
int r2l_forward_rays_cfg(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab,
                         const float* wstream, const float* params, int n_block, float* rgb, float* save_x,
                         float* save_t, int64_t N, void* stream, const r2l_config* cfg);
int r2l_forward_pose_cfg(const float* c2w_host12, int H, int W, float focal, const float* ztab, const float* wstream,
                         const float* params, int n_block, float* rgb, void* stream, const r2l_config* cfg);
/* K frames in ONE launch — the test-set loop of main.py:300-309 (render_path renders every test pose: 200 at testskip=1)
 * without a launch and a partly filled last round of workgroups per frame: rgb[K*H*W,3], frame k from the DEVICE table
 * c2w_dev[K][3][4].  Same values as K calls of r2l_forward_pose.  One-wave-per-tile tilings only (an error otherwise). */
int r2l_forward_poses_cfg(const float* c2w_dev, int K, int H, int W, float focal, const float* ztab, const float* wstream,
                          const float* params, int n_block, float* rgb, void* stream, const r2l_config* cfg);

/* ---- range control of the fp16 kernels (precision FP16X2, the default) -----------------------------------------------------
 * fp16 ends at 65504; the reference's fp32 activations (model/nerf_raybased.py:461-465: an un-normalised 88-layer residual
 * stream) are not bounded a priori.  The library keeps them in range by itself, on the device: the forward weight stream is
 * packed for a power-of-two activation scale s (head weights and all biases divided by s — a ReLU net is positively
 * homogeneous, so every activation is divided by s and nothing else changes; the kernels multiply by s where values leave
 * the chain: exact), every launch records its largest |activation|, and r2l_pack_forward* picks s for the next launches from
 * it (s = 1 while activations stay below 8192: bit-identical to an unscaled stream).  A launch that nevertheless meets a
 * value >= 32768 is redone by the bf16x3 kernel launched behind it (no host involvement, results still exact products), the
 * stream is re-packed for a larger s by that fallback, and the NEXT launch is back on the fp16 kernels.  The training
 * backward does the same with the power-of-two scale of its gradient chain, step to step (r2l_backward_status_words).
 * Hosts only need to (a) zero-fill `wstream` / `wstream_bwd` once after allocating them (stale contents of a previous
 * instance would be taken for history: harmless, but runs are then not reproducible bit for bit) and (b) may read the
 * telemetry below, e.g. to log head-room.  Words (uint32 / float bit patterns) of the forward area:
 *   [0] guard flag of the launch in flight   [1] largest |activation| / s since the scale was last chosen (float)
 *   [2] s (float)   [3] 1 / s   [5] launches that fell back to the bf16x3 kernel   [6] largest |activation| (unscaled) of the
 *   previous epoch (float)   [7] times s changed;   others: private.
 * of the backward area: [0] flag: this step ran on the bf16x3 kernels   [4] gradient scale (float)   [8] largest |chain value|
 *   x scale of this step (float)   [10] steps that fell back   [11] largest unscaled |chain value| of the last clean step. */
const unsigned* r2l_forward_status_words(const float* wstream, int n_block);
const unsigned* r2l_backward_status_words(const float* wstream_bwd, int n_block);

/* rgb[N,3] = NeRF_v3_2.forward(emb[N,1008])  — the module-boundary form (model/nerf_raybased.py:539-544) for callers
 * that still run their own sampler/embedder. */
int r2l_forward_emb(const float* emb, const float* wstream, const float* params, int n_block, float* rgb,
                    float* save_x, float* save_t, int64_t N, void* stream);
/* ... with a config: precision = bf16x3 (fp16x2 is served by the same kernels: this path has no range-guard fallback) runs a
 * forward-only launch (save_x == save_t == NULL) as head on the fp32 MFMA -> X_0 in x0_scratch (r2l_padded_rows(N) * 256 floats,
 * caller-owned) -> body + tail on the bf16x3 chain; every other case is r2l_forward_emb (x0_scratch may then be NULL). */
int r2l_forward_emb_cfg(const float* emb, const float* wstream, const float* params, int n_block, float* rgb, float* save_x,
                        float* save_t, int64_t N, float* x0_scratch, void* stream, const r2l_config* cfg);

/* ---- student backward + optimizer ------------------------------------------------------------------------------
 * Replaces loss.backward() of main.py:1377-1404 for the R2L branch (autograd over the ops above, anomaly mode on in the
 * reference: model/nerf_raybased.py:4) by three hand-written stages: the dX chain through the transposed layers, the
 * per-layer weight-gradient GEMMs (reduction over rays) and the head gradient with the encoding recomputed.
 *   MSE mode  (target != NULL): dL/drgb = grad_scale * (rgb - target)   [grad_scale = 2*lw_rgb / (3*N_global)];
 *                               sqerr_partial[r2l_num_tiles(N)] receives per-32-ray sums of (rgb-target)^2.
 *   generic   (target == NULL): dL/drgb = drgb[N,3] supplied by the caller (autograd bridge); grad_scale is ignored (the
 *                               power-of-two scale the fp16 kernels run the chain on is derived from max |drgb| on the device).
 * Head input: emb[N,1008] if given, else recomputed from (rays_o, rays_d, t_rand, ztab) exactly as the forward did.
 * save_x/save_t: the stash written by the forward of the same N (r2l_forward_rays, or r2l_forward_emb when emb is given).
 * Scratch owned by the caller: dpre[N,3], gx[(n_block+1) slots], gt[n_block slots] (slots of r2l_stash_slot_floats(N)
 * floats, like the stash), dw_slab[r2l_dw_slab_floats()] (204 MB: per-workgroup partial body-layer gradients, per-slice head
 * partials and tail partials in disjoint regions, each summed in a fixed order: bit-reproducible; NULL selects fp32 atomics
 * instead, no scratch but run-to-run rounding differences).  A call that computes body AND head gradients of a small launch
 * (<= 16 384 rays) runs head + tail on a second stream of the library's own beside the body's, forked behind the dX chain and
 * joined before the call returns: for the caller's stream nothing changes (R2L_NO_DW_OVERLAP=1 turns it off).  Gradients are ACCUMULATED into `grads` (flat, same layout as params): zero it first unless
 * accumulation is wanted. */
int64_t r2l_num_tiles(int64_t N);
int64_t r2l_padded_rows(int64_t N);
/* Floats per stash slot: size save_x / gx as (n_block+1) * r2l_stash_slot_floats(N) floats and save_t / gt as
 * n_block * r2l_stash_slot_floats(N) (Np*264: 1 KiB per ray of data + the forward's ReLU mask words; the layout inside the
 * buffers is private to the library). */
int64_t r2l_stash_slot_floats(int64_t N);
int64_t r2l_dw_slab_floats(void);
int r2l_backward(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab, const float* emb,
                 const float* rgb, const float* target, const float* drgb, const float* save_x, const float* save_t,
                 const float* wstream_bwd, const float* params, int n_block, float grad_scale, float* dpre, float* gx,
                 float* gt, float* sqerr_partial, float* grads, float* dw_slab, int64_t N, void* stream);

/* The same backward cut into stages for data-parallel hosts (replaces nn.DataParallel's ReduceAddCoalesced after
 * loss.backward(), main.py:37-42,472-479,1404): `parts` = OR of the R2L_BWD_* bits; R2L_BWD_BODY computes the weight and
 * bias gradients of the body layers [layer_lo, layer_hi) of the 2*n_block (layer 2b = body.b.body.0, 2b+1 = body.b.body.2),
 * which are complete in `grads` once the call's kernels have run — the host can all-reduce that contiguous range of the
 * flat buffer while later stages still execute.  R2L_BWD_CHAIN must come first in a step; all stages of a step go to
 * one stream (they share dw_slab).  r2l_backward(...) == r2l_backward_part(..., R2L_BWD_ALL, 0, 2*n_block). */
#define R2L_BWD_CHAIN 1
#define R2L_BWD_BODY 2
#define R2L_BWD_HEAD 4
#define R2L_BWD_TAIL 8
#define R2L_BWD_ALL 15
int r2l_backward_part(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab, const float* emb,
                      const float* rgb, const float* target, const float* drgb, const float* save_x, const float* save_t,
                      const float* wstream_bwd, const float* params, int n_block, float grad_scale, float* dpre, float* gx,
                      float* gt, float* sqerr_partial, float* grads, float* dw_slab, int64_t N, void* stream, int parts,
                      int layer_lo, int layer_hi);
/* r2l_backward_part with explicit dispatch: the config the forward of this step was given (cfg == NULL: the plain form;
 * parts = R2L_BWD_ALL, layers [0, 2 n_block) = r2l_backward). */
int r2l_backward_part_cfg(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab, const float* emb,
                          const float* rgb, const float* target, const float* drgb, const float* save_x,
                          const float* save_t, const float* wstream_bwd, const float* params, int n_block,
                          float grad_scale, float* dpre, float* gx, float* gt, float* sqerr_partial, float* grads,
                          float* dw_slab, int64_t N, void* stream, int parts, int layer_lo, int layer_hi,
                          const r2l_config* cfg);
/* A data-parallel host with idle CUs (small per-GPU batches: the cooperative chains occupy one CU per 32 or 64 rays) can cut the
 * dX chain itself: R2L_BWD_CHAIN with a layer range [layer_lo, layer_hi) that is a proper sub-range of [0, 2 n_block) runs ONE
 * SEGMENT of the chain — whole blocks, issued from the top (layer_hi = 2 n_block first) down to layer_lo = 0, each on the same
 * stream — and the weight gradients of a finished segment (R2L_BWD_BODY over the same range, on ANOTHER stream, behind an
 * event) and their all-reduce run beside the next segment.  Results are bit-identical to the uncut chain.  Only where
 * r2l_chain_segments_ok_cfg(N, n_block, cfg) says 1, and only together with R2L_BWD_NOFALLBACK on every stage of the step:
 * the bf16x3 fallback kernels are not launched, so a step whose chain raised the status word (*r2l_backward_status_word != 0
 * afterwards: range guard, or the forward had fallen back) has NO valid gradient — r2l_adam_step_guarded skips its update on
 * the device, the host notices later (no sync) and goes back to the uncut form.  r2l_amd/train_step.py is the worked example. */
#define R2L_BWD_NOFALLBACK 16
int r2l_chain_segments_ok_cfg(int64_t N, int n_block, const r2l_config* cfg);
const unsigned* r2l_backward_status_word(const float* wstream_bwd, int n_block);

/* ---- gradient all-reduce for hosts without torch.distributed ----------------------------------------------------------
 * The one exchange of data-parallel training (replaces nn.DataParallel's ReduceAddCoalesced + parameter broadcast,
 * main.py:37-42,472-479): in-place SUM of grads[n] over the ranks, RCCL over xGMI, enqueued on the caller's stream.  RCCL
 * is dlopen'ed on first use (librccl.so.1; override with R2L_RCCL_PATH).  One communicator per process = per GPU (the
 * device current at r2l_allreduce_init).  Rank 0 makes the 128-byte id, the host hands it to the other ranks.  Ranges of
 * the flat buffer finished by r2l_backward_part can be reduced one by one (the call is asynchronous on `stream`).
 * Errors: R2L_ERR_RCCL_BASE + ncclResult_t (R2L_ERR_RCCL_BASE alone: library missing / bad arguments). */
#define R2L_ERR_RCCL_BASE 10000
typedef struct r2l_comm r2l_comm;
int r2l_allreduce_unique_id(void* id_out128);
int r2l_allreduce_init(const void* id128, int world, int rank, r2l_comm** out);
int r2l_grad_allreduce(r2l_comm* comm, float* grads, int64_t n, void* stream);
int r2l_allreduce_destroy(r2l_comm* comm);

/* torch.optim.Adam(lr, betas, eps, weight_decay 0) on flat buffers (main.py:465-467, 1406); `step` counts from 1;
 * grads are multiplied by grad_scale first (1/world_size after a sum all-reduce). */
int r2l_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                  float beta1, float beta2, float eps, int step, float grad_scale, void* stream);
/* ... the same update unless *skip_if != 0 (a device word, e.g. r2l_backward_status_word of a R2L_BWD_NOFALLBACK step, or its
 * MAX over the ranks): then parameters and moments are left untouched; skip_if == NULL: r2l_adam_step. */
int r2l_adam_step_guarded(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                          float beta1, float beta2, float eps, int step, float grad_scale, const unsigned* skip_if,
                          void* stream);
/* The same update (bit for bit) with the re-pack of the fp16x2 weight streams folded in: what r2l_adam_step_guarded followed by
 * r2l_pack_forward_layout(.., 2, ..) and r2l_pack_backward_layout(.., 2, ..) leave behind — the optimizer kernel writes the body
 * weights' (hi, mid) stage pieces of both streams itself and commits the activation scale (range control), a second small kernel
 * packs the head / bias stages for it — two launches instead of four, 42 us of kernel time instead of 65.  (The trainer of this repo
 * keeps the separate packs by default: packed a step early, the backward stream is cold when the dX chain reads it, which costs
 * what the fusion saves; profiles/r05_small_step_ab.txt.)  wstream_fwd / wstream_bwd: the buffers of
 * r2l_fwd_stream_floats / r2l_bwd_stream_floats; only their fp16x2 parts are written (the other layouts stay stale until packed).
 * *skip_if != 0: nothing is touched.  The default trio's step (r2l_forward_layout_for_cfg == r2l_backward_layout_for_cfg == 2). */
int r2l_adam_step_packed(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n_block, float lr,
                         float beta1, float beta2, float eps, int step, float grad_scale, const unsigned* skip_if,
                         float* wstream_fwd, float* wstream_bwd, void* stream);

/* out2[0] = inv_denom * sum(sqerr_partial) (= img2mse * lw_rgb, helpers:19), out2[1] = psnr (helpers:20). */
int r2l_loss_finish(const float* sqerr_partial, int64_t n_partial, float inv_denom, float* out2, void* stream);

/* ---- NeRF teacher (pseudo-data generation) --------------------------------------------------------------------------
 * tparams: flat fp32 state_dict-order buffer of NeRF(D=8, W=256, input_ch=63, input_ch_views=27, use_viewdirs=True)
 * (model/nerf_raybased.py:357-375, built at utils/create_data.py:251-265): pts_linears.{0..7}, views_linears.0,
 * feature_linear, alpha_linear, rgb_linear. */
int64_t r2l_teacher_param_count(void);     /* 595 844 */
int64_t r2l_teacher_stream_floats(void);
int r2l_pack_teacher(const float* tparams, float* wstream, void* stream);
/* Range control of the fp16 teacher kernel: same scheme and word layout as r2l_forward_status_words (zero-fill `wstream`
 * once after allocating it). */
const unsigned* r2l_teacher_status_words(const float* wstream);

/* raw[R,S,4] = network_query_fn(pts = o + d*z, viewdirs, NeRF): run_network (create_data.py:55-77: embed xyz L=10 and
 * dirs L=4, helpers:24-74; the netchunk loop disappears) + NeRF.forward (model/nerf_raybased.py:377-401), fused. */
int r2l_teacher_mlp(const float* rays_o, const float* rays_d, const float* viewdirs, const float* z,
                    const float* wstream, const float* tparams, float* raw, int64_t R, int S, void* stream);
/* ... with explicit dispatch: only cfg->precision matters (cfg == NULL: the plain form). */
int r2l_teacher_mlp_cfg(const float* rays_o, const float* rays_d, const float* viewdirs, const float* z,
                        const float* wstream, const float* tparams, float* raw, int64_t R, int S, void* stream,
                        const r2l_config* cfg);

/* z_out[R,S] = near*(1-t)+far*t, with stratified jitter when t_rand[R,S] != NULL (create_data.py:457-482).
 * near/far: per-ray values read at near[r*nf_stride], far[r*nf_stride]; ttab[2S] = t_vals ++ (1 - t_vals). */
int r2l_stratified_z(const float* near, const float* far, int nf_stride, const float* ttab, const float* t_rand,
                     float* z_out, int64_t R, int S, void* stream);

/* raw2outputs (create_data.py:335-402 == main.py:556-621 == model/nerf_raybased.py:226-295).  noise[R,S] (already
 * scaled by raw_noise_std) and weights[R,S] are optional (NULL).  1 <= S <= 256. */
int r2l_raw2outputs(const float* raw, const float* z, const float* rays_d, const float* noise, int white_bkgd,
                    float* rgb_map, float* disp_map, float* acc_map, float* weights, float* depth_map, int64_t R, int S,
                    void* stream);

/* Hierarchical sampling on the GPU (the reference round-trips through the CPU, create_data.py:505-515):
 *   z_samples[R,NI] = sample_pdf(.5*(z[1:]+z[:-1]), weights[:,1:-1], NI, u)     (helpers:283-330)
 *   z_all[R,S+NI]   = sort(cat[z, z_samples])  ;  z_std[R] = std(z_samples, unbiased=False)  (optional)
 * u: the uniforms, read at u[r*u_stride + i] (u_stride = 0: one shared row, e.g. det=True's linspace(0,1,NI)). */
int r2l_sample_pdf_sort(const float* z, const float* weights, const float* u, int64_t u_stride, float* z_samples,
                        float* z_all, float* z_std, int64_t R, int S, int NI, void* stream);

/* ---- test-set metric ------------------------------------------------------------------------------------------------
 * out[0] = SSIM(img1, img2): utils/ssim_torch.py:28-56,86-94 as called at main.py:46,254,334 (11x11 Gaussian sigma 1.5,
 * zero padding, C1 = 0.01^2, C2 = 0.03^2, mean over every pixel and channel), fused into one kernel + a fixed-order
 * finish.  img1/img2: device [H, W, C] fp32 (the layout render_path holds, no permute).  window_host: host pointer to
 * the 121 window values (ssim_torch.py:19-25) or NULL to have them computed here.  partial: device scratch of
 * r2l_ssim_partial_count(H, W, C) floats. */
int64_t r2l_ssim_partial_count(int H, int W, int C);
int r2l_ssim(const float* img1, const float* img2, int H, int W, int C, const float* window_host, float* partial,
             float* out, void* stream);

/* ---- hard-ray pool (training data path) ---------------------------------------------------------------------------------
 * The three data movements of main.py:1325-1347 (n_hard_out random pool rows [o, d, rgb] appended to every batch) and
 * main.py:1410-1425 (the hard rays of the step enter the pool, appended until it is full, then replacing the rows that were
 * handed out), one kernel each; ranking the per-ray errors stays a sort on the host's side of the ABI.
 *   r2l_pool_pick: idx_out[i], i < n_out = n_out DISTINCT rows of [0, n_rows), every row equally likely: a keyed bijection
 *     (4-round Feistel, cycle-walked) evaluated at 0 .. n_out-1 — replaces np.random.permutation(n_rows)[:n_out]; same key,
 *     same rows.
 *   r2l_pool_augment: out_{o,d,t}[B + n_out, 3] (contiguous) = the batch's rows (inputs may be column slices of a [B, 9] shard
 *     batch: row strides in floats) followed by pool rows idx[0 .. n_out).
 *   r2l_pool_store: pool[dst(i)] = [o, d, t][hard[i]], i < n_in, dst(i) = dst_idx[i] (replace) or dst0 + i (dst_idx NULL: append).
 * pool: [rows, 9] fp32 row-major.  idx / hard / dst_idx: device int64. */
int r2l_pool_pick(int64_t* idx_out, int64_t n_out, int64_t n_rows, uint64_t key, void* stream);
int r2l_pool_augment(const float* rays_o, const float* rays_d, const float* target, int64_t stride_o, int64_t stride_d,
                     int64_t stride_t, const float* pool, const int64_t* idx, int64_t B, int64_t n_out, float* out_o, float* out_d,
                     float* out_t, void* stream);
int r2l_pool_store(const float* rays_o, const float* rays_d, const float* target, int64_t stride_o, int64_t stride_d,
                   int64_t stride_t, const int64_t* hard, float* pool, const int64_t* dst_idx, int64_t dst0, int64_t n_in,
                   void* stream);

/* ---- frame writer (host threads; test-set evaluation) ------------------------------------------------------------------
 * Replaces `imageio.imwrite(filename, to8b(rgb))` of every prediction / ground-truth frame in render_path (main.py:337-344):
 * a pool of encoder threads (zlib, Sub filter; lossless, so the decoded pixels are the bytes handed over).  `pixels`: HOST
 * buffer of H*W*C bytes (C = 1, 3 or 4; row-major), owned by the caller and left untouched until the job is done;
 * `ready_event`: NULL, or a hipEvent_t the worker waits for before reading `pixels` (the frame's asynchronous device-to-host
 * copy).  r2l_png_writer_wait(job): job and all earlier ones are on disk (job < 0: everything submitted); non-zero if any
 * job failed (r2l_last_error).  level: zlib 0..9 (1: ~2 ms per 400x400 frame). */
typedef struct r2l_png_writer r2l_png_writer;
int r2l_png_writer_open(int n_threads, int level, r2l_png_writer** out);
int r2l_png_writer_submit(r2l_png_writer* w, const char* path, const unsigned char* pixels, int H, int W, int C,
                          void* ready_event, int64_t* job_id);
int r2l_png_writer_wait(r2l_png_writer* w, int64_t job_id);
int r2l_png_writer_close(r2l_png_writer* w);

/* ---- ray-shard reader (host threads; --data_mode rays) ------------------------------------------------------------
 * Replaces BlenderDataset_v2.__getitem__ (dataset/load_blender.py:257-324: np.load of one [4096,9] f32 shard),
 * InfiniteSamplerWrapper (main.py:759-776: random permutations of the file list, forever) and the DataLoader's
 * batch_size=N_rand collate + pin_memory (main.py:794-806).  `slot_ptrs[depth]` are caller-owned (pinned) host buffers
 * of files_per_batch * rows * cols * 4 bytes each; reader threads pread() shard payloads straight into them.  All
 * shards must have the shape of paths[0].  Format: NumPy .npy v1/v2/v3, '<f4', C order, 2-D. */
typedef struct r2l_reader r2l_reader;
int r2l_npy_shape(const char* path, int64_t* rows, int64_t* cols);
int r2l_reader_open(const char* const* paths, int64_t n_paths, int files_per_batch, int n_threads, uint64_t seed,
                    void* const* slot_ptrs, int depth, r2l_reader** out);
int r2l_reader_info(r2l_reader* r, int64_t* rows, int64_t* cols, int64_t* files_read);
/* Blocks until the oldest scheduled batch is complete; *slot = index of the buffer holding it.  The buffer is not
 * rewritten until r2l_reader_release(slot), which queues the next batch into it. */
int r2l_reader_next(r2l_reader* r, int* slot);
int r2l_reader_release(r2l_reader* r, int slot);
int r2l_reader_close(r2l_reader* r);

#ifdef __cplusplus
}
#endif
#endif /* R2L_HIP_H */
