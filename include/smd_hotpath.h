/* smd_hotpath.h — C ABI of the MI355X (gfx950) view-synthesis loss hot path.
 *
 * This is the drop-in boundary for the ONE path this repository accelerates: the self-supervised
 * view-synthesis loss of jspenmar/slowtv_monodepth.  The reference has no FFI of its own (it is pure
 * Python on ATen), so each entry point below names the reference Python interface it replaces; the
 * ctypes binding a maintainer would add to the reference is shown in INTEGRATION.md.
 *
 * Conventions
 *   - All pointers are DEVICE pointers to contiguous fp32 (or uint8 where stated) arrays in the
 *     reference's own layouts (NCHW images, row-major 4x4 matrices).  Inputs are const; the library
 *     never allocates, frees or retains a pointer; the caller (PyTorch) owns every buffer.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Nothing synchronises.
 *   - Return value: 0 on success, negative on error (SMD_E_*); smd_last_error() returns the message
 *     for the calling thread.
 *   - Symbols: b batch, n support frames, S scales, (h, w) image size.  Every (S,b,...) tensor is
 *     scale-major, matching the `torch.stack(list(depths.values()))` of src/core/handlers.py:48.
 */
#ifndef SMD_HOTPATH_H
#define SMD_HOTPATH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMD_ABI_VERSION 8

#define SMD_OK 0
#define SMD_E_INVALID (-1)     /* bad argument (null pointer, size out of range, unsupported flag combination) */
#define SMD_E_LAUNCH (-2)      /* hipLaunch / hipGetLastError failure */
#define SMD_E_WORKSPACE (-3)   /* workspace too small: call the matching *_workspace_bytes() */
#define SMD_E_UNSUPPORTED (-4) /* the request exists, but not in this build / not for these arguments (the caller takes the general path) */

/* flags for smd_image_recon_* / smd_recon_reduce_* */
#define SMD_USE_MIN 0x1        /* ReconstructionLoss(use_min=True): min over supports, else mean (reconstruction.py:43-44) */
#define SMD_USE_AUTOMASK 0x2   /* ReconstructionLoss(use_automask=True) (reconstruction.py:59-77) */
#define SMD_LOSS_L1 0x4        /* loss_name='l1' (DenseL1Error) instead of 'ssim' (PhotoError 0.85/0.15) */
#define SMD_NEED_K_GRAD 0x8    /* backward also emits dL/dK and dL/dK_inv (learned intrinsics) */
#define SMD_USE_EDGES 0x10     /* SmoothReg(use_edges=True) (smooth.py:91-94) */
#define SMD_LOSS_L2 0x20       /* DenseL2Error (photometric.py:17-20; loss_name='l2', un-fused operators only) */
#define SMD_MASK_EXPLAINABILITY 0x80   /* smd_recon_reduce_*: ReconstructionLoss(mask_name='explainability'): err * mask (reconstruction.py:55) */
#define SMD_MASK_UNCERTAINTY 0x100     /* smd_recon_reduce_*: mask_name='uncertainty': err * exp(-mask) + mask (reconstruction.py:56) */
#define SMD_BWD_SKIP_DEAD_ROWS 0x400   /* smd_image_recon*_bwd: the liveness-gated row loop — a wave scans the `sel` rows of its strip first and then re-synthesises,
                                        * scores and back-propagates only the rows a pixel of its columns routes gradient through.  Same result bit for bit; all-masked
                                        * input: 52 instead of 117 us at 12x192x640; pays from ~75 % dead (row, strip) units on, costs 12-24 % where every row is live
                                        * (profiles/r04_skip_regimes.txt).  Default: off.  (The knob `bwd_skip`, smd_set_knob, overrides.) */
#define SMD_BWD_NO_LIVE 0x1000         /* smd_image_recon*_bwd, smd_loss_path_bwd: do not consult the forward's liveness table (the caller knows that a launch-shape knob
                                        * changed between the forward that filled it and this call: the table's strip heights would be read with another partition) */
#define SMD_USE_LAPLACIAN 0x200        /* smd_disp_smooth_*: SmoothReg(use_laplacian=True): second-order differences (smooth.py:33-48) */
#define SMD_EDGES_READY 0x800  /* smd_disp_smooth_fwd: `edge_weights` was already filled by smd_disp_smooth_prep() for this frame and pyramid */
#define SMD_PACKED_READY 0x40  /* smd_image_recon_*_fwd: `supp_packed` was already filled by smd_image_recon_prep() for these frames */
/* RegressionLoss (src/losses/regression.py:40-75) */
#define SMD_REGR_L1 0x0
#define SMD_REGR_LOG_L1 0x1
#define SMD_REGR_BERHU 0x2
#define SMD_REGR_INVERT 0x4    /* RegressionLoss(invert=True): both inputs through to_inv first (:70) */

#define SMD_MAX_SCALES 8
#define SMD_MAX_SUPPORTS 8
#define SMD_SEL_MASKED 255     /* value of `sel` where automasking removed the pixel */

const char* smd_last_error(void);
int smd_abi_version(void);

/* Launch-shape knobs.  Process-wide overrides of the built-in heuristics that choose between partitions / code paths which MUST give the
 * same results: rows per strip, the tapered partition, shared LDS ring vs. per-wave loads in the forward, the backward's two row loops,
 * guest work vs. launches of their own.  They exist so that the parity tests can pin both sides of each such choice and compare; the
 * library never reads the environment.  Names: fwd_rh bwd_rh fwd_taper_b bwd_taper_b fwd_taper_rh bwd_taper_rh fwd_ni fwd_share bwd_skip
 * bwd_wps bwd_guest_finalize bwd_direct_level loss_path_guests bwd_live bwd_scales_block (and, in -DSMD_EXPERIMENTS builds only: fwd_ahead bwd_pair smooth_chain).
 * smd_set_knob: 0, SMD_E_INVALID for an unknown name, SMD_E_UNSUPPORTED for an experiments-only knob in the product build. */
int smd_set_knob(const char* name, int value);
void smd_reset_knobs(void);

/* ------------------------------------------------------------------------------------------------
 * K0 — upsample + disparity->depth.  Replaces, per scale, `ops.interpolate_like(disp, imgs, 'bilinear')`
 * followed by `to_scaled(.., min, max)[1]` or `to_inv` (src/core/trainer.py:316-321, src/tools/ops.py:311-314,
 * src/tools/geometry.py:62-90).
 *   disp[s]        : (b,1,hs[s],ws[s])  sigmoid disparity of scale s (host array of S device pointers)
 *   depth_up       : (S,b,h,w) out     depth
 *   disp_up        : (S,b,h,w) out or NULL — the upsampled (un-scaled) disparity (`fwd['disp_up']`)
 *   min_depth/max_depth <= 0 mean "not set" (both unset -> to_inv of the raw disparity).
 * Backward: (depth_up, g_depth_up) (S,b,h,w) -> g_disp[s] (b,1,hs,ws), overwritten.  Needs only the forward's
 * OUTPUT (d depth/d d = -depth^2 on the pass-through branch), not the low-resolution disparity. */
int smd_disp_to_depth_fwd(const float* const* disp, const int* hs, const int* ws, int S, int b, int h, int w,
                          float min_depth, float max_depth, float* depth_up, float* disp_up, void* stream);
size_t smd_disp_to_depth_workspace_bytes(const int* hs, const int* ws, int S, int b, int h, int w);
int smd_disp_to_depth_bwd(const int* hs, const int* ws, int S, int b, int h, int w, float min_depth, float max_depth,
                          const float* depth_up, const float* g_depth_up, float* const* g_disp,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused image reconstruction.  Replaces `handlers.image_recon(crit, synth, depths, None, imgs, supp_imgs, Ts, Ks)`
 * (src/core/handlers.py:14-67) = ViewSynth.forward (src/tools/geometry.py:366-391) + ReconstructionLoss.forward
 * (src/losses/reconstruction.py:98-126) + PhotoError (src/losses/photometric.py:54-88), without materialising
 * the (n,S*b,...) expansions, point clouds, grids or warped images.
 *   depth   (S,b,h,w)    tgt (b,3,h,w)    supp (n,b,3,h,w)    T (n,b,4,4)    K, K_inv (b,4,4)
 *   noise   (S,b,h,w) or NULL: the `randn_like` draw of reconstruction.py:72; NULL -> counter-based
 *           in-kernel Gaussian keyed by `seed` (statistically equivalent tie-break, different stream)
 *   supp_packed out, smd_packed_supports_bytes() bytes (< 4 GB): what every scale and support shares, produced once per
 *           sample by the forward's prep kernel and kept by the caller for the backward — the supports as padded 12-byte
 *           RGB texels (n,b,h+1,w+1,3), the target as RGB texels (b,h,w,3), and per target pixel the SSIM window sums of the
 *           target and the identity error of the automask (2 x (b,h,w,4)); then a small tail of launch scratch (the K0
 *           row table and the arrival counters of the in-launch reductions; the backward resets its counters, which is
 *           why it takes the buffer non-const).  It depends on the frames only: smd_image_recon_prep() can fill it ahead of
 *           time (e.g. on a side stream while the networks run) and the forward is then called with SMD_PACKED_READY.
 *   err     (S,b,h,w) out or NULL (allowed when n <= 4): per-pixel error after min/mean-reprojection and automasking; the
 *           training path does not need it (the loss is reduced in the kernel) and saves its 4 bytes per pixel and scale
 *   sel     (S,b,h,w) out uint8: winning support index, or SMD_SEL_MASKED where the static error won
 *   loss    (1) out: mean of err  (= `loss_img_recon`)
 *   warp0   (n,b,3,h,w) out or NULL: warped supports of scale 0 (`loss_dict['supp_imgs_warp']`)
 *   workspace: smd_image_recon_workspace_bytes() bytes of scratch (fwd and bwd may share it).
 * Backward (loss is the only differentiable output):
 *   g_loss  (1) device scalar dL/dloss
 *   g_depth (S,b,h,w) out;  g_T (n,b,4,4) out (row 3 zero);  g_K, g_Kinv (b,4,4) out or NULL
 *           (required iff SMD_NEED_K_GRAD; only the 3x3 / 2x3 blocks the path reads are non-zero). */
size_t smd_image_recon_workspace_bytes(int b, int n, int S, int h, int w);
size_t smd_packed_supports_bytes(int b, int n, int h, int w);
/* The frame-only half of the forward (texel repack, target window sums, identity error; reference: the un-warped `source`
 * branch of src/losses/reconstruction.py:70-71 and the per-scale re-reads of handlers.py:45-56).  `flags` must carry the same
 * SMD_USE_MIN / SMD_USE_AUTOMASK / SMD_LOSS_L1 bits as the forward that follows.  hs, ws (S entries) describe the disparity
 * pyramid of the K0-fused forward (its row table is built here); NULL / S = 0 for the depth-input forward. */
int smd_image_recon_prep(const float* tgt, const float* supp, float* supp_packed, const int* hs, const int* ws, int S,
                         int b, int n, int h, int w, int flags, void* stream);
/* Supports one forward launch holds in registers (4 unless SMD_FWD_NI overrides it): `err` may be NULL only when n <= this. */
int smd_image_recon_supports_per_pass(void);
int smd_image_recon_fwd(const float* depth, const float* tgt, const float* supp, const float* T, const float* K,
                        const float* K_inv, const float* noise, uint64_t seed, float* supp_packed, float* err, uint8_t* sel, float* loss,
                        float* warp0, void* workspace, size_t workspace_bytes,
                        int b, int n, int S, int h, int w, int flags, void* stream);
int smd_image_recon_bwd(const float* depth, const float* tgt, float* supp_packed, const float* T, const float* K,
                        const float* K_inv, const uint8_t* sel, const float* g_loss,
                        float* g_depth, float* g_T, float* g_K, float* g_Kinv, void* workspace, size_t workspace_bytes,
                        int b, int n, int S, int h, int w, int flags, void* stream);

/* K0 fused into the reconstruction: the same operator fed with the network's multi-scale sigmoid
 * disparity instead of the up-sampled depth.  Replaces, in one launch sequence, `forward_postprocess`' per-scale
 * `ops.interpolate_like` + `to_scaled` / `to_inv` (src/core/trainer.py:316-321) AND `handlers.image_recon`
 * (src/core/handlers.py:14-67): the fused kernel up-samples the disparity of the rows it is about to warp, converts it to
 * depth and writes `depth_up` (S,b,h,w) once (the backward and `fwd['depth_up']` read it); no K0 launch is needed.
 *   disp[s] (b,1,hs[s],ws[s]) host array of S device pointers;  min_depth / max_depth <= 0 mean "not set" (to_inv).
 * Backward: recomputes nothing of K0 — it reads depth_up, adds `g_depth_up_in` (S,b,h,w; the gradient reaching depth_up from
 * other consumers, or NULL), applies d depth / d disparity inside the fused backward and sends the result through the
 * bilinear adjoint -> g_disp[s] (b,1,hs,ws), overwritten.  Workspace: smd_image_recon_disp_workspace_bytes(). */
size_t smd_image_recon_disp_workspace_bytes(const int* hs, const int* ws, int S, int b, int n, int h, int w);
int smd_image_recon_disp_fwd(const float* const* disp, const int* hs, const int* ws, int S, float min_depth, float max_depth,
                             const float* tgt, const float* supp, const float* T, const float* K, const float* K_inv,
                             const float* noise, uint64_t seed, float* supp_packed, float* depth_up, float* err, uint8_t* sel, float* loss,
                             float* warp0, void* workspace, size_t workspace_bytes, int b, int n, int h, int w, int flags, void* stream);
int smd_image_recon_disp_bwd(const int* hs, const int* ws, int S, float min_depth, float max_depth, const float* depth_up,
                             float* supp_packed, const float* T, const float* K, const float* K_inv, const uint8_t* sel,
                             const float* g_loss, const float* g_depth_up_in, float* const* g_disp, float* g_T, float* g_K, float* g_Kinv,
                             void* workspace, size_t workspace_bytes, int b, int n, int h, int w, int flags, void* stream);

/* The whole `forward_loss` of the kbr configuration as ONE operator (round 5): `handlers.image_recon` on the K0-fused path +
 * `handlers.disp_smooth` with `SmoothReg(use_edges=True)` + the weighted sum `sum_k w_k l_k` (src/core/trainer.py:383-392, 436-437,
 * 462-464) — and, in the backward, the chain rule through `T_from_AAt` (+ `T.inverse()`) and `build_K` / `resize_K` down to the pose
 * network's outputs (src/core/trainer.py:250-262).  Five launches for forward + backward where the separate operators need eleven:
 *   forward:  the K0-fused reconstruction launch, with the smoothness sweep as guest blocks behind its own and the weighted sum formed
 *             in-launch by whichever of the two final reducers arrives second -> loss3 = {total, l_recon, l_smooth};
 *   backward: the fused reconstruction backward; the K0 adjoint's first launch, carrying as guests the pose epilogue — continued in the
 *             same wave to g_aa, g_t (and g_fs, g_cs) — and the smoothness adjoint, which WRITES its share into g_disp[s] (adds, for a
 *             level of the image's own size that the reconstruction backward already wrote); the K0 adjoint's second launch ADDS.
 * flags: SMD_USE_MIN | SMD_USE_AUTOMASK | SMD_USE_EDGES (required) | SMD_PACKED_READY | SMD_EDGES_READY | SMD_NEED_K_GRAD |
 *        SMD_BWD_SKIP_DEAD_ROWS.  In-kernel tie-break noise (`seed`), no error map, no warped images: the trainer's hot call.
 * Returns SMD_E_UNSUPPORTED (nothing launched) for what it does not serve — loss_name 'l1', more than four supports, a smoothness term
 * other than first-order edge-aware, fewer than two pyramid levels, a level taller than the image: the caller then uses the operators above.
 *   scale_keys[s]: the `s` of `loss_s / 2**s` (src/core/handlers.py:279);  stats (S,b,2): per-image (mean, E) kept for the backward;
 *   edge_weights: smd_disp_smooth_edge_weight_bytes() (filled here unless SMD_EDGES_READY);  w_recon / w_smooth: the frozen loss weights;
 *   g_loss: device scalar dL/d total.  aa, t, invert (n*b rows, as given to smd_pose_fwd) with g_aa, g_t, and fs, cs with g_fs, g_cs
 *   (needs SMD_NEED_K_GRAD), are optional (all NULL: stop at g_T / g_K / g_Kinv, which are always written).
 * Same values as the separate operators: l_recon, l_smooth, sel, depth_up and every gradient bit for bit (tests/test_gpu_parity.py). */
size_t smd_loss_path_workspace_bytes(const int* hs, const int* ws, int S, int b, int n, int h, int w);
int smd_loss_path_fwd(const float* const* disp, const int* hs, const int* ws, const int* scale_keys, int S, float min_depth, float max_depth,
                      const float* tgt, const float* supp, const float* T, const float* K, const float* K_inv, uint64_t seed,
                      float* supp_packed, float* edge_weights, float* depth_up, uint8_t* sel, float* loss3, float* stats,
                      void* workspace, size_t workspace_bytes, int b, int n, int h, int w, int flags, float w_recon, float w_smooth, void* stream);
int smd_loss_path_bwd(const float* const* disp, const int* hs, const int* ws, const int* scale_keys, int S, float min_depth, float max_depth,
                      const float* depth_up, float* supp_packed, const float* T, const float* K, const float* K_inv, const uint8_t* sel,
                      const float* stats, const float* edge_weights, const float* g_loss, float w_recon, float w_smooth,
                      const float* aa, const float* t, const uint8_t* invert, const float* fs, const float* cs,
                      float* const* g_disp, float* g_T, float* g_K, float* g_Kinv, float* g_aa, float* g_t, float* g_fs, float* g_cs,
                      void* workspace, size_t workspace_bytes, int b, int n, int h, int w, int flags, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Edge-aware disparity smoothness over all scales.  Replaces `handlers.disp_smooth(crit, disps, imgs)`
 * (src/core/handlers.py:262-281) = per scale SmoothReg.forward (src/regularizers/smooth.py:71-97) on the
 * bilinearly resized image, then mean_s(loss_s / 2^s).  SMD_USE_LAPLACIAN selects the second-order form (smooth.py:33-48);
 * use_blur: the host side blurs the disparity and the resized image with smd_gaussian_blur3x3 and calls this path on the results (first-order form).
 *   disp[s] (b,1,hs,ws)   img (b,3,h,w)   scale_keys[s]: the dictionary key of scale s (loss_s is divided by 2^key;
 *   NULL -> key = s)
 *   loss (1) out;  stats (S,b,2) out: per (scale, sample) {mean disparity, un-normalised edge sum E} kept for backward
 *   disp_grad, image_grad: (b,1,hs[0],ws[0]) out or NULL (aux maps of the first scale, smooth.py:86,89)
 *   edge_weights: required with SMD_USE_EDGES, else NULL; smd_disp_smooth_edge_weight_bytes() bytes holding
 *   {exp(-mean_c |dI/dx|), exp(-mean_c |dI/dy|)} per pixel of every scale (and a few launch counters behind them).  They depend on the frames
 *   alone: smd_disp_smooth_prep() fills the buffer ahead of time (e.g. on a side stream while the networks run, next to
 *   smd_image_recon_prep) and the forward is then called with SMD_EDGES_READY; without that flag the forward fills it first itself.
 *   Handing the same buffer to the backward spares it every image access.
 * Backward: g_disp[s] (b,1,hs,ws) out. */
/* SmoothReg(use_blur=True) (smooth.py:21): `kornia.filters.gaussian_blur2d(x, kernel_size=(3, 3), sigma=(1, 1))` on `planes` planes of h x w
 * floats — separable 3-tap Gaussian (0.27406862, 0.45186276, 0.27406862), reflect border (kornia 0.6.10 filter2d_separable; restated from its
 * published source, parity unpinned: the library is absent from the build image).  adjoint != 0: the transpose of that linear map (backward).
 * x != out; h, w >= 2. */
int smd_gaussian_blur3x3(const float* x, float* out, int planes, int h, int w, int adjoint, void* stream);
size_t smd_disp_smooth_workspace_bytes(const int* hs, const int* ws, int S, int b);
size_t smd_disp_smooth_edge_weight_bytes(const int* hs, const int* ws, int S, int b);
int smd_disp_smooth_prep(const float* img, const int* hs, const int* ws, int S, int b, int h, int w, int flags, float* edge_weights, void* stream);
int smd_disp_smooth_fwd(const float* const* disp, const int* hs, const int* ws, const int* scale_keys, int S, int b,
                        const float* img, int h, int w, int flags, float* loss, float* stats, float* disp_grad, float* image_grad,
                        float* edge_weights, void* workspace, size_t workspace_bytes, void* stream);
int smd_disp_smooth_bwd(const float* const* disp, const int* hs, const int* ws, const int* scale_keys, int S, int b,
                        const float* img, int h, int w, int flags, const float* stats, const float* edge_weights,
                        const float* g_loss, float* const* g_disp, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Un-fused operators (class-level drop-ins for callers that hold the intermediate tensors).
 *
 * smd_view_synth_*: `ViewSynth.forward(input, depth, T, K, K_inv)` (src/tools/geometry.py:366-391) for any
 * channel count C.  B is the already-expanded batch.  warp (B,C,h,w); depth_warp (B,1,h,w) or NULL;
 * mask_valid (B,1,h,w) uint8 or NULL.  Backward takes g_warp (and optional g_depth_warp) and emits
 * g_input (B,C,h,w) or NULL, g_depth (B,1,h,w), g_T (B,4,4), g_K / g_Kinv (B,4,4) or NULL. */
size_t smd_view_synth_workspace_bytes(int B, int h, int w);
int smd_view_synth_fwd(const float* input, const float* depth, const float* T, const float* K, const float* K_inv,
                       float* warp, float* depth_warp, uint8_t* mask_valid, int B, int C, int h, int w, void* stream);
int smd_view_synth_bwd(const float* input, const float* depth, const float* T, const float* K, const float* K_inv,
                       const float* g_warp, const float* g_depth_warp,
                       float* g_input, float* g_depth, float* g_T, float* g_K, float* g_Kinv,
                       void* workspace, size_t workspace_bytes, int B, int C, int h, int w, void* stream);

/* smd_photo_error_*: `PhotoError(weight_ssim)(pred, target)`, `DenseL1Error` with SMD_LOSS_L1, `DenseL2Error` with SMD_LOSS_L2
 * (src/losses/photometric.py:11-20, 54-88) for any channel count C (images: 3; `feat_recon` features: 64..256).
 * weight_ssim in [0, 1] (photometric.py:65-73; `ReconstructionLoss` builds 0.85; 0 = L1 only, 1 = SSIM only); ignored by L1 / L2.
 * pred, target (N,C,h,w) -> err (N,1,h,w).  Backward: g_err (N,1,h,w) -> g_pred (N,C,h,w). */
size_t smd_photo_error_workspace_bytes(int N, int C, int h, int w);
int smd_photo_error_fwd(const float* pred, const float* target, float* err, int N, int C, int h, int w, int flags, float weight_ssim, void* stream);
int smd_photo_error_bwd(const float* pred, const float* target, const float* g_err, float* g_pred,
                        void* workspace, size_t workspace_bytes, int N, int C, int h, int w, int flags, float weight_ssim, void* stream);

/* smd_regression_*: `RegressionLoss.forward(pred, target, mask)` (src/losses/regression.py:69-75) used by the
 * `stereo_const` and `depth_regr` handlers (src/core/handlers.py:152-259).  pred, target: N floats; mask: N uint8 or
 * NULL (all ones); flags: SMD_REGR_{L1,LOG_L1,BERHU} | SMD_REGR_INVERT.  -> loss (1), err (N, `err_regr`) or NULL,
 * stats (8 floats, handed back to the backward).  Backward: g_loss (1) -> g_pred and/or g_target (N). */
size_t smd_regression_workspace_bytes(size_t N);
int smd_regression_fwd(const float* pred, const float* target, const uint8_t* mask, size_t N, int flags, float* loss, float* err,
                       float* stats, void* workspace, size_t workspace_bytes, void* stream);
int smd_regression_bwd(const float* pred, const float* target, const uint8_t* mask, size_t N, int flags, float* stats,
                       const float* g_loss, float* g_pred, float* g_target, void* workspace, size_t workspace_bytes, void* stream);

/* smd_recon_reduce_*: the reduction half of `ReconstructionLoss.forward` on per-support error maps
 * (reconstruction.py:43-57, 59-77, 125).  err_warp (n,B,h,w); err_static (n,B,h,w) (required with SMD_USE_AUTOMASK);
 * mask (B,n,h,w) or NULL: the predictive weighting mask of `apply_mask` (reconstruction.py:46-57), reference layout, applied to the
 * warped AND the static errors before the reductions, with SMD_MASK_EXPLAINABILITY (err * mask) or SMD_MASK_UNCERTAINTY
 * (err * exp(-mask) + mask); noise (B,h,w) or NULL (in-kernel tie-break noise keyed by `seed`).
 * -> err (B,h,w), sel (B,h,w) uint8, loss (1).  Backward -> g_err_warp (n,B,h,w) and, with a mask, g_mask (B,n,h,w)
 * (then err_warp / err_static / mask are read again; all three may be NULL without a mask). */
size_t smd_recon_reduce_workspace_bytes(int B, int h, int w);
int smd_recon_reduce_fwd(const float* err_warp, const float* err_static, const float* mask, const float* noise, uint64_t seed,
                         float* err, uint8_t* sel, float* loss, void* workspace, size_t workspace_bytes,
                         int n, int B, int h, int w, int flags, void* stream);
int smd_recon_reduce_bwd(const uint8_t* sel, const float* g_loss, float* g_err_warp, const float* err_warp, const float* err_static,
                         const float* mask, float* g_mask, int n, int B, int h, int w, int flags, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Monodepth decoder glue: everything between two 3x3 convolutions of
 * `MonodepthDecoder.forward` (src/networks/decoders/monodepth.py:71-89; conv_block / conv3x3 with reflection padding,
 * src/networks/decoders/utils.py:44-54) as one gather that writes the next convolution's padded input.
 *
 * smd_elu_pad_*:        out (B,C,h+2,w+2) = reflect_pad1(apply_elu ? elu(x + bias) : x + bias),  x (B,C,h,w) the raw
 *                       (bias-free) conv output, bias (C) or NULL.
 * smd_elu_up_cat_pad_*: out (B,Ca+Cs,2h+2,2w+2) = reflect_pad1(cat(nearest_x2(elu(a + bias)), skip)),  a (B,Ca,h,w),
 *                       bias (Ca) or NULL, skip (B,Cs,2h,2w) or NULL with Cs = 0.
 * Backward: g_out -> g_x / (g_a, g_skip) and g_bias (C) or NULL (needs the workspace); g_a or g_skip may be NULL.
 * dtypes: bit 0 (SMD_GLUE_A_BF16) x / a and their gradients are bfloat16, bit 1 (SMD_GLUE_SKIP_BF16) skip and its gradient,
 * bit 2 (SMD_GLUE_OUT_BF16) out and its gradient — for decoders running under bf16 autocast; bias, g_bias and the arithmetic are fp32. */
#define SMD_GLUE_A_BF16 1
#define SMD_GLUE_SKIP_BF16 2
#define SMD_GLUE_OUT_BF16 4
size_t smd_decoder_glue_workspace_bytes(int B, int C, int h, int w);
int smd_elu_pad_fwd(const void* x, const float* bias, void* out, int B, int C, int h, int w, int apply_elu, int dtypes, void* stream);
int smd_elu_pad_bwd(const void* x, const float* bias, const void* g_out, void* g_x, float* g_bias,
                    void* workspace, size_t workspace_bytes, int B, int C, int h, int w, int apply_elu, int dtypes, void* stream);
int smd_elu_up_cat_pad_fwd(const void* a, const float* bias, const void* skip, void* out, int B, int Ca, int Cs, int h, int w, int dtypes,
                           void* stream);
int smd_elu_up_cat_pad_bwd(const void* a, const float* bias, const void* g_out, void* g_a, void* g_skip, float* g_bias,
                           void* workspace, size_t workspace_bytes, int B, int Ca, int Cs, int h, int w, int dtypes, void* stream);

/* Output head of the Monodepth decoder (ABI 7, round 5; reference: src/networks/decoders/monodepth.py:52, 86-87 — `self.act(self.out[i](x))` with
 * `conv3x3(num_ch_dec[i], 1)`, decoders/utils.py:44-46): y (B,1,h,w) = act(conv3x3(xp; weight (1,C,3,3)) + bias (1) or NULL), where xp (B,C,h+2,w+2)
 * is the reflection-padded activation smd_elu_pad_fwd leaves; act 0: identity, 1: sigmoid.  A one-output-channel convolution is a stencil: three
 * streaming kernels, fp32, deterministic.  Backward: g_y and the saved y -> g_xp (B,C,h+2,w+2) (or NULL), g_weight (1,C,3,3) with g_bias (1) (g_weight
 * NULL: neither; g_bias may be NULL alone); the weight gradient needs the workspace.
 * act | SMD_HEAD_X_BF16 (ABI 8): xp and g_xp are bfloat16 (the decoder under bf16 autocast: `cfg/kbr/default.yaml` trains in bf16-mixed — the glue kernels
 * then leave a bf16 activation; these kernels are HBM-bound, so half the bytes); weights, bias, y, g_y and every sum stay fp32. */
#define SMD_HEAD_X_BF16 2
size_t smd_conv3x3_head_workspace_bytes(int B, int C, int h, int w);
int smd_conv3x3_head_fwd(const void* xp, const float* weight, const float* bias, float* y, int B, int C, int h, int w, int act, void* stream);
int smd_conv3x3_head_bwd(const void* xp, const float* weight, const float* y, const float* g_y, void* g_xp, float* g_weight, float* g_bias,
                         void* workspace, size_t workspace_bytes, int B, int C, int h, int w, int act, void* stream);

/* The decoder's thin up-convolutions (ABI 7, round 5; reference: src/networks/decoders/monodepth.py:45-50, 80-84 — `ConvELU(cin, 16)`; the bias and the ELU
 * are the next glue kernel's): y (B,16,h,w) = conv3x3(xp (B,C,h+2,w+2); weight (16,C,3,3)), bias-free, C = 16 or 32 (anything else: SMD_E_UNSUPPORTED),
 * on the matrix cores in fp32 (v_mfma_f32_16x16x4_f32: exact f32 products, f32 accumulation; the order of the sum differs from ATen's).
 * Backward: g_y (B,16,h,w) -> g_xp (B,C,h+2,w+2) (or NULL) and g_weight (16,C,3,3) (or NULL; needs xp and the workspace); deterministic. */
size_t smd_conv3x3_thin_workspace_bytes(int B, int C, int h, int w);
int smd_conv3x3_thin_fwd(const float* xp, const float* weight, float* y, int B, int C, int h, int w, void* stream);
int smd_conv3x3_thin_bwd(const float* xp, const float* weight, const float* g_y, float* g_xp, float* g_weight, void* workspace, size_t workspace_bytes,
                         int B, int C, int h, int w, void* stream);

/* The decoder's wide up-convolutions (ABI 8, round 6; reference: src/networks/decoders/monodepth.py:40-50, 71-84 — `ConvELU(cin, cout)`, decoders/utils.py:44-54;
 * the bias and the ELU are the next glue kernel's): y (B,CO,h,w) = conv3x3(xp (B,C,h+2,w+2); weight (CO,C,3,3)), bias-free, fp32 in and out, computed on the
 * bf16 matrix cores with every fp32 operand split exactly into `pieces` bf16 pieces (3: six bf16 products per fp32 product, what is dropped is below
 * 2^-25 of the product — fp32-class results at 6/16 of the f32 MFMA's time; 2: three products, 16 significant bits, an experiment setting, never the
 * library's choice).  smd_conv3x3_mfma_pack writes the weights' pieces in the operand order of the forward (wp_fwd) and of the data gradient (wp_bwd), each
 * smd_conv3x3_mfma_packed_bytes(C, CO, pieces) bytes (either may be NULL); the backward reads what the forward's pack left.
 * pieces = 1 (ABI 8): the tensors xp, y, g_y, g_xp are BFLOAT16 (the decoder under bf16 autocast, `cfg/kbr/default.yaml`): one bf16 product per product, the
 * weights enter as their bf16 rounding (what autocast hands a bf16 convolution), accumulation and the weight gradient stay fp32.
 * Served: pieces in {1, 2, 3}; forward C % 16 == 0 and CO % 32 == 0; data gradient CO % 16 == 0 and C % 32 == 0; weight gradient CO % 32 == 0 (any C); and the
 * thin last stage, CO == 16 with C == 16 or 32 (`ConvELU(cin, 16)`, monodepth.py:45-50: all three operators, on the 16 x 16 x 32 form of the instruction);
 * anything else SMD_E_UNSUPPORTED, nothing launched.  g_xp (B,C,h+2,w+2) is the gradient of the PADDED input.  Every call takes a workspace of
 * smd_conv3x3_mfma_workspace_bytes (the coarse decoder levels — few pixels, thousands of K — split K over blocks and add the splits' outputs in split
 * order; the weight gradient leaves per-block sums that a fixed-order fp64 second stage adds — in slices, whose fp64 sums sit behind the per-block sums in the
 * same workspace).  Sizes: one sample's xp and one sample's g_y under 2 GiB each, B x ceil(C / 32) x ceil(CO / 32) < 65536 (else SMD_E_INVALID).  Deterministic. */
size_t smd_conv3x3_mfma_packed_bytes(int C, int CO, int pieces);
size_t smd_conv3x3_mfma_workspace_bytes(int B, int C, int CO, int h, int w);
int smd_conv3x3_mfma_pack(const float* weight, void* wp_fwd, void* wp_bwd, int C, int CO, int pieces, void* stream);
int smd_conv3x3_mfma_fwd(const void* xp, const void* wp_fwd, void* y, void* workspace, size_t workspace_bytes,
                         int B, int C, int CO, int h, int w, int pieces, void* stream);
int smd_conv3x3_mfma_bwd_data(const void* g_y, const void* wp_bwd, void* g_xp, void* workspace, size_t workspace_bytes,
                              int B, int C, int CO, int h, int w, int pieces, void* stream);
int smd_conv3x3_mfma_bwd_weight(const void* xp, const void* g_y, float* g_weight, void* workspace, size_t workspace_bytes,
                                int B, int C, int CO, int h, int w, int pieces, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Producer side of the path: training-mode BatchNorm2d of the ResNet encoders fused with the residual add and ReLU
 * that follow it (timm resnet blocks built at src/networks/depth.py:95-98, src/networks/pose.py:39-41;
 * `F.batch_norm(training=True, momentum, eps)` semantics: biased variance for normalisation, unbiased for the running
 * estimate).  x, y, residual (N,C,H,W) with HW = H*W; gamma, beta, running_*, save_* (C).
 *   y = [relu]( (x - mean)/sqrt(var + eps) * gamma + beta [+ residual] )
 * Backward: g_y -> g_x, g_gamma, g_beta and, when g_residual != NULL, the gradient of the residual branch
 * (g_y masked by y > 0 when relu). */
size_t smd_bn_workspace_bytes(int N, int C, int HW);
int smd_bn_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float* running_mean, float* running_var,
               float momentum, float eps, int relu, float* y, float* save_mean, float* save_invstd,
               void* workspace, size_t workspace_bytes, int N, int C, int HW, void* stream);
int smd_bn_bwd(const float* x, const float* y, const float* g_y, const float* gamma, const float* save_mean, const float* save_invstd,
               int relu, float* g_x, float* g_residual, float* g_gamma, float* g_beta,
               void* workspace, size_t workspace_bytes, int N, int C, int HW, void* stream);

/* smd_maxpool3x3s2_*: `nn.MaxPool2d(3, 2, 1)` of the ResNet stem.  x (N,C,H,W) -> y (N,C,Ho,Wo), Ho = (H-1)/2+1, and
 * idx (N,C,Ho,Wo) uint8 = winning window position dh*3+dw (first maximum in scan order, as ATen).  Backward gathers. */
int smd_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int N, int C, int H, int W, void* stream);
int smd_maxpool3x3s2_bwd(const float* g_y, const uint8_t* idx, float* g_x, int N, int C, int H, int W, void* stream);

/* smd_dwconv7x7_*: the depthwise 7x7 convolution (stride 1, padding 3, groups = C) of the ConvNeXt blocks
 * (timm convnext_* encoders of BASELINE configs 2-4, built at src/networks/depth.py:95-98).
 * x, y (N,C,H,W); weight (C,1,7,7); bias (C) or NULL.  flip = 0: forward.  flip = 1: data gradient (call with x = g_y,
 * bias = NULL; weights are read mirrored).  wrw: g_weight (C,1,7,7) and g_bias (C) or NULL from x and g_y. */
size_t smd_dwconv7x7_workspace_bytes(int C, int H, int W);
int smd_dwconv7x7_fwd(const float* x, const float* weight, const float* bias, float* y, int N, int C, int H, int W, int flip, void* stream);
int smd_dwconv7x7_wrw(const float* x, const float* g_y, float* g_weight, float* g_bias, void* workspace, size_t workspace_bytes,
                      int N, int C, int H, int W, void* stream);

/* smd_layernorm_cf_*: LayerNorm over the channel dimension of an NCHW tensor, i.e. timm's `LayerNorm2d` and the block norm
 * of ConvNeXt evaluated without the NCHW <-> NHWC permutes (`F.layer_norm(x.permute(0,2,3,1), (C,), gamma, beta, eps)`).
 * x, y (N,C,H,W) with HW = H*W; gamma, beta (C); mean, rstd (N*HW) kept for the backward.  y may be written as bfloat16
 * (y_is_bf16 != 0) and g_y read as bfloat16 (g_y_is_bf16 != 0) when the consumer is a bf16 convolution under autocast;
 * the arithmetic is fp32 either way.  Backward: g_y -> g_x, g_gamma, g_beta (fp32). */
size_t smd_layernorm_cf_workspace_bytes(int N, int C, int HW);
int smd_layernorm_cf_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_is_bf16, float* mean, float* rstd,
                         int N, int C, int HW, float eps, void* stream);
int smd_layernorm_cf_bwd(const float* x, const void* g_y, int g_y_is_bf16, const float* gamma, const float* mean, const float* rstd,
                         float* g_x, float* g_gamma, float* g_beta, void* workspace, size_t workspace_bytes, int N, int C, int HW, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pose / intrinsics prologue — one launch each instead of ~45 eager ATen launches.
 *
 * smd_pose_*: `T_from_AAt(aa, t)` (src/tools/geometry.py:181-209), followed by `T.inverse()` where invert[i] != 0
 * (src/core/trainer.py:253: supports behind the target when `always_fwd_pose`).  aa, t (N,3) -> T (N,4,4).
 * Backward: g_T (N,4,4) -> g_aa, g_t (N,3).
 *
 * smd_intrinsics_*: with fs != NULL: `resize_K(PoseNet.build_K(fs, cs), (h, w))` (src/networks/pose.py:60-73,
 * src/tools/geometry.py:249-263) -> K (b,4,4) and its inverse K_inv (the `K.inverse()` of geometry.py:383).
 * With fs == NULL: K_inv of the caller's K_in (b,4,4) (3x3 block, rest identity).  Backward (learned K only):
 * g_K, g_Kinv (b,4,4) -> g_fs, g_cs (b,2). */
int smd_pose_fwd(const float* aa, const float* t, const uint8_t* invert, int N, float* T, void* stream);
int smd_pose_bwd(const float* aa, const float* t, const uint8_t* invert, int N, const float* g_T, float* g_aa, float* g_t, void* stream);
int smd_intrinsics_fwd(const float* fs, const float* cs, const float* K_in, int b, int h, int w, float* K, float* K_inv, void* stream);
int smd_intrinsics_bwd(const float* fs, const float* cs, int b, int h, int w, const float* g_K, const float* g_Kinv,
                       float* g_fs, float* g_cs, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GPU-side aspect-ratio augmentation.  Replaces `crop_aug` + `resize_aug` of
 * src/core/aspect_ratio.py:67-151 for the image tensors of a batch and `centre_crop_K` + `resize_K`
 * (src/tools/geometry.py:233-263) for its intrinsics, in one launch: every plane of every segment (a tensor viewed as
 * (planes[k], H, W): x.imgs, y.imgs, x.supp_imgs, y.supp_imgs, depth ...) is centre-cropped to (crop_h, crop_w) as kornia's
 * `center_crop(size, mode='bilinear', align_corners=False)` does it (aspect_ratio.py:78) — the integer window starting at
 * int(H/2 - crop_h/2), bilinearly RE-SAMPLED at x(i) = ((i + 0.5)(w - 1)/w + x0) W/(W - 1) - 0.5 with zero padding: kornia's
 * (n - 1)-normalised warp under align_corners=False grids — and the crop resized to (out_h, out_w) with
 * `F.interpolate(mode='bilinear', align_corners=False)` semantics; dst[k] is (planes[k], out_h, out_w).  crop == input size:
 * resize only, no resampling of the frame (the augmentation's not-applied branch); out == crop size: crop only.  H, W >= 2.  K_in / K_out (nK,4,4) or both NULL.  The shapes are sampled on the host
 * (`slowtv_monodepth_amd.aspect_ratio`, same random streams as the reference). */
int smd_crop_resize(const float* const* src, float* const* dst, const int* planes, int nseg, int H, int W, int crop_h, int crop_w,
                    int out_h, int out_w, const float* K_in, float* K_out, int nK, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Kernel timing hooks for bench.py: while enabled, every call of the named entry point records a HIP event pair on the
 * caller's stream, either around its DOMINANT kernel (the fused strip kernel k_recon_main / k_recon_bwd) or around ALL
 * of its launches (forward: per-sample prep, unless SMD_PACKED_READY, + main; backward: the fused adjoint [+ the K0 adjoint is
 * outside]; the scalar / pose reductions run inside those launches since ABI 4), or around the prep launches wherever they run.
 * smd_profile_collect() waits for the recorded events and returns their durations in ms.  Not thread-safe; one device. */
#define SMD_PROF_RECON_FWD 0       /* smd_image_recon_fwd, dominant kernel */
#define SMD_PROF_RECON_BWD 1       /* smd_image_recon_bwd, dominant kernel */
#define SMD_PROF_RECON_FWD_ALL 2   /* smd_image_recon_fwd, every launch */
#define SMD_PROF_RECON_BWD_ALL 3   /* smd_image_recon_bwd, every launch */
#define SMD_PROF_RECON_PREP 4      /* k_recon_prep launches (inside the forward entry point or smd_image_recon_prep) */
/* The template instantiation the process's last fused forward (which = SMD_PROF_RECON_FWD) / backward (SMD_PROF_RECON_BWD) launch
 * picked, spelled as rocprofv3 prints kernel names ("smd::k_recon_main<2, true, true, false, true, 1, true>"); "" before the first launch. */
const char* smd_last_kernel_variant(int which);
int smd_profile_enable(int which, int capacity);
int smd_profile_collect(int which, float* ms_out, int max_out, int* n_out);

/* Measurement aid: STREAM-style sweep over nbytes (multiple of 16): mode 0 copies src -> dst (read + write), mode 1 only
 * reads src; add 2 (modes 2, 3) for the variant with eight instead of four 16-byte loads in flight per lane and plain instead of
 * non-temporal accesses, add 4 (modes 4, 5) for the variant in which every block walks one contiguous chunk 32 KB at a time
 * (non-temporal).  bench.py times all of them and quotes the best as the measured HBM ceiling of the box beside the
 * datasheet peak. */
int smd_debug_stream_copy(const void* src, void* dst, size_t nbytes, int mode, void* stream);

/* Debug/self-test: out[l] = {value held by lane l-1, value held by lane l+1} for in[l] = l (64 lanes).
 * Used by the GPU tests to pin the cross-lane primitive the stencil kernels rely on. */
int smd_debug_lane_shift(float* out_left, float* out_right, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SMD_HOTPATH_H */
