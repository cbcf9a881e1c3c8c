// smd_kernels.h — argument blocks and host-side launchers shared between the kernel translation units and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/smd_hotpath.h"

namespace smd {

// A wave owns a strip of 64 lanes = 64 consecutive columns; the outer 1 (forward) or 2 (backward) lanes on each
// side are halo (their stencil results are incomplete), so a strip advances by 62 / 60 columns.
constexpr int kFwdCols = 62;
constexpr int kBwdCols = 60;
#ifndef SMD_WAVES_PER_BLOCK
#define SMD_WAVES_PER_BLOCK 4
#endif
constexpr int kWavesPerBlock = SMD_WAVES_PER_BLOCK;
#ifndef SMD_SMOOTH_CHUNK
#define SMD_SMOOTH_CHUNK 1024
#endif
constexpr int kSmoothChunk = SMD_SMOOTH_CHUNK;   // pixels per block of the smoothness sweeps at the large scales
// Pixels per block for a scale with n pixels: the coarse scales have few pixels but the longest latency chains (their
// image taps are strided gathers from the full-resolution frame), so they get one pixel per thread and start first.
__host__ __device__ inline int smooth_chunk_px(int n) { return n > 32768 ? kSmoothChunk : 256; }
__host__ __device__ inline int smooth_chunks_of(int n) { return (n + smooth_chunk_px(n) - 1)/smooth_chunk_px(n); }
// The forward sweep streams: a wave owns 63 columns (+ 1 halo lane for the right-hand neighbour) and walks down kSmoothRows rows.
constexpr int kSmoothCols = 63, kSmoothRows = 8;
__host__ __device__ inline int smooth_units_of(int hs, int ws) { return ((ws + kSmoothCols - 1)/kSmoothCols)*((hs + kSmoothRows - 1)/kSmoothRows); }
// The sweep over the disparities (k_smooth_main) reads 12 bytes per pixel and does almost nothing with them: taller units, fewer waves and partials.
constexpr int kSmoothRowsMain = 16;
__host__ __device__ inline int smooth_units_main(int hs, int ws) { return ((ws + kSmoothCols - 1)/kSmoothCols)*((hs + kSmoothRowsMain - 1)/kSmoothRowsMain); }
constexpr int kPoseSums = 12;  // accumulated d/d(H[9], a0, a1, tz) per (support, sample)

struct ScaleSet {  // the multi-scale disparity pyramid, passed by value
  const float* p[SMD_MAX_SCALES];
  float* g[SMD_MAX_SCALES];
  int hs[SMD_MAX_SCALES];
  int ws[SMD_MAX_SCALES];
  int key[SMD_MAX_SCALES];  // dictionary key of the scale (the `s` of `loss_s / 2**s`, src/core/handlers.py:279)
  int S;
};

// Layout of the caller-kept buffer `packed` (smd_packed_supports_bytes; below 4 GB so that one buffer resource spans it):
//   texels (n,b,h+1,w+1,3): the supports as RGB texels (12 bytes: one dwordx3 load per bilinear tap, a wave's 64 adjacent taps
//                           span 768 contiguous bytes), padded by one zero texel to the right and below so that the 2x2 tap
//                           block of any clamped coordinate stays in range;
//   ypix   (b,h,w,3):       the target as RGB texels (one 12-byte load per lane and row instead of three planar ones);
//   ta     (b,h,w,4):       {S_y (3 channels), c_0}   with c = 9 S_yy - S_y^2 + 81 C2: what every scale and support shares
//   tb     (b,h,w,4):       {c_1, c_2, identity error of the automask, 0}              at a target pixel.
// followed by a small tail that is not image data (packed_tail_*):
//   rowtab  [SMD_MAX_SCALES][h+4] uint4: vertical half of the K0 up-sampling per (scale, image row), written by the prep kernel
//   arrive  [1 + b] unsigned:  arrival counters of the in-launch reductions (entry 0: loss of the forward; 1 + bi: pose sums of
//                              sample bi in the backward).  Zeroed by the prep kernel, reset to zero by the last arriver.
__host__ __device__ inline size_t packed_texel_floats(int b, int n, int h, int w) { return (size_t)n*b*(size_t)(h + 1)*(size_t)(w + 1)*3; }
__host__ __device__ inline size_t packed_ypix_floats(int b, int h, int w) { return (size_t)b*(size_t)h*(size_t)w*3; }
__host__ __device__ inline size_t packed_tpix_floats(int b, int h, int w) { return (size_t)b*(size_t)h*(size_t)w*4; }   // each of ta, tb
__host__ __device__ inline size_t packed_image_floats(int b, int n, int h, int w) {   // the part the buffer resources of the fused kernels span
  return packed_texel_floats(b, n, h, w) + packed_ypix_floats(b, h, w) + 2*packed_tpix_floats(b, h, w);
}
__host__ __device__ inline size_t packed_rowtab_offset_floats(int b, int n, int h, int w) { return (packed_image_floats(b, n, h, w) + 3) & ~(size_t)3; }   // 16-byte aligned
__host__ __device__ inline size_t packed_arrive_offset_floats(int b, int n, int h, int w) { return packed_rowtab_offset_floats(b, n, h, w) + (size_t)SMD_MAX_SCALES*(size_t)(h + 4)*4; }
// ... and (round 5) the LIVENESS table the forward leaves for the backward: which columns of each forward strip have, in ANY of the strip's rows,
// a pixel whose final selection is support k — one 64-bit lane mask per (scale, sample, forward strip, k < 4): bit l <-> column 62*sx - 1 + l.
// A backward wave (one support of one strip) whose 3x3-dilated footprint overlaps no such column has nothing to do: every gradient it would
// compute is an exact zero.  Entry layout: a reserved header of live_header_floats(b) words (the backward gets the forward's partition — rows per forward strip of each
// sample: the tapered partition differs per sample — as launch arguments fwd_rh / fwd_b1 / fwd_rh2, not from memory),
// then masks [SMD_MAX_SCALES][b][live_max_strips][4] uint64.  Every entry is owned by one forward wave, which stores it unconditionally on the
// launch's last pass: no zero-fill, no atomics.
constexpr int kLiveSupports = 4;
__host__ __device__ inline size_t live_max_strips(int h, int w) { return (size_t)((w + kFwdCols - 1)/kFwdCols)*(size_t)((h + 3)/4); }   // strips of >= 4 rows
__host__ __device__ inline size_t packed_live_offset_floats(int b, int n, int h, int w) { return (packed_arrive_offset_floats(b, n, h, w) + (((size_t)b + 1 + 3) & ~(size_t)3) + 63) & ~(size_t)63; }   // 256-byte aligned
__host__ __device__ inline size_t live_header_floats(int b) { return ((size_t)b + 63) & ~(size_t)63; }
__host__ __device__ inline size_t live_table_floats(int b, int h, int w) { return live_header_floats(b) + (size_t)SMD_MAX_SCALES*b*live_max_strips(h, w)*kLiveSupports*2; }
__host__ __device__ inline size_t packed_total_floats(int b, int n, int h, int w) { return packed_live_offset_floats(b, n, h, w) + live_table_floats(b, h, w); }

// The fused loss path (round 5): total = w_rec*l_rec + w_sm*l_sm is formed in-launch by whichever final reducer arrives second.
struct LossCombine { float* out3; unsigned* arrive; float w_rec, w_sm; };   // out3 = {total, l_rec, l_sm}; null: no combination
// The smoothness sweep over the disparities (smd_smooth_dev.h: smooth_main_block), as a kernel of its own or as guest blocks of k_recon_main.
struct SmoothFwdJob {
  const float* edge_w;     // {wx, wy} per pixel of every level, or null (no edge weighting)
  float* partial; int max_units; float* stats; float* loss;
  unsigned* arrive;        // [S*b + 1] counters of the in-launch second stage (null: two-launch form)
  double* contrib;         // [S*b]
  LossCombine comb;
};

struct ReconPrepArgs {     // k_recon_prep: texel repack + target window sums + identity error, once per sample
  const float* tgt;        // (b,3,h,w)
  const float* supp;       // (n,b,3,h,w) planar
  float* packed;           // out, layout above
  int b, n, h, w;
  int i0, ni;              // supports [i0, i0+ni) handled by this launch (ni <= 4)
  int flags;
  int rh, nsx, nsy;
  int first_pass, last_pass;
  uint4* rowtab;           // K0 fused: [S][h+4] vertical up-sampling table for the main kernel (first pass), or null
  unsigned* arrive;        // the arrival counters in the tail of `packed`: zeroed here (first pass)
  int sc_S, sc_hs[SMD_MAX_SCALES], sc_ws[SMD_MAX_SCALES];
};

struct ReconMainArgs {     // k_recon_main: warp + SSIM/L1 + min/mean over supports + automask, per (strip, sample, scale)
  const float* depth;      // (S,b,h,w), read unless depth_out is set
  float* depth_out;        // K0 fused: (S,b,h,w) written from `sc` (the low-resolution disparity pyramid), or null
  ScaleSet sc;             // K0 fused: p[s] (b,1,hs,ws), hs, ws
  float a_scale, a_off;    // K0 fused: d = a_scale*disp_up + a_off, depth = (d > 0)/max(d, eps)
  const uint4* rowtab;     // K0 fused: [S][h+4] vertical up-sampling table (written by the prep kernel)
  const float* packed;     // layout above
  const float* T;          // (n,b,4,4)
  const float* K;          // (b,4,4)
  const float* Kinv;       // (b,4,4)
  const float* noise;      // (S,b,h,w) or null
  float* err;              // (S,b,h,w) running / final error
  uint8_t* sel;            // (S,b,h,w) running / final selection
  float* partial;          // [grid blocks] DOUBLES: per-block loss sums (last pass only; 8-byte aligned)
  unsigned* arrive;        // arrival counter of the in-launch loss reduction (zero on entry; the last block resets it)
  float* loss;             // (1) out: sum of the partials x loss_scale, written by the block that arrives last (last pass only)
  double loss_scale;       // 1/(S*b*h*w)
  float* warp0;            // (n,b,3,h,w) or null
  int b, n, S, h, w;
  int i0, ni;              // supports [i0, i0+ni) handled by this launch (ni <= 4)
  int flags;
  int rh;                  // rows per strip
  int nsx, nsy;            // strips per image in x / y
  // Tapered partition: the first b1 samples (in dispatch order) use strips of `rh` rows, the remaining b - b1 samples strips of
  // `rh2` rows (nsy2 per image), so that the work units dispatched last are short and the launch does not end with a few
  // long waves running alone (b1 == b: no taper).  Loss partials: [S*b1*nsx*nsy] followed by [S*(b-b1)*nsx*nsy2].
  int b1, rh2, nsy2;
  float wscale, hscale;    // w/(w-1), h/(h-1)
  float inv_n;
  uint32_t seed_lo, seed_hi;
  int first_pass, last_pass;
  int lookahead;           // 1 or 2 row steps between a tap gather and its first use (2: the hot K0-fused instantiation with N <= 2 only)
  int share;               // 1: a block is the FOUR scales of one strip and the target-side rows reach them through one LDS ring (S == 4, strip heights multiples of 4)
  // Fused loss path (K0-fused single-pass instantiations only): blocks [main_blocks, main_blocks + guest_blocks) of the grid are GUEST blocks
  // running the smoothness sweep over `sc` (sm); comb: the reconstruction loss also takes part in the in-launch weighted sum.
  int main_blocks, guest_blocks;
  SmoothFwdJob sm;
  LossCombine comb;
  unsigned* live;          // the liveness table in the tail of `packed` (written on the last pass), or null
};

struct ReconBwdArgs {
  const float* depth; const float* packed; const float* T; const float* K; const float* Kinv;
  const uint8_t* sel;
  const float* g_loss;    // device scalar
  float g_scale;          // host-side factor of g_loss (the loss weight of the fused loss path; 1 otherwise)
  float* g_depth;         // (S,b,h,w)
  const float* g_in;      // (S,b,h,w) gradient reaching depth from other consumers, added on the last support pass, or null
  float k0_scale;         // K0 fused: != 0 -> g_depth receives dL/d(up-sampled disparity) = dL/d depth * (-depth^2 * k0_scale) (0 where the
                          //   depth is pinned), so that the K0 adjoint neither reads the depth again nor multiplies; 0 -> dL/d depth
  float* pose_partial;    // [n*b][pose_stride][kPoseSums], the first S*nstrips entries of a (support, sample) used
  unsigned* arrive;       // [b] arrival counters, one per sample (zero on entry): the block that completes a sample turns its pose sums
                          //   into g_T / g_K / g_Kinv (the former k_pose_finalize launch) and resets the counter
  float* g_T; float* g_K; float* g_Kinv;   // (n,b,4,4); (b,4,4) or null
  int b, n, S, h, w;
  int flags;
  int rh, nsx, nsy;
  int b1, rh2, nsy2;      // tapered partition, as in ReconMainArgs
  int pose_stride;        // entries reserved per (support, sample) in pose_partial: S*nsx*max(nsy, nsy2)
  int wps;                // waves per strip (1 .. min(n, 4)): the supports of a strip are split over this many waves of one block
  int scales_block;       // wps == 1, S == 4: a block is the four SCALES of one strip (they read the same target-side rows and gather near the same texels) instead of four strips of one scale
  float wscale, hscale;
  int skip_level;         // 0..2, see k_recon_bwd
  int pair;               // 1: two supports per wave (k_recon_bwd_pair; n = 2 or 4, min-reprojection, plain row loop)
  const unsigned* live;   // the forward's liveness table (smd_kernels.h: packed_live_offset_floats), or null: every wave runs its row loop
  int fwd_rh, fwd_b1, fwd_rh2;   // ... and the forward's partition: rows per forward strip of the first fwd_b1 samples / of the rest (the table is indexed by forward strips)
  float* g_direct;        // K0 fused: rows of scale `direct_scale` (a pyramid level that already has the image size: its K0 adjoint is the
  int direct_scale;       //   identity) go straight to that level's gradient tensor (b,h,w) instead of g_depth; -1: none
};

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1)/b; }

constexpr int SMD_MAX_AR_SEGMENTS = 8;
struct CropResizeArgs {    // k_crop_resize: centre crop + bilinear resize of up to 8 tensors of (planes, H, W) + the intrinsics
  const float* src[SMD_MAX_AR_SEGMENTS]; float* dst[SMD_MAX_AR_SEGMENTS];
  int planes[SMD_MAX_AR_SEGMENTS], first_plane[SMD_MAX_AR_SEGMENTS];
  int nseg;
  int H, W;                // input size
  int y0, x0, ch, cw;      // crop window
  int resample;            // 1: a crop pixel is kornia's center_crop(align_corners=False) resample of the window; 0: the window's pixel itself (crop == frame)
  int oh, ow;              // output size
  const float* K_in; float* K_out; int nK;
};
hipError_t launch_crop_resize(const CropResizeArgs& a, hipStream_t st);

// XCD-aware block -> (strip block, sample, scale) mapping for the two fused kernels (1-D grid of nbx*b*S blocks).
// MI355X dispatches workgroup p to XCD p % 8, each XCD with its own 4 MB L2.  The S scales of one image region gather from
// the same support texels, so they are placed on ONE XCD in consecutive dispatch slots: the texels are fetched from HBM once
// and re-read from that L2 by the other scales (with blockIdx.z = scale they ran a whole batch apart, on any XCD).
__device__ __forceinline__ void decode_tile(unsigned p, int nbx, int b, int S, int& xb, int& bi, int& s) {
  const unsigned nq = (unsigned)nbx*(unsigned)b, full = nq & ~7u;
  unsigned q;
  if (p < full*(unsigned)S) { const unsigned slot = p >> 3; q = (slot/(unsigned)S)*8u + (p & 7u); s = (int)(slot % (unsigned)S); }
  else { const unsigned r = p - full*(unsigned)S; q = full + r/(unsigned)S; s = (int)(r % (unsigned)S); }
  bi = (int)(q/(unsigned)nbx); xb = (int)(q - (unsigned)bi*(unsigned)nbx);
}

// (Round 2 tried one block = the four scales of ONE strip with strips scattered over the XCDs: L2 requests fell by 20 %, but
// neighbouring strips then land on different XCDs and HBM reads went back up from 143 to 196 MB; no gain in time.  Round 3's
// shared-ring forward (smd_recon_fwd.hip, SH) uses that block shape with THIS decode, reading `s` as the strip within a tile of
// four adjacent strips: the tile stays on one XCD.)
__host__ __device__ inline unsigned recon_grid_blocks(int nstrips, int b, int S, int spb = kWavesPerBlock) {
  return (unsigned)ceil_div(nstrips, spb)*(unsigned)b*(unsigned)S;
}
__device__ __forceinline__ void decode_wave(unsigned p, int wid, int nstrips, int b, int S, int& strip, int& bi, int& s) {
  int xb;
  decode_tile(p, ceil_div(nstrips, kWavesPerBlock), b, S, xb, bi, s);
  strip = xb*kWavesPerBlock + wid;
}

void note_variant(int which, const char* fmt, ...);   // smd_api.hip: the instantiation a fused launch picked (which: 0 forward, 1 backward), process-wide
// launchers (return hipError_t from hipGetLastError after the launch)
hipError_t launch_recon_prep(const ReconPrepArgs& a, hipStream_t st);
hipError_t launch_recon_main(const ReconMainArgs& a, hipStream_t st);
hipError_t launch_recon_bwd(const ReconBwdArgs& a, hipStream_t st);
hipError_t launch_sum_partials(const float* partial, int count, double scale, float* out, hipStream_t st);
hipError_t launch_pose_finalize(const float* pose_partial, int entries, int stride, const float* T, const float* K, const float* Kinv,
                                float* g_T, float* g_K, float* g_Kinv, int b, int n, hipStream_t st);

hipError_t launch_disp_to_depth_fwd(const ScaleSet& sc, int b, int h, int w, float min_depth, float max_depth,
                                    float* depth_up, float* disp_up, hipStream_t st);
struct BwdMap;
size_t disp_to_depth_bwd_tmp_floats(const ScaleSet& sc, int b, int h, int w, BwdMap* map);
// Guest work of the K0 adjoint's first launch: the per-sample epilogue of the fused backward (pose sums -> dL/dT, dL/dK, dL/dK^-1) and, in the
// fused loss path (round 5), the chain rule on to the pose network's outputs (the former smd_pose_bwd / smd_intrinsics_bwd launches) and the
// smoothness adjoint as extra blocks that write (or, for a level the reconstruction backward already wrote, add to) the level gradients.
struct PoseChain {        // all null: stop at dL/dT, dL/dK, dL/dK^-1
  const float* aa; const float* t; const uint8_t* invert; float* g_aa; float* g_t;   // (n*b,3) each; invert (n*b) or null
  const float* fs; const float* cs; float* g_fs; float* g_cs;                        // (b,2) each or null
  int h, w;
};
struct SmoothBwdJob { const float* stats; const float* edge_w; const float* g_loss; float g_scale; int blocks_per_sample; int accumulate_scale; };   // blocks_per_sample 0: none
struct PoseFinJob { ReconBwdArgs a; int entries1, entries2, b1; PoseChain chain; SmoothBwdJob sm; };
hipError_t launch_disp_to_depth_bwd(const ScaleSet& sc, int b, int h, int w, float min_depth, float max_depth,
                                    const float* depth_up, const float* g_depth_up, float* tmp, bool premultiplied, hipStream_t st,
                                    const PoseFinJob* job = nullptr, int skip_scale = -1, bool accumulate = false);

hipError_t launch_smooth_fwd(const ScaleSet& sc, int b, const float* img, int h, int w, int flags, float* loss, float* stats,
                             float* disp_grad, float* image_grad, float* ws_sums, float* edge_w, bool edges_ready, hipStream_t st);
hipError_t launch_blur3(const float* x, float* out, int planes, int h, int w, bool adjoint, hipStream_t st);   // 3x3 Gaussian, reflect border (SmoothReg(use_blur)); adjoint: its transpose
hipError_t launch_smooth_edges(const ScaleSet& sc, int b, const float* img, int h, int w, float* edge_w, hipStream_t st);   // frame-only: edge weights of every level + zeroed arrival counters
size_t smooth_edge_bytes(const ScaleSet& sc, int b);
hipError_t launch_smooth_bwd(const ScaleSet& sc, int b, const float* img, int h, int w, int flags, const float* stats,
                             const float* g_loss, const float* edge_w, hipStream_t st, float g_scale = 1.f, int accumulate_scale = -1);
hipError_t launch_smooth_main(const ScaleSet& sc, int b, const SmoothFwdJob& job, hipStream_t st);   // the sweep alone (the fused loss path without guest blocks)
void smooth_fwd_job(const ScaleSet& sc, int b, float* loss, float* stats, float* ws_sums, float* edge_w, SmoothFwdJob* job);   // carve the sweep's workspace / counters

hipError_t launch_view_synth_fwd(const float* input, const float* depth, const float* T, const float* K, const float* Kinv,
                                 float* warp, float* depth_warp, uint8_t* mask_valid, int B, int C, int h, int w, hipStream_t st);
hipError_t launch_view_synth_bwd(const float* input, const float* depth, const float* T, const float* K, const float* Kinv,
                                 const float* g_warp, const float* g_depth_warp, float* g_input, float* g_depth,
                                 float* g_T, float* g_K, float* g_Kinv, float* ws, int B, int C, int h, int w, hipStream_t st);
hipError_t launch_photo_error_fwd(const float* pred, const float* target, float* err, int N, int C, int h, int w, int flags, float w_ssim, hipStream_t st);
hipError_t launch_photo_error_bwd(const float* pred, const float* target, const float* g_err, float* g_pred, float* ws,
                                  int N, int C, int h, int w, int flags, float w_ssim, hipStream_t st);
hipError_t launch_recon_reduce_fwd(const float* err_warp, const float* err_static, const float* mask, const float* noise, uint64_t seed,
                                   float* err, uint8_t* sel, float* loss, float* ws, int n, int B, int h, int w, int flags,
                                   hipStream_t st);
hipError_t launch_recon_reduce_bwd(const uint8_t* sel, const float* g_loss, float* g_err_warp, const float* err_warp, const float* err_static,
                                   const float* mask, float* g_mask, int n, int B, int h, int w, int flags, hipStream_t st);
hipError_t launch_debug_lane_shift(float* out_left, float* out_right, hipStream_t st);
hipError_t launch_stream_copy(const void* src, void* dst, size_t nbytes, int mode, hipStream_t st);
size_t conv_head_partials(int B, int C, int h, int w);
hipError_t launch_conv_head_fwd(const void* xp, int x_bf16, const float* wgt, const float* bias, float* y, int B, int C, int h, int w, int act, hipStream_t st);
hipError_t launch_conv_head_bwd(const void* xp, int x_bf16, const float* wgt, const float* y, const float* gy, void* g_xp, float* g_w, float* g_bias, float* partial,
                                int B, int C, int h, int w, int act, hipStream_t st);
hipError_t launch_conv_thin_fwd(const float* xp, const float* wgt, float* y, int B, int C, int h, int w, hipStream_t st);
size_t conv_thin_partials(int B, int C, int h, int w);
hipError_t launch_conv_thin_bwd_wgt(const float* xp, const float* gy, float* g_w, float* partial, int B, int C, int h, int w, hipStream_t st);
hipError_t launch_conv_thin_bwd_data(const float* gy, const float* wgt, float* g_xp, int B, int C, int h, int w, hipStream_t st);
// smd_conv_mfma.hip: the wide decoder convolutions on the bf16 matrix cores, fp32 operands split into `pieces` bf16 pieces (3: fp32-class results)
size_t conv_mfma_packed_elems(int C, int CO, int pieces);
void set_conv_two_tiles(int v);     // launch-shape knob of the forward / data-gradient form (smd_api.hip: conv_two_tiles)
size_t conv_mfma_wgrad_partials(int B, int C, int CO, int h, int w);
hipError_t launch_conv_mfma_pack(const float* w, void* wp_fwd, void* wp_bwd, int C, int CO, int pieces, hipStream_t st);
size_t conv_mfma_fwd_split_elems(int B, int C, int CO, int h, int w);     // floats of K-split partial outputs the forward / the data gradient wants (0: none)
size_t conv_mfma_bwd_split_elems(int B, int C, int CO, int h, int w);
hipError_t launch_conv_mfma_fwd(const void* xp, const void* wp_fwd, void* y, float* split_ws, int B, int C, int CO, int h, int w, int pieces, hipStream_t st);
hipError_t launch_conv_mfma_bwd_data(const void* gy, const void* wp_bwd, void* g_xp, float* split_ws, int B, int C, int CO, int h, int w, int pieces, hipStream_t st);
hipError_t launch_conv_mfma_bwd_wgt(const void* xp, const void* gy, float* g_w, float* partial, int B, int C, int CO, int h, int w, int pieces, hipStream_t st);
size_t decoder_bias_partials(int B, int C, int h, int w);
hipError_t launch_elu_pad_fwd(const void* x, const float* bias, void* out, int B, int C, int h, int w, int apply_elu, int dt, hipStream_t st);
hipError_t launch_elu_pad_bwd(const void* x, const float* bias, const void* g_out, void* g_x, float* g_bias, float* ws, int B, int C, int h, int w,
                              int apply_elu, int dt, hipStream_t st);
hipError_t launch_elu_up_cat_pad_fwd(const void* a, const float* bias, const void* skip, void* out, int B, int Ca, int Cs, int h, int w, int dt,
                                     hipStream_t st);
hipError_t launch_elu_up_cat_pad_bwd(const void* a, const float* bias, const void* g_out, void* g_a, void* g_skip, float* g_bias, float* ws,
                                     int B, int Ca, int Cs, int h, int w, int dt, hipStream_t st);
int bn_chunks(int N, int HW);
hipError_t launch_bn_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float* running_mean, float* running_var,
                         float momentum, float eps, int relu, float* y, float* save_mean, float* save_invstd, float* ws, int N, int C, int HW,
                         hipStream_t st);
hipError_t launch_bn_bwd(const float* x, const float* y, const float* g_y, const float* gamma, const float* save_mean, const float* save_invstd,
                         int relu, float* g_x, float* g_res, float* g_gamma, float* g_beta, float* ws, int N, int C, int HW, hipStream_t st);
hipError_t launch_maxpool_fwd(const float* x, float* y, uint8_t* idx, size_t planes, int H, int W, hipStream_t st);
hipError_t launch_maxpool_bwd(const float* g_y, const uint8_t* idx, float* g_x, size_t planes, int H, int W, hipStream_t st);
int dwconv_tiles(int H, int W);
hipError_t launch_dwconv7(const float* x, const float* w, const float* bias, float* y, int N, int C, int H, int W, int flip, hipStream_t st);
hipError_t launch_dwconv7_wrw(const float* x, const float* gy, float* gw, float* gb, float* ws, int N, int C, int H, int W, hipStream_t st);
int ln_cf_chunks(size_t npix);
hipError_t launch_ln_cf_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_bf16, float* mean, float* rstd, int N, int C, int HW,
                            float eps, hipStream_t st);
hipError_t launch_ln_cf_bwd(const float* x, const void* g_y, int g_bf16, const float* gamma, const float* mean, const float* rstd, float* g_x, float* g_gamma,
                            float* g_beta, float* ws, int N, int C, int HW, hipStream_t st);
int regr_blocks(size_t N);
hipError_t launch_regression_fwd(const float* pred, const float* target, const uint8_t* mask, size_t N, int flags, float* loss, float* err,
                                 float* stats, float* ws, hipStream_t st);
hipError_t launch_regression_bwd(const float* pred, const float* target, const uint8_t* mask, size_t N, int flags, float* stats,
                                 const float* g_loss, float* g_pred, float* g_target, float* ws, hipStream_t st);
hipError_t launch_pose_fwd(const float* aa, const float* t, const uint8_t* invert, int N, float* T, hipStream_t st);
hipError_t launch_pose_bwd(const float* aa, const float* t, const uint8_t* invert, int N, const float* g_T, float* g_aa, float* g_t, hipStream_t st);
hipError_t launch_intrinsics_fwd(const float* fs, const float* cs, const float* Kin, int b, int h, int w, float* K, float* Kinv, hipStream_t st);
hipError_t launch_intrinsics_bwd(const float* fs, const float* cs, int b, int h, int w, const float* g_K, const float* g_Kinv,
                                 float* g_fs, float* g_cs, hipStream_t st);



// Rows per strip.  More, shorter waves than fit at once balance better than one resident round (the waves of a round do
// not finish together), and the forward kernel gains from short strips even though each pays two halo rows.  Measured
// (scripts/dev/microbench.py, rh sweep 4..64): cfg 2 forward 93 -> 87 us, backward 221 -> 203 us; cfg 4 forward 229 -> 188 us,
// backward 422 -> 397 us; cfg 5 forward 513 -> 430 us.  Rule (round 2, with the tapered tail): about 6k waves — cfg 2: 16 rows,
// backward 127 -> 115 us —; the forward never longer than 16 rows.
inline int pick_rows_per_strip(int b, int S, int h, int w, int cols, int halo) {
  const int nsx = ceil_div(w, cols);
  const long target_waves = 6144;   // 1.5 x the chip's 4096 wave slots; with the tapered tail (smd_api.hip: taper) fewer, longer waves win
  const int rh_max = 16;   // forward: short strips win; backward: <= 16 rows keep the cross-support sum of dL/d depth in LDS (k_recon_bwd, ACC)
  (void)cols;
  int best = 8;
  for (int rh = rh_max; rh >= 8; rh -= 4) {
    long waves = (long)nsx*ceil_div(h, rh)*b*S;
    best = rh;
    if (waves >= target_waves) break;
  }
  (void)halo;
  return best;
}

}  // namespace smd
