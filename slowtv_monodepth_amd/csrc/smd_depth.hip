// smd_depth.hip — K0: bilinear upsample of the multi-scale sigmoid disparity + conversion to depth, and its adjoint.
//
// Restates `F.interpolate(mode='bilinear', align_corners=False)` (called through src/tools/ops.py:311-314 at
// src/core/trainer.py:320) and `to_scaled` / `to_inv` (src/tools/geometry.py:62-90, applied at trainer.py:321).
// One launch handles every scale and writes the scale-major (S,b,h,w) stack the fused kernels read, so the
// `torch.stack` of src/core/handlers.py:48 never materialises.
#include <string.h>
#include "smd_common.h"
#include "smd_kernels.h"
#include "smd_pose_fin.h"
#include "smd_pose_dev.h"
#include "smd_smooth_dev.h"

namespace smd {

// ATen area_pixel_compute_source_index (align_corners=False, non-cubic) + index/lambda split.
__device__ __forceinline__ void src_index(int dst, float scale, int n_in, int& i0, int& i1, float& l1) {
  float src = fmaxf(fmaf(scale, (float)dst + 0.5f, -0.5f), 0.f);
  i0 = min((int)src, n_in - 1);
  i1 = min(i0 + 1, n_in - 1);
  l1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
}

__global__ __launch_bounds__(256) void k_disp_to_depth_fwd(const ScaleSet sc, int b, int h, int w, float a_scale, float a_off,
                                                           float* __restrict__ depth_up, float* __restrict__ disp_up) {
  const int s = blockIdx.z, bi = blockIdx.y;
  const int hs = sc.hs[s], ws = sc.ws[s];
  const float* __restrict__ src = sc.p[s] + (size_t)bi*hs*ws;
  const float sy = (float)hs/(float)h, sx = (float)ws/(float)w;
  const size_t obase = ((size_t)s*b + bi)*h*w;
  for (int pix = blockIdx.x*256 + threadIdx.x; pix < h*w; pix += gridDim.x*256) {
    const int v = pix/w, u = pix - v*w;
    int y0, y1, x0, x1; float ly, lx;
    src_index(v, sy, hs, y0, y1, ly);
    src_index(u, sx, ws, x0, x1, lx);
    float p00 = src[y0*ws + x0], p01 = src[y0*ws + x1], p10 = src[y1*ws + x0], p11 = src[y1*ws + x1];
    float val = (1.f - ly)*((1.f - lx)*p00 + lx*p01) + ly*((1.f - lx)*p10 + lx*p11);
    if (disp_up) disp_up[obase + pix] = val;
    float d = fmaf(a_scale, val, a_off);
    depth_up[obase + pix] = (d > 0.f) ? 1.f/fmaxf(d, kEps32) : 0.f;
  }
}

hipError_t launch_disp_to_depth_fwd(const ScaleSet& sc, int b, int h, int w, float min_depth, float max_depth,
                                    float* depth_up, float* disp_up, hipStream_t st) {
  float a_scale = 1.f, a_off = 0.f;
  if (min_depth > 0.f || max_depth > 0.f) {  // to_scaled: i_max = 1/min, i_min = 1/max (0 if unset)
    const float i_max = 1.f/min_depth, i_min = max_depth > 0.f ? 1.f/max_depth : 0.f;
    a_scale = i_max - i_min; a_off = i_min;
  }
  dim3 grid(min(ceil_div(h*w, 256), 512), b, sc.S);
  hipLaunchKernelGGL(k_disp_to_depth_fwd, grid, dim3(256), 0, st, sc, b, h, w, a_scale, a_off, depth_up, disp_up);
  return hipGetLastError();
}

// Adjoint.  depth = 1/d on the pass-through branch, so d depth/d d = -depth^2 (0 where depth is 0 or pinned at 1/eps).
// A scale whose disparity already has the image size is a pure element-wise product.  For the others the bilinear
// adjoint A_y^T (G .* f') A_x is evaluated separably and as GATHERS (deterministic, no atomics), ROWS FIRST: the pass over
// the full-resolution gradient (the only large operand) reads it along image rows, i.e. coalesced, and the strided column
// gathers of the second pass run on an array that is already f times smaller:
//   pass 1: tmp[jy][u] = sum_v wy(v -> jy) G[v][u] f'(depth[v][u])      thread per (jy, u), walks <= 2f+2 rows
//   pass 2: out[jy][jx] = a * sum_u wx(u -> jx) tmp[jy][u]               thread per (jy, jx), walks <= 2f+2 columns
// (Columns first — the round-1 order — made every wave of pass 1 read 64 columns f apart: 34 us at cfg 2 against 9.)
// Blocks are mapped to (scale, chunk) through a prefix table so that no block is launched for work that does not exist.
struct BwdMap { int first_block[SMD_MAX_SCALES + 1]; size_t tmp_off[SMD_MAX_SCALES]; };

__device__ __forceinline__ int scale_of_block(const BwdMap& map, int S, int blk) {
  int s = 0;
#pragma unroll
  for (int k = 1; k < SMD_MAX_SCALES; ++k) if (k < S && blk >= map.first_block[k]) s = k;
  return s;
}

__device__ __forceinline__ void footprint(int j, float f, int n_lo, int n_hi, int& lo, int& hi) {
  lo = max((int)floorf(((float)j - 0.5f)*f - 0.5f), 0);
  hi = min((int)ceilf(((float)j + 1.5f)*f - 0.5f), n_hi - 1);
  if (j == 0) lo = 0;
  if (j == n_lo - 1) hi = n_hi - 1;
}

// Streaming form of pass 1 for the pyramid levels a decoder produces (integer, even ratio f = h/hs; premultiplied gradient): a thread owns one image
// column and a chunk of C low-resolution rows, requests the C*f + f full-resolution rows that feed them at once (coalesced along the row, all
// independent) and adds each into the one or two low-resolution rows it belongs to.  Which rows and with what weight is known at compile time
// — row k of the chunk's window sits at (k + 0.5)/f - 1 low-resolution rows below the chunk's first — except at the image's top and bottom,
// where ATen clamps the source index: a row whose upper neighbour would be row -1 gives its whole weight to row 0, one whose lower neighbour
// would be row hs gives it to row hs-1 (area_pixel_compute_source_index + the index clamp of upsample_bilinear2d).  The per-thread gather it
// replaces walked 2f + 2 rows per low-resolution row, i.e. read every gradient 2.3 times, one dependent-latency loop per thread
// (15.9 -> ~7 us at cfg 2; round 4).  Rows that belong to a neighbouring chunk's low-resolution rows are read by both (f of C*f + f).
constexpr int k0_chunk_rows(int f) { return f == 2 ? 8 : (f == 4 ? 4 : (f == 8 ? 2 : 1)); }
__host__ __device__ inline bool k0_streamable(int h, int w, int hs, int ws) {
  if (hs < 1 || h % hs != 0) return false;
  const int f = h/hs;
  return (f == 2 || f == 4 || f == 8 || f == 16) && hs >= 2;
}
template <int F>
__device__ __forceinline__ void k0_bwd_v_stream(const float* __restrict__ g, float* __restrict__ out, int h, int w, int hs, int chunk, int u) {
  constexpr int C = k0_chunk_rows(F), R = C*F + F;
  const int j0 = chunk*C, vb = F*j0 - F/2;
  float r[R];
#pragma unroll
  for (int k = 0; k < R; ++k) { const int v = vb + k; r[k] = (v >= 0 && v < h) ? g[(size_t)v*w + u] : 0.f; }
  float acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 0.f;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    // un-clamped source row of window row k, relative to j0: (k + 0.5)/F - 1 = c_lo + ly
    constexpr float invF = 1.f/(float)F;
    const int c_lo = (2*k + 1 >= 2*F) ? (2*k + 1 - 2*F)/(2*F) : -1;            // floor((k + 0.5)/F) - 1, compile-time after unrolling
    const float ly = ((float)k + 0.5f)*invF - 1.f - (float)c_lo;
    const int a_lo = j0 + c_lo;                                                // absolute low-resolution rows a_lo, a_lo + 1 (wave-uniform)
    float w_lo = 1.f - ly, w_hi = ly;
    if (a_lo < 0) { w_lo = 0.f; w_hi = 1.f; }                                  // source index clamped at 0: everything to row 0
    if (a_lo + 1 > hs - 1) { w_lo = 1.f; w_hi = 0.f; }                         // neighbour index clamped at hs-1: everything to row hs-1
    if (c_lo >= 0 && c_lo < C) acc[c_lo] = fmaf(w_lo, r[k], acc[c_lo]);
    if (c_lo + 1 >= 0 && c_lo + 1 < C) acc[c_lo + 1] = fmaf(w_hi, r[k], acc[c_lo + 1]);
  }
#pragma unroll
  for (int c = 0; c < C; ++c) if (j0 + c < hs) out[(size_t)(j0 + c)*w + u] = acc[c];
}

__global__ __launch_bounds__(256) void k_disp_to_depth_bwd_v(const ScaleSet sc, const BwdMap map, int b, int h, int w, float a_scale,
                                                             const float* __restrict__ depth_up, const float* __restrict__ g_depth_up,
                                                             float* __restrict__ tmp, const PoseFinJob job) {
  // Guest work (training path): the launch that precedes this one — the fused reconstruction backward — leaves per-block pose
  // sums; one extra block per sample (its first wave) turns them into dL/dT, dL/dK, dL/dK^-1 here, beside the resampling
  // blocks, so that neither a launch of its own nor an in-launch hand-off at the tail of the big kernel is needed.
  // (The guest blocks come FIRST in dispatch order: their single wave is a chain of latencies, ~8 us, that should start with the
  // launch and hide under the resampling blocks rather than trail them.)
  const int guest = job.a.pose_partial != nullptr ? 1 : 0;
  if (guest && blockIdx.x == 0) {
    // two of the block's waves take the supports in turn (their round trips run side by side); static LDS counts against every
    // block of this launch, so no more than that
    __shared__ double scratch[fin_scratch_doubles(2)];
    __shared__ float chain_lds[SMD_MAX_SUPPORTS*16 + 32];
    const bool chain = job.chain.g_aa != nullptr;
    pose_finalize<true>(job.a, (int)blockIdx.y, (int)blockIdx.y < job.b1 ? job.entries1 : job.entries2, scratch, (int)(threadIdx.x >> 6), 2, chain ? chain_lds : nullptr);
    if (chain && (threadIdx.x >> 6) == 0) {
      // Fused loss path: the chain rule runs on to the pose network's outputs in the same wave (the former smd_pose_bwd / smd_intrinsics_bwd
      // launches): lane i < n takes support i's dL/dT from LDS through the Rodrigues adjoint, one more lane the intrinsics.
      wave_lds_sync();
      const int lane = threadIdx.x & 63, n = job.a.n, bi = (int)blockIdx.y;
      if (lane < n) pose_bwd_one(job.chain.aa, job.chain.t, job.chain.invert, lane*job.a.b + bi, chain_lds + lane*16, job.chain.g_aa, job.chain.g_t);
      else if (lane == n && job.chain.g_fs) intrinsics_bwd_one(job.chain.fs, job.chain.cs, bi, job.chain.h, job.chain.w, chain_lds + n*16, chain_lds + n*16 + 16, job.chain.g_fs, job.chain.g_cs);
    }
    return;
  }
  // depth_up == nullptr: the incoming gradient already carries d depth / d disparity (the fused backward applied it)
  const int bx = (int)blockIdx.x - guest;
  if (bx >= map.first_block[SMD_MAX_SCALES]) {   // fused loss path: the smoothness adjoint's blocks of this sample ride behind the resampling blocks
    int s_, q;
    smooth_bwd_decode(sc, bx - map.first_block[SMD_MAX_SCALES], s_, q);
    smooth_bwd_block(sc, b, s_, (int)blockIdx.y, q, job.sm.stats, job.sm.g_loss, job.sm.g_scale, job.sm.edge_w, s_ == job.sm.accumulate_scale);
    return;
  }
  const int s = scale_of_block(map, sc.S, bx);
  const int blk = bx - map.first_block[s], bi = blockIdx.y;
  const int hs = sc.hs[s], ws = sc.ws[s];
  const size_t ibase = ((size_t)s*b + bi)*h*w;
  const float dmax = 1.f/kEps32;
  if (hs == h && ws == w) {   // identity resampling: element-wise, 4 consecutive pixels per thread
    float* __restrict__ gout = sc.g[s] + (size_t)bi*hs*ws;
    const int pix0 = (blk*256 + threadIdx.x)*4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pix = pix0 + k;
      if (pix < h*w) {
        if (depth_up) { const float dep = depth_up[ibase + pix]; gout[pix] = g_depth_up[ibase + pix]*((dep < dmax) ? -dep*dep : 0.f)*a_scale; }
        else gout[pix] = g_depth_up[ibase + pix];
      }
    }
    return;
  }
  if (!depth_up && k0_streamable(h, w, hs, ws)) {   // streaming form: blocks of 256 columns x one chunk of low-resolution rows
    const int nbc = ceil_div(w, 256), chunk = blk/nbc, u = (blk - chunk*nbc)*256 + (int)threadIdx.x;
    if (u >= w) return;
    const float* g = g_depth_up + ibase;
    float* out = tmp + map.tmp_off[s] + (size_t)bi*hs*w;
    switch (h/hs) {
      case 2: k0_bwd_v_stream<2>(g, out, h, w, hs, chunk, u); break;
      case 4: k0_bwd_v_stream<4>(g, out, h, w, hs, chunk, u); break;
      case 8: k0_bwd_v_stream<8>(g, out, h, w, hs, chunk, u); break;
      default: k0_bwd_v_stream<16>(g, out, h, w, hs, chunk, u); break;
    }
    return;
  }
  const int idx = blk*256 + threadIdx.x;    // (jy, u), u fastest
  if (idx >= hs*w) return;
  const int jy = idx/w, u = idx - jy*w;
  const float sy = (float)hs/(float)h;
  int vlo, vhi;
  footprint(jy, (float)h/(float)hs, hs, h, vlo, vhi);
  const float* __restrict__ dcol = depth_up ? depth_up + ibase + u : nullptr;
  const float* __restrict__ gcol = g_depth_up + ibase + u;
  float acc = 0.f;
#pragma unroll 6   // the footprint's loads are independent: keep several in flight (the loop is latency-shaped)
  for (int v = vlo; v <= vhi; ++v) {
    int y0, y1; float ly;
    src_index(v, sy, hs, y0, y1, ly);
    const float wy = ((y0 == jy) ? 1.f - ly : 0.f) + ((y1 == jy) ? ly : 0.f);
    float gv = gcol[(size_t)v*w];
    if (dcol) { const float dep = dcol[(size_t)v*w]; gv *= (dep < dmax) ? -dep*dep : 0.f; }
    acc = fmaf(wy, gv, acc);
  }
  tmp[map.tmp_off[s] + ((size_t)bi*hs + jy)*w + u] = acc;
}

__global__ __launch_bounds__(256) void k_disp_to_depth_bwd_h(const ScaleSet sc, const BwdMap map, int b, int h, int w, float a_scale,
                                                             const float* __restrict__ tmp, int accumulate) {
  const int s = scale_of_block(map, sc.S, blockIdx.x);
  const int blk = blockIdx.x - map.first_block[s], bi = blockIdx.y;
  const int hs = sc.hs[s], ws = sc.ws[s];
  if (hs == h && ws == w) return;           // handled element-wise in pass 1
  const int lp = blk*256 + threadIdx.x;
  if (lp >= hs*ws) return;
  const int jy = lp/ws, jx = lp - jy*ws;
  const float sx = (float)ws/(float)w;
  int ulo, uhi;
  footprint(jx, (float)w/(float)ws, ws, w, ulo, uhi);
  const float* __restrict__ row = tmp + map.tmp_off[s] + ((size_t)bi*hs + jy)*w;
  float acc = 0.f;
#pragma unroll 6
  for (int u = ulo; u <= uhi; ++u) {
    int x0, x1; float lx;
    src_index(u, sx, ws, x0, x1, lx);
    const float wx = ((x0 == jx) ? 1.f - lx : 0.f) + ((x1 == jx) ? lx : 0.f);
    acc = fmaf(wx, row[u], acc);
  }
  float* out = sc.g[s] + (size_t)bi*hs*ws + lp;
  *out = accumulate ? *out + acc*a_scale : acc*a_scale;   // accumulate: the smoothness adjoint already wrote its share (fused loss path)
}

size_t disp_to_depth_bwd_tmp_floats(const ScaleSet& sc, int b, int h, int w, BwdMap* map) {
  size_t off = 0;
  for (int s = 0; s < sc.S; ++s) {
    if (map) map->tmp_off[s] = off;
    if (!(sc.hs[s] == h && sc.ws[s] == w)) off += (size_t)b*sc.hs[s]*w;
  }
  return off;
}

hipError_t launch_disp_to_depth_bwd(const ScaleSet& sc, int b, int h, int w, float min_depth, float max_depth,
                                    const float* depth_up, const float* g_depth_up, float* tmp, bool premultiplied, hipStream_t st,
                                    const PoseFinJob* job, int skip_scale, bool accumulate) {
  float a_scale = 1.f;
  if (min_depth > 0.f || max_depth > 0.f) a_scale = 1.f/min_depth - (max_depth > 0.f ? 1.f/max_depth : 0.f);
  if (premultiplied) { a_scale = 1.f; depth_up = nullptr; }
  BwdMap m1, m2;
  disp_to_depth_bwd_tmp_floats(sc, b, h, w, &m1);
  m2 = m1;
  int n1 = 0, n2 = 0;
  bool resampled = false;
  for (int s = 0; s < SMD_MAX_SCALES; ++s) {
    m1.first_block[s] = n1; m2.first_block[s] = n2;
    if (s < sc.S) {
      const bool ident = sc.hs[s] == h && sc.ws[s] == w;
      if (s == skip_scale) continue;   // the producer of g_depth_up already wrote this (identity) level's gradient in place: no blocks
      const bool stream = premultiplied && !ident && k0_streamable(h, w, sc.hs[s], sc.ws[s]);
      n1 += ident ? ceil_div(h*w, 1024) : (stream ? ceil_div(w, 256)*ceil_div(sc.hs[s], k0_chunk_rows(h/sc.hs[s])) : ceil_div(sc.hs[s]*w, 256));
      n2 += ident ? 0 : ceil_div(sc.hs[s]*sc.ws[s], 256);
      resampled |= !ident;
    }
  }
  m1.first_block[SMD_MAX_SCALES] = n1; m2.first_block[SMD_MAX_SCALES] = n2;
  PoseFinJob none;
  memset(&none, 0, sizeof(none));
  const int sm_blocks = job ? job->sm.blocks_per_sample : 0;   // guest blocks of the smoothness adjoint (fused loss path)
  if (n1 > 0) hipLaunchKernelGGL(k_disp_to_depth_bwd_v, dim3(n1 + (job ? 1 : 0) + sm_blocks, b), dim3(256), 0, st, sc, m1, b, h, w, a_scale, depth_up, g_depth_up, tmp,
                                 job ? *job : none);
  else if (job) return hipErrorInvalidValue;
  if (resampled) hipLaunchKernelGGL(k_disp_to_depth_bwd_h, dim3(n2, b), dim3(256), 0, st, sc, m2, b, h, w, a_scale, tmp, accumulate ? 1 : 0);
  return hipGetLastError();
}

}  // namespace smd
