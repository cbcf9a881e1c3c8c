// smd_depth.hip — K0: bilinear upsample of the multi-scale sigmoid disparity + conversion to depth, and its adjoint.
//
// Restates `F.interpolate(mode='bilinear', align_corners=False)` (called through src/tools/ops.py:311-314 at
// src/core/trainer.py:320) and `to_scaled` / `to_inv` (src/tools/geometry.py:62-90, applied at trainer.py:321).
// One launch handles every scale and writes the scale-major (S,b,h,w) stack the fused kernels read, so the
// `torch.stack` of src/core/handlers.py:48 never materialises.
#include "smd_common.h"
#include "smd_kernels.h"

namespace smd {

// ATen area_pixel_compute_source_index (align_corners=False, non-cubic) + index/lambda split.
__device__ __forceinline__ void src_index(int dst, float scale, int n_in, int& i0, int& i1, float& l1) {
  float src = fmaxf(scale*((float)dst + 0.5f) - 0.5f, 0.f);
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

// Adjoint.  depth = 1/d on the pass-through branch, so d depth/d d = -depth^2 (0 where depth is 0 or pinned at 1/eps);
// the bilinear adjoint is evaluated as a GATHER per low-resolution pixel (deterministic, no atomics): 16 lanes
// share one low-res pixel and stride over its full-resolution footprint.
__global__ __launch_bounds__(256) void k_disp_to_depth_bwd(const ScaleSet sc, int b, int h, int w, float a_scale,
                                                           const float* __restrict__ depth_up, const float* __restrict__ g_depth_up) {
  const int s = blockIdx.z, bi = blockIdx.y;
  const int hs = sc.hs[s], ws = sc.ws[s];
  float* __restrict__ gout = sc.g[s] + (size_t)bi*hs*ws;
  const size_t ibase = ((size_t)s*b + bi)*h*w;
  const float sy = (float)hs/(float)h, sx = (float)ws/(float)w;
  const float fy = (float)h/(float)hs, fx = (float)w/(float)ws;
  const int sub = threadIdx.x & 15;
  const float dmax = 1.f/kEps32;
  for (int lp = blockIdx.x*16 + (threadIdx.x >> 4); lp < ((hs*ws + 15)/16)*16; lp += gridDim.x*16) {
    const bool live = lp < hs*ws;
    const int jy = live ? lp/ws : 0, jx = live ? lp - (lp/ws)*ws : 0;
    float acc = 0.f;
    if (live) {
      int vlo = max((int)floorf(((float)jy - 0.5f)*fy - 0.5f) - 1, 0), vhi = min((int)ceilf(((float)jy + 1.5f)*fy - 0.5f) + 1, h - 1);
      int ulo = max((int)floorf(((float)jx - 0.5f)*fx - 0.5f) - 1, 0), uhi = min((int)ceilf(((float)jx + 1.5f)*fx - 0.5f) + 1, w - 1);
      if (jy == 0) vlo = 0;
      if (jx == 0) ulo = 0;
      if (jy == hs - 1) vhi = h - 1;
      if (jx == ws - 1) uhi = w - 1;
      const int nu = uhi - ulo + 1, cnt = (vhi - vlo + 1)*nu;
      for (int k = sub; k < cnt; k += 16) {
        const int v = vlo + k/nu, u = ulo + k - (k/nu)*nu;
        int y0, y1, x0, x1; float ly, lx;
        src_index(v, sy, hs, y0, y1, ly);
        src_index(u, sx, ws, x0, x1, lx);
        float wy = ((y0 == jy) ? 1.f - ly : 0.f) + ((y1 == jy) ? ly : 0.f);
        float wx = ((x0 == jx) ? 1.f - lx : 0.f) + ((x1 == jx) ? lx : 0.f);
        if (wy != 0.f && wx != 0.f) {
          float dep = depth_up[ibase + (size_t)v*w + u];
          float dd = (dep < dmax) ? -dep*dep : 0.f;
          acc = fmaf(wy*wx, g_depth_up[ibase + (size_t)v*w + u]*dd, acc);
        }
      }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (live && sub == 0) gout[lp] = acc*a_scale;
  }
}

hipError_t launch_disp_to_depth_bwd(const ScaleSet& sc, int b, int h, int w, float min_depth, float max_depth,
                                    const float* depth_up, const float* g_depth_up, hipStream_t st) {
  float a_scale = 1.f;
  if (min_depth > 0.f || max_depth > 0.f) a_scale = 1.f/min_depth - (max_depth > 0.f ? 1.f/max_depth : 0.f);
  int maxpix = 0;
  for (int s = 0; s < sc.S; ++s) maxpix = max(maxpix, sc.hs[s]*sc.ws[s]);
  dim3 grid(min(ceil_div(maxpix, 16), 2048), b, sc.S);
  hipLaunchKernelGGL(k_disp_to_depth_bwd, grid, dim3(256), 0, st, sc, b, h, w, a_scale, depth_up, g_depth_up);
  return hipGetLastError();
}

}  // namespace smd
