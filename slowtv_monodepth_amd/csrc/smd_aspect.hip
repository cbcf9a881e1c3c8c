// smd_aspect.hip — GPU-side aspect-ratio augmentation: centre crop + bilinear resize of every image
// tensor of a training batch, and the matching update of the intrinsics, as ONE launch.
//
// Restates `aspect_ratio_aug` = `crop_aug` + `resize_aug` (src/core/aspect_ratio.py:35-166): the reference crops
// x.imgs / y.imgs / x.supp_imgs / y.supp_imgs (and depth maps) with `kornia.geometry.transform.center_crop(size, mode='bilinear',
// align_corners=False)` (:78-84), then resizes the crops with `F.interpolate(bilinear, align_corners=False)` (:141-151);
// `centre_crop_K` and `resize_K` (src/tools/geometry.py:233-263) follow the images.  Here the two passes are composed: an output
// pixel is the bilinear sample of the CROP (neighbour indices clamped to the crop's own border, exactly what interpolating the
// materialised crop does), and each of the up to four crop pixels it blends is evaluated on the fly from the un-cropped tensor —
// the crop is never written.
// What a crop pixel is (round 4; oracle/aspect_ratio_oracle.py has the derivation from kornia 0.6.10's call chain): kornia warps
// between the integer window (start = int(H/2 - h/2)) and the output box with `warp_affine`, whose homography is normalised with
// (n - 1) denominators (align_corners=True convention) while `affine_grid` / `grid_sample` run with the caller's align_corners=False
// and zero padding.  The conventions do not cancel: crop column i is the bilinear sample of the source at
//     x(i) = ((i + 0.5)(w - 1)/w + x0) * W/(W - 1) - 0.5          (likewise in y; taps outside the image read 0)
// — a slightly zoomed resample, not the slice x0 + i (which it is for align_corners=True only).  kornia is absent from the build
// image, so this half stays "parity unpinned" (DESIGN.md §2); the resize half is pinned on reference fixtures.  A crop of the full
// frame means "no crop" (the augmentation's not-applied branch resizes only, aspect_ratio.py:60) and reads the pixels themselves.
#include "smd_common.h"
#include "smd_kernels.h"

namespace smd {

__device__ __forceinline__ void ar_src_index(int dst, float scale, int n_in, int& i0, int& i1, float& l1) {   // ATen area_pixel_compute_source_index, align_corners=False
  const float src = fmaxf(fmaf(scale, (float)dst + 0.5f, -0.5f), 0.f);
  i0 = min((int)src, n_in - 1);
  i1 = min(i0 + 1, n_in - 1);
  l1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
}

// One block row (blockIdx.y) per image plane over all segments; a segment = one tensor (its planes share the geometry).
__global__ __launch_bounds__(256) void k_crop_resize(const CropResizeArgs a) {
  int seg = 0;
#pragma unroll
  for (int k = 1; k < SMD_MAX_AR_SEGMENTS; ++k) if (k < a.nseg && (int)blockIdx.y >= a.first_plane[k]) seg = k;
  const int plane = (int)blockIdx.y - a.first_plane[seg];
  if (seg == a.nseg - 1 && a.K_in != nullptr && plane >= a.planes[seg]) {
    // the intrinsics ride along as a pseudo-segment: centre_crop_K then resize_K (geometry.py:233-263) on (nK,4,4)
    for (int e = blockIdx.x*256 + threadIdx.x; e < a.nK*16; e += gridDim.x*256) {
      const int r = (e >> 2) & 3, c = e & 3;
      float v = a.K_in[e];
      if (r == 0 && c == 2) v *= (float)a.cw/(float)a.W;          // K[..., 0, 2] *= crop_w / w
      if (r == 1 && c == 2) v *= (float)a.ch/(float)a.H;
      if (r == 0) v *= (float)a.ow/(float)a.cw;                   // K[..., 0, :] *= new_w / crop_w
      if (r == 1) v *= (float)a.oh/(float)a.ch;
      a.K_out[e] = v;
    }
    return;
  }
  if (plane >= a.planes[seg]) return;
  const float* __restrict__ img = a.src[seg] + (size_t)plane*a.H*a.W;
  float* __restrict__ dst = a.dst[seg] + (size_t)plane*a.oh*a.ow;
  const float sy = (float)a.ch/(float)a.oh, sx = (float)a.cw/(float)a.ow;
  const bool same = (a.ch == a.oh && a.cw == a.ow);
  // source coordinate of crop row / column i (double: the product reaches ~10^3 and the fraction is a blend weight)
  const double ky = (double)(a.ch - 1)/(double)a.ch, my = (double)a.H/(double)(a.H - 1), kx = (double)(a.cw - 1)/(double)a.cw, mx = (double)a.W/(double)(a.W - 1);
  auto crop_px = [&](int yc, int xc) -> float {
    if (!a.resample) return img[(size_t)(a.y0 + yc)*a.W + (a.x0 + xc)];
    const double ys = (((double)yc + 0.5)*ky + (double)a.y0)*my - 0.5, xs = (((double)xc + 0.5)*kx + (double)a.x0)*mx - 0.5;
    const double yf = floor(ys), xf = floor(xs);
    const int iy = (int)yf, ix = (int)xf;
    const float fy = (float)(ys - yf), fx = (float)(xs - xf);
    // grid_sample(bilinear, zeros): nw * v(iy, ix) + ne * v(iy, ix+1) + sw * v(iy+1, ix) + se * v(iy+1, ix+1), taps outside the image are 0
    auto tap = [&](int y, int x) -> float { return (y >= 0 && y < a.H && x >= 0 && x < a.W) ? img[(size_t)y*a.W + x] : 0.f; };
    const float nw = (1.f - fx)*(1.f - fy), ne = fx*(1.f - fy), sw = (1.f - fx)*fy, se = fx*fy;
    return nw*tap(iy, ix) + ne*tap(iy, ix + 1) + sw*tap(iy + 1, ix) + se*tap(iy + 1, ix + 1);
  };
  for (int pix = blockIdx.x*256 + threadIdx.x; pix < a.oh*a.ow; pix += gridDim.x*256) {
    const int v = pix/a.ow, u = pix - v*a.ow;
    if (same) { dst[pix] = crop_px(v, u); continue; }
    int ya, yb, xa, xb; float ly, lx;
    ar_src_index(v, sy, a.ch, ya, yb, ly);
    ar_src_index(u, sx, a.cw, xa, xb, lx);
    const float p00 = crop_px(ya, xa), p01 = crop_px(ya, xb), p10 = crop_px(yb, xa), p11 = crop_px(yb, xb);
    dst[pix] = (1.f - ly)*((1.f - lx)*p00 + lx*p01) + ly*((1.f - lx)*p10 + lx*p11);
  }
}

hipError_t launch_crop_resize(const CropResizeArgs& a, hipStream_t st) {
  int total = 0;
  for (int k = 0; k < a.nseg; ++k) total += a.planes[k];
  if (a.K_in) total += 1;
  hipLaunchKernelGGL(k_crop_resize, dim3(min(ceil_div(a.oh*a.ow, 256), 256), total), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace smd
