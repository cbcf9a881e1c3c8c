// smd_recon_fwd.hip — fused forward of the view-synthesis photometric loss for gfx950.
//
// One launch covers every (strip, sample, scale).  A wave owns 64 consecutive columns (62 interior + 1 halo
// lane per side) of a `rh`-row strip and streams down the rows:
//   per row    : depth + target row (coalesced), per support: 6 FMAs of projective geometry, one rcp, 6 unaligned
//                8-byte gathers from the planar support frame, bilinear blend                    (K1c-K1g of SURVEY §2.2)
//   horizontal : 3-tap sums of {x, x^2, xy} (and {y, y^2}) through DPP wave shifts, with reflection weights
//   vertical   : forward-accumulated row sums (two registers per quantity), so the 3x3 SSIM window never
//                touches LDS or HBM                                                              (K2a-K2c)
//   per pixel  : SSIM + L1, min/mean over supports in registers, automask against the identity error,
//                error + selection written once, loss reduced per wave                           (K2d-K2f)
// The identity ("static") error does not depend on the scale, so it is produced once per sample by the same
// template with WARP = false instead of S times as in the reference (reconstruction.py:71).
#include "smd_common.h"
#include "smd_kernels.h"

namespace smd {

template <int NI, bool WARP>
__global__ __launch_bounds__(256) void k_recon_fwd(const ReconFwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int strip = blockIdx.x*kWavesPerBlock + wid;
  if (strip >= a.nsx*a.nsy) return;
  const int sxi = strip % a.nsx, syi = strip/a.nsx;
  const int bi = blockIdx.y, s = blockIdx.z;
  const int h = a.h, w = a.w;
  const int c0 = sxi*kFwdCols;
  const int r0 = syi*a.rh, r1 = min(r0 + a.rh, h);

  const int u = c0 - 1 + lane;
  const bool col_ok = (u >= 0) && (u < w);
  const int uc = min(max(u, 0), w - 1);
  const bool interior = (lane >= 1) && (lane <= kFwdCols) && (u < w);
  float wl, wr;
  reflect_weights(uc, w, wl, wr);
  if (!col_ok) { wl = 0.f; wr = 0.f; }
  const float uf = (float)u;

  const bool use_min = a.flags & SMD_USE_MIN;
  const bool automask = a.flags & SMD_USE_AUTOMASK;
  const bool l1_only = a.flags & SMD_LOSS_L1;
  const size_t hw = (size_t)h*w;

  Cam cam[NI];
  const float* splane[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int i = a.i0 + k;
    if (WARP) make_cam(cam[k], a.T + ((size_t)i*a.b + bi)*16, a.K + (size_t)bi*16, a.Kinv + (size_t)bi*16);
    splane[k] = a.supp + ((size_t)i*a.b + bi)*3*hw;
  }
  const float* tgt_b = a.tgt + (size_t)bi*3*hw;
  const float* depth_sb = WARP ? a.depth + ((size_t)s*a.b + bi)*hw : nullptr;
  const size_t out_base = ((size_t)s*a.b + bi)*hw;

  // forward-accumulated vertical sums: acc1 -> row being completed next, acc0 -> the row after it
  float ay1[3][2] = {}, ay0[3][2] = {};
  float ax1[NI][3][3] = {}, ax0[NI][3][3] = {};
  float yprev[3] = {}, xprev[NI][3] = {};
  float lsum = 0.f;

  const int jstart = max(r0 - 1, 0);
  for (int j = jstart; j <= r1; ++j) {
    const bool compute = j < h;
    float hy[3][2] = {}, hxs[NI][3][3] = {};
    float ycur[3] = {}, xcur[NI][3] = {};

    if (compute) {
      float D = 0.f;
      if (WARP) D = col_ok ? depth_sb[(size_t)j*w + uc] : 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float y = col_ok ? tgt_b[(size_t)c*hw + (size_t)j*w + uc] : 0.f;
        ycur[c] = y;
        if (!l1_only) { hy[c][0] = hsum3(y, wl, wr); hy[c][1] = hsum3(y*y, wl, wr); }
      }
      const float vf = (float)j;
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        if (WARP) {
          const Cam& cm = cam[k];
          float hx = fmaf(cm.H[0], uf, fmaf(cm.H[1], vf, cm.H[2]));
          float hyy = fmaf(cm.H[3], uf, fmaf(cm.H[4], vf, cm.H[5]));
          float hz = fmaf(cm.H[6], uf, fmaf(cm.H[7], vf, cm.H[8]));
          float nx = fmaf(D, hx, cm.a0), ny = fmaf(D, hyy, cm.a1), yz = fmaf(D, hz, cm.tz);
          float rz = __builtin_amdgcn_rcpf(fmaxf(yz, kZMin));
          float sx = fmaf(nx*rz, a.wscale, -0.5f), sy = fmaf(ny*rz, a.hscale, -0.5f);
          Taps tp = make_taps(sx, sy, h, w);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            float x = bilerp(splane[k] + (size_t)c*hw, tp, w);
            xcur[k][c] = col_ok ? x : 0.f;
          }
          if (a.warp0 != nullptr && s == 0 && interior && j >= r0 && j < r1) {
            float* wo = a.warp0 + ((size_t)(a.i0 + k)*a.b + bi)*3*hw + (size_t)j*w + u;
#pragma unroll
            for (int c = 0; c < 3; ++c) wo[(size_t)c*hw] = xcur[k][c];
          }
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) xcur[k][c] = col_ok ? splane[k][(size_t)c*hw + (size_t)j*w + uc] : 0.f;
        }
        if (!l1_only) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            float x = xcur[k][c];
            hxs[k][c][0] = hsum3(x, wl, wr);
            hxs[k][c][1] = hsum3(x*x, wl, wr);
            hxs[k][c][2] = hsum3(x*ycur[c], wl, wr);
          }
        }
      }
    }

    // ---- emit row v = j-1 ------------------------------------------------------------------
    const int v = j - 1;
    if (v >= r0 && v < r1) {
      float lo_v, hi_v;
      reflect_weights(v, h, lo_v, hi_v);
      const float ninth = 1.f/9.f;
      float my[3], cy1[3], cy2[3];
      if (!l1_only) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float sy_ = fmaf(hi_v, hy[c][0], ay1[c][0])*ninth, syy = fmaf(hi_v, hy[c][1], ay1[c][1])*ninth;
          my[c] = sy_; cy1[c] = fmaf(sy_, sy_, kC1); cy2[c] = (syy - sy_*sy_) + kC2;
        }
      }
      float best = 0.f, acc = 0.f;
      int bsel = a.i0;
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        float es = 0.f, el = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          el += fabsf(xprev[k][c] - yprev[c]);
          if (!l1_only) {
            float mx = fmaf(hi_v, hxs[k][c][0], ax1[k][c][0])*ninth;
            float exx = fmaf(hi_v, hxs[k][c][1], ax1[k][c][1])*ninth;
            float exy = fmaf(hi_v, hxs[k][c][2], ax1[k][c][2])*ninth;
            es += ssim_err(mx, exx, exy, my[c], cy1[c], cy2[c]);
          }
        }
        float e = l1_only ? el*(1.f/3.f) : fmaf(kWSsim/3.f, es, ((1.f - kWSsim)/3.f)*el);
        if (k == 0) { best = e; acc = e; }
        else {
          acc += e;
          if (e < best) { best = e; bsel = a.i0 + k; }
        }
      }
      if (interior) {
        const size_t idx = out_base + (size_t)v*w + u;
        if (!a.first_pass) {
          float prev = a.err[idx];
          if (use_min) { if (!(best < prev)) { best = prev; bsel = a.sel ? a.sel[idx] : 0; } }
          else acc += prev;
        }
        if (!a.last_pass) {
          a.err[idx] = use_min ? best : acc;
          if (a.sel) a.sel[idx] = (uint8_t)bsel;
        } else {
          float e = use_min ? best : acc/(float)a.n;
          if (!use_min) bsel = 0;
          if (automask) {
            float nz = a.noise ? a.noise[idx] : gauss_noise(a.seed_lo, a.seed_hi, (uint32_t)idx);
            float est = fmaf(kEps32, nz, a.e_static[(size_t)bi*hw + (size_t)v*w + u]);
            if (est < e) { e = est; bsel = SMD_SEL_MASKED; }
          }
          a.err[idx] = e;
          if (a.sel) a.sel[idx] = (uint8_t)bsel;
          lsum += e;
        }
      }
    }

    // ---- roll the vertical accumulators -----------------------------------------------------
    float lo_n, hi_n;
    reflect_weights(min(j + 1, h - 1), h, lo_n, hi_n);  // lo-weight with which row j enters out(j+1)
    if (j + 1 >= h) lo_n = 0.f;
    if (!l1_only) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) { ay1[c][q] = ay0[c][q] + hy[c][q]; ay0[c][q] = lo_n*hy[c][q]; }
#pragma unroll
      for (int k = 0; k < NI; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int q = 0; q < 3; ++q) { ax1[k][c][q] = ax0[k][c][q] + hxs[k][c][q]; ax0[k][c][q] = lo_n*hxs[k][c][q]; }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) yprev[c] = ycur[c];
#pragma unroll
    for (int k = 0; k < NI; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) xprev[k][c] = xcur[k][c];
  }

  if (a.last_pass && a.partial != nullptr) {
    float tot = wave_sum(lsum);
    if (lane == 0) a.partial[((size_t)s*a.b + bi)*(a.nsx*a.nsy) + strip] = tot;
  }
}

hipError_t launch_recon_fwd(const ReconFwdArgs& a, int ni, bool warp, hipStream_t st) {
  dim3 grid(ceil_div(a.nsx*a.nsy, kWavesPerBlock), a.b, a.S), block(64*kWavesPerBlock);
  if (warp) {
    if (ni == 1) hipLaunchKernelGGL((k_recon_fwd<1, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_recon_fwd<2, true>), grid, block, 0, st, a);
  } else {
    if (ni == 1) hipLaunchKernelGGL((k_recon_fwd<1, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_recon_fwd<2, false>), grid, block, 0, st, a);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Deterministic second stage of every scalar reduction: one block, fp64 accumulation.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sum_partials(const float* __restrict__ partial, int count, double scale, float* out) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < count; i += 256) acc += (double)partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int sft = 128; sft > 0; sft >>= 1) {
    if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0]*scale);
}

hipError_t launch_sum_partials(const float* partial, int count, double scale, float* out, hipStream_t st) {
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, st, partial, count, scale, out);
  return hipGetLastError();
}

__global__ void k_debug_lane_shift(float* out_left, float* out_right) {
  float x = (float)threadIdx.x;
  out_left[threadIdx.x] = lane_left(x);
  out_right[threadIdx.x] = lane_right(x);
}
hipError_t launch_debug_lane_shift(float* out_left, float* out_right, hipStream_t st) {
  hipLaunchKernelGGL(k_debug_lane_shift, dim3(1), dim3(64), 0, st, out_left, out_right);
  return hipGetLastError();
}

}  // namespace smd
