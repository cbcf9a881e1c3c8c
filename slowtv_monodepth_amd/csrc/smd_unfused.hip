// smd_unfused.hip — class-level (un-fused) operators: ViewSynth, PhotoError, and the reduction half of ReconstructionLoss.
//
// These are the drop-ins for callers that hold the intermediate tensors themselves (the reference's `ViewSynth.forward`,
// `PhotoError.forward`, `ReconstructionLoss.forward` on already-warped images).  They share the device
// helpers of the fused kernels but are plain one-thread-per-pixel kernels: the fused path in smd_recon_*.hip is the hot
// one, this file is about API completeness and exact semantics (any channel count, depth_warp / mask_valid outputs,
// gradients to the warped INPUT as feature reconstruction needs them).
#include "smd_common.h"
#include "smd_kernels.h"

namespace smd {

constexpr int kUfBlock = 256;

struct PixGeom { float hx, hy, hz, nx, ny, yz, rz, sx, sy; };

__device__ __forceinline__ PixGeom pix_geom(const Cam& cm, float D, float uf, float vf, float wscale, float hscale) {
  PixGeom g;
  g.hx = fmaf(cm.H[0], uf, fmaf(cm.H[1], vf, cm.H[2]));
  g.hy = fmaf(cm.H[3], uf, fmaf(cm.H[4], vf, cm.H[5]));
  g.hz = fmaf(cm.H[6], uf, fmaf(cm.H[7], vf, cm.H[8]));
  g.nx = fmaf(D, g.hx, cm.a0); g.ny = fmaf(D, g.hy, cm.a1); g.yz = fmaf(D, g.hz, cm.tz);
  g.rz = __builtin_amdgcn_rcpf(fmaxf(g.yz, kZMin));
  g.sx = fmaf(g.nx*g.rz, wscale, -0.5f); g.sy = fmaf(g.ny*g.rz, hscale, -0.5f);
  return g;
}

// ---------------------------------------------------------------------------------------------
// ViewSynth.forward (src/tools/geometry.py:366-391)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kUfBlock) void k_view_synth_fwd(const float* __restrict__ input, const float* __restrict__ depth,
                                                             const float* __restrict__ T, const float* __restrict__ K,
                                                             const float* __restrict__ Kinv, float* __restrict__ warp,
                                                             float* __restrict__ depth_warp, uint8_t* __restrict__ mask_valid,
                                                             int C, int h, int w, float wscale, float hscale) {
  const int bi = blockIdx.y;
  const int pix = blockIdx.x*kUfBlock + threadIdx.x;
  if (pix >= h*w) return;
  Cam cm;
  make_cam(cm, T + (size_t)bi*16, K + (size_t)bi*16, Kinv + (size_t)bi*16);
  const int v = pix/w, u = pix - v*w;
  const size_t hw = (size_t)h*w;
  const PixGeom g = pix_geom(cm, depth[(size_t)bi*hw + pix], (float)u, (float)v, wscale, hscale);
  const Taps tp = make_taps(g.sx, g.sy, h, w, w);
  const float* in_b = input + (size_t)bi*C*hw;
  for (int c = 0; c < C; ++c) warp[((size_t)bi*C + c)*hw + pix] = bilerp(in_b + (size_t)c*hw, tp, w);
  if (depth_warp) depth_warp[(size_t)bi*hw + pix] = fmaxf(g.yz, kEps32);                      // geometry.py:340
  if (mask_valid) {
    const float gx = (g.nx*g.rz/(float)(w - 1) - 0.5f)*2.f, gy = (g.ny*g.rz/(float)(h - 1) - 0.5f)*2.f;   // :347-349
    mask_valid[(size_t)bi*hw + pix] = (fabsf(gx) < 1.f && fabsf(gy) < 1.f) ? 1 : 0;                        // :388
  }
}

hipError_t launch_view_synth_fwd(const float* input, const float* depth, const float* T, const float* K, const float* Kinv,
                                 float* warp, float* depth_warp, uint8_t* mask_valid, int B, int C, int h, int w, hipStream_t st) {
  const float wscale = (float)((double)w/(double)(w - 1)), hscale = (float)((double)h/(double)(h - 1));
  hipLaunchKernelGGL(k_view_synth_fwd, dim3(ceil_div(h*w, kUfBlock), B), dim3(kUfBlock), 0, st, input, depth, T, K, Kinv, warp, depth_warp,
                     mask_valid, C, h, w, wscale, hscale);
  return hipGetLastError();
}

// Backward: dL/d(input) by scattering the four tap weights (atomics), dL/d depth per pixel, and the twelve pose sums
// per block that k_pose_finalize (smd_recon_bwd.hip) turns into dL/dT, dL/dK, dL/dKinv.
__global__ __launch_bounds__(kUfBlock) void k_view_synth_bwd(const float* __restrict__ input, const float* __restrict__ depth,
                                                             const float* __restrict__ T, const float* __restrict__ K,
                                                             const float* __restrict__ Kinv, const float* __restrict__ g_warp,
                                                             const float* __restrict__ g_depth_warp, float* __restrict__ g_input,
                                                             float* __restrict__ g_depth, float* __restrict__ pose_partial,
                                                             int C, int h, int w, float wscale, float hscale) {
  __shared__ float red[kUfBlock/64][kPoseSums];
  const int bi = blockIdx.y;
  const int pix = blockIdx.x*kUfBlock + threadIdx.x;
  const bool live = pix < h*w;
  Cam cm;
  make_cam(cm, T + (size_t)bi*16, K + (size_t)bi*16, Kinv + (size_t)bi*16);
  float ps[kPoseSums] = {};
  if (live) {
    const int v = pix/w, u = pix - v*w;
    const size_t hw = (size_t)h*w;
    const float D = depth[(size_t)bi*hw + pix];
    const float uf = (float)u, vf = (float)v;
    const PixGeom g = pix_geom(cm, D, uf, vf, wscale, hscale);
    const Taps tp = make_taps(g.sx, g.sy, h, w, w);
    const float* in_b = input + (size_t)bi*C*hw;
    float gsx = 0.f, gsy = 0.f;
    for (int c = 0; c < C; ++c) {
      float ddx, ddy;
      (void)bilerp(in_b + (size_t)c*hw, tp, w, ddx, ddy);
      const float go = g_warp[((size_t)bi*C + c)*hw + pix];
      gsx = fmaf(go, ddx, gsx); gsy = fmaf(go, ddy, gsy);
      if (g_input) {
        float* gi = g_input + ((size_t)bi*C + c)*hw + tp.off;
        atomicAdd(gi, go*(1.f - tp.fx)*(1.f - tp.fy)); atomicAdd(gi + 1, go*tp.fx*(1.f - tp.fy));
        atomicAdd(gi + w, go*(1.f - tp.fx)*tp.fy); atomicAdd(gi + w + 1, go*tp.fx*tp.fy);
      }
    }
    const float gpx = gsx*tp.mx*wscale, gpy = gsy*tp.my*hscale;
    const float gnx = gpx*g.rz, gny = gpy*g.rz;
    float gz = (g.yz >= kZMin) ? -(gpx*g.nx + gpy*g.ny)*g.rz*g.rz : 0.f;
    if (g_depth_warp && g.yz >= kEps32) gz += g_depth_warp[(size_t)bi*hw + pix];               // depth_warp = clamp(Yz, eps)
    g_depth[(size_t)bi*hw + pix] = fmaf(gnx, g.hx, fmaf(gny, g.hy, gz*g.hz));
    const float dnx = gnx*D, dny = gny*D, dz = gz*D;
    ps[0] = dnx*uf; ps[1] = dnx*vf; ps[2] = dnx; ps[3] = dny*uf; ps[4] = dny*vf; ps[5] = dny;
    ps[6] = dz*uf; ps[7] = dz*vf; ps[8] = dz; ps[9] = gnx; ps[10] = gny; ps[11] = gz;
  }
#pragma unroll
  for (int k = 0; k < kPoseSums; ++k) {
    const float tot = wave_sum(ps[k]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = tot;
  }
  __syncthreads();
  if (threadIdx.x < kPoseSums) {
    float tot = 0.f;
    for (int wv = 0; wv < kUfBlock/64; ++wv) tot += red[wv][threadIdx.x];
    pose_partial[((size_t)bi*gridDim.x + blockIdx.x)*kPoseSums + threadIdx.x] = tot;
  }
}

hipError_t launch_view_synth_bwd(const float* input, const float* depth, const float* T, const float* K, const float* Kinv,
                                 const float* g_warp, const float* g_depth_warp, float* g_input, float* g_depth,
                                 float* g_T, float* g_K, float* g_Kinv, float* ws, int B, int C, int h, int w, hipStream_t st) {
  const float wscale = (float)((double)w/(double)(w - 1)), hscale = (float)((double)h/(double)(h - 1));
  const int nblk = ceil_div(h*w, kUfBlock);
  if (g_input) { hipError_t e = hipMemsetAsync(g_input, 0, (size_t)B*C*h*w*sizeof(float), st); if (e != hipSuccess) return e; }
  hipLaunchKernelGGL(k_view_synth_bwd, dim3(nblk, B), dim3(kUfBlock), 0, st, input, depth, T, K, Kinv, g_warp, g_depth_warp, g_input, g_depth,
                     ws, C, h, w, wscale, hscale);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_pose_finalize(ws, nblk, nblk, T, K, Kinv, g_T, g_K, g_Kinv, B, 1, st);
}

// ---------------------------------------------------------------------------------------------
// PhotoError(0.85) / DenseL1Error / DenseL2Error for any channel count (src/losses/photometric.py:11-20, 54-88)
// ---------------------------------------------------------------------------------------------
enum { kPhotoSsim = 0, kPhotoL1 = 1, kPhotoL2 = 2 };

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2*(n - 1) - i : i); }

// Nine-tap (reflection padded) un-normalised window sums of one channel at (v, u).
__device__ __forceinline__ void window_sums(const float* __restrict__ x, const float* __restrict__ y, int h, int w, int v, int u,
                                            float& sx, float& sxx, float& sxy, float& sy, float& syy) {
  sx = sxx = sxy = sy = syy = 0.f;
#pragma unroll
  for (int dv = -1; dv <= 1; ++dv) {
    const int rv = reflect1(v + dv, h)*w;
#pragma unroll
    for (int du = -1; du <= 1; ++du) {
      const int idx = rv + reflect1(u + du, w);
      const float a = x[idx], b = y[idx];
      sx += a; sxx = fmaf(a, a, sxx); sxy = fmaf(a, b, sxy); sy += b; syy = fmaf(b, b, syy);
    }
  }
}

__global__ __launch_bounds__(kUfBlock) void k_photo_error_fwd(const float* __restrict__ pred, const float* __restrict__ target,
                                                              float* __restrict__ err, int C, int h, int w, int mode, float w_ssim) {
  const int ni = blockIdx.y;
  const int pix = blockIdx.x*kUfBlock + threadIdx.x;
  if (pix >= h*w) return;
  const int v = pix/w, u = pix - v*w;
  const size_t hw = (size_t)h*w;
  constexpr float c1 = 81.f*kC1, c2 = 81.f*kC2;
  float es = 0.f, el = 0.f, e2 = 0.f;
  for (int c = 0; c < C; ++c) {
    const float* x = pred + ((size_t)ni*C + c)*hw; const float* y = target + ((size_t)ni*C + c)*hw;
    const float d = x[pix] - y[pix];
    el += fabsf(d); e2 = fmaf(d, d, e2);
    if (mode == kPhotoSsim) {
      float sx, sxx, sxy, sy, syy;
      window_sums(x, y, h, w, v, u, sx, sxx, sxy, sy, syy);
      const float t = sx*sy, sx2 = sx*sx;
      const float num = fmaf(2.f, t, c1)*fmaf(2.f, fmaf(9.f, sxy, -t), c2);
      const float den = (sx2 + fmaf(sy, sy, c1))*(fmaf(9.f, sxx, -sx2) + (fmaf(9.f, syy, c2) - sy*sy));
      es += fminf(fmaxf(fmaf(-0.5f, num/den, 0.5f), 0.f), 1.f);
    }
  }
  const float rc = 1.f/(float)C;
  // PhotoError(weight_ssim) (src/losses/photometric.py:85-86): weight_ssim * mean_c SSIM + (1 - weight_ssim) * mean_c |.|
  err[(size_t)ni*hw + pix] = mode == kPhotoL2 ? sqrtf(fmaxf(e2, kEps32)) : (mode == kPhotoL1 ? el*rc : fmaf(w_ssim*rc, es, ((1.f - w_ssim)*rc)*el));
}

static inline int photo_mode(int flags) { return (flags & SMD_LOSS_L2) ? kPhotoL2 : ((flags & SMD_LOSS_L1) ? kPhotoL1 : kPhotoSsim); }

hipError_t launch_photo_error_fwd(const float* pred, const float* target, float* err, int N, int C, int h, int w, int flags, float w_ssim, hipStream_t st) {
  int mode = photo_mode(flags);
  if (mode == kPhotoSsim && w_ssim == 0.f) mode = kPhotoL1;      // PhotoError(0): the SSIM term is not evaluated (photometric.py:72)
  hipLaunchKernelGGL(k_photo_error_fwd, dim3(ceil_div(h*w, kUfBlock), N), dim3(kUfBlock), 0, st, pred, target, err, C, h, w, mode, w_ssim);
  return hipGetLastError();
}

// Backward, pass 1: per pixel p the three SSIM partials (w.r.t. the x9 sums) times the upstream gradient -> coef (N,9,h,w).
__global__ __launch_bounds__(kUfBlock) void k_photo_coef(const float* __restrict__ pred, const float* __restrict__ target,
                                                         const float* __restrict__ g_err, float* __restrict__ coef, int C, int h, int w, float w_ssim) {
  const int ni = blockIdx.y;
  const int pix = blockIdx.x*kUfBlock + threadIdx.x;
  if (pix >= h*w) return;
  const int v = pix/w, u = pix - v*w;
  const size_t hw = (size_t)h*w;
  constexpr float c1 = 81.f*kC1, c2 = 81.f*kC2;
  const float g = g_err[(size_t)ni*hw + pix]*(w_ssim/(float)C);
  for (int c = 0; c < C; ++c) {
    const float* x = pred + ((size_t)ni*C + c)*hw; const float* y = target + ((size_t)ni*C + c)*hw;
    float sx, sxx, sxy, sy, syy;
    window_sums(x, y, h, w, v, u, sx, sxx, sxy, sy, syy);
    const float t = sx*sy, sx2 = sx*sx;
    const float a1 = fmaf(2.f, t, c1), a2 = fmaf(2.f, fmaf(9.f, sxy, -t), c2);
    const float b1 = sx2 + fmaf(sy, sy, c1), b2 = fmaf(9.f, sxx, -sx2) + (fmaf(9.f, syy, c2) - sy*sy);
    const float rden = 1.f/(b1*b2), val = a1*a2*rden, e = fmaf(-0.5f, val, 0.5f);
    const float prd = ((e >= 0.f && e <= 1.f) ? -0.5f*g : 0.f)*rden;
    float* cp = coef + ((size_t)ni*3*C + c*3)*hw + pix;
    cp[0] = prd*(2.f*sy*(a2 - a1) - 2.f*sx*val*(b2 - b1));
    cp[hw] = prd*(-9.f*val*b1);
    cp[2*hw] = prd*(18.f*a1);
  }
}

// Backward, pass 2: adjoint of (reflection pad + 3x3 sum) applied to the coefficient maps, plus the L1 term.
__global__ __launch_bounds__(kUfBlock) void k_photo_error_bwd(const float* __restrict__ pred, const float* __restrict__ target,
                                                              const float* __restrict__ g_err, const float* __restrict__ coef,
                                                              float* __restrict__ g_pred, int C, int h, int w, int mode, float w_ssim) {
  const int ni = blockIdx.y;
  const int pix = blockIdx.x*kUfBlock + threadIdx.x;
  if (pix >= h*w) return;
  const int v = pix/w, u = pix - v*w;
  const size_t hw = (size_t)h*w;
  const float ge = g_err[(size_t)ni*hw + pix];
  if (mode == kPhotoL2) {   // sqrt(clamp(sum d^2, eps)): zero gradient where the clamp is active
    float e2 = 0.f;
    for (int c = 0; c < C; ++c) { const float d = pred[((size_t)ni*C + c)*hw + pix] - target[((size_t)ni*C + c)*hw + pix]; e2 = fmaf(d, d, e2); }
    const float k = (e2 >= kEps32) ? ge/sqrtf(e2) : 0.f;
    for (int c = 0; c < C; ++c) g_pred[((size_t)ni*C + c)*hw + pix] = k*(pred[((size_t)ni*C + c)*hw + pix] - target[((size_t)ni*C + c)*hw + pix]);
    return;
  }
  const float gl = ge*(mode == kPhotoL1 ? 1.f : (1.f - w_ssim))/(float)C;
  float wv[3], wu[3];
  reflect_weights_adj(v, h, wv[0], wv[2]); wv[1] = 1.f;
  reflect_weights_adj(u, w, wu[0], wu[2]); wu[1] = 1.f;
  for (int c = 0; c < C; ++c) {
    const float x = pred[((size_t)ni*C + c)*hw + pix], y = target[((size_t)ni*C + c)*hw + pix];
    const float d = x - y;
    float gx = gl*((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f));
    if (mode == kPhotoSsim) {
      const float* cp = coef + ((size_t)ni*3*C + c*3)*hw;
      float SA = 0.f, SB = 0.f, SC = 0.f;
#pragma unroll
      for (int dv = -1; dv <= 1; ++dv) {
        const int pv = v + dv;
        if (pv < 0 || pv >= h) continue;
#pragma unroll
        for (int du = -1; du <= 1; ++du) {
          const int pu = u + du;
          if (pu < 0 || pu >= w) continue;
          const float wt = wv[dv + 1]*wu[du + 1];
          const int idx = pv*w + pu;
          SA = fmaf(wt, cp[idx], SA); SB = fmaf(wt, cp[hw + idx], SB); SC = fmaf(wt, cp[2*hw + idx], SC);
        }
      }
      gx += fmaf(2.f*x, SB, fmaf(y, SC, SA));
    }
    g_pred[((size_t)ni*C + c)*hw + pix] = gx;
  }
}

hipError_t launch_photo_error_bwd(const float* pred, const float* target, const float* g_err, float* g_pred, float* ws,
                                  int N, int C, int h, int w, int flags, float w_ssim, hipStream_t st) {
  int mode = photo_mode(flags);
  if (mode == kPhotoSsim && w_ssim == 0.f) mode = kPhotoL1;
  dim3 grid(ceil_div(h*w, kUfBlock), N);
  if (mode == kPhotoSsim) hipLaunchKernelGGL(k_photo_coef, grid, dim3(kUfBlock), 0, st, pred, target, g_err, ws, C, h, w, w_ssim);
  hipLaunchKernelGGL(k_photo_error_bwd, grid, dim3(kUfBlock), 0, st, pred, target, g_err, ws, g_pred, C, h, w, mode, w_ssim);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Reduction half of ReconstructionLoss.forward (src/losses/reconstruction.py:43-44, 59-77, 125)
// ---------------------------------------------------------------------------------------------
// Predictive weighting masks of `ReconstructionLoss.apply_mask` (src/losses/reconstruction.py:46-57): mask (B,n,h,w) in the
// reference's layout, one channel per support.  mode 1 'explainability': err * mask; mode 2 'uncertainty': err * exp(-mask) + mask.
__device__ __forceinline__ float masked_err(float e, float m, int mode) { return mode == 1 ? e*m : (mode == 2 ? fmaf(e, __expf(-m), m) : e); }

__global__ __launch_bounds__(kUfBlock) void k_recon_reduce_fwd(const float* __restrict__ err_warp, const float* __restrict__ err_static,
                                                               const float* __restrict__ noise, uint32_t seed_lo, uint32_t seed_hi,
                                                               float* __restrict__ err, uint8_t* __restrict__ sel, float* __restrict__ partial,
                                                               int n, unsigned total, int use_min, const float* __restrict__ mask, int mask_mode, unsigned hw) {
  __shared__ float red[kUfBlock/64];
  const unsigned idx = blockIdx.x*kUfBlock + threadIdx.x;
  float e = 0.f;
  if (idx < total) {
    const unsigned bi = idx/hw, pix = idx - bi*hw;
    auto mk = [&](int i) { return mask_mode ? mask[((size_t)bi*n + i)*hw + pix] : 0.f; };
    float best = masked_err(err_warp[idx], mk(0), mask_mode), acc = best;
    int bsel = 0;
    for (int i = 1; i < n; ++i) { const float v = masked_err(err_warp[(size_t)i*total + idx], mk(i), mask_mode); acc += v; if (v < best) { best = v; bsel = i; } }
    e = use_min ? best : acc/(float)n;
    if (!use_min) bsel = 0;
    if (err_static) {
      float sb = masked_err(err_static[idx], mk(0), mask_mode), sa = sb;
      for (int i = 1; i < n; ++i) { const float v = masked_err(err_static[(size_t)i*total + idx], mk(i), mask_mode); sa += v; sb = fminf(sb, v); }
      float est = use_min ? sb : sa/(float)n;
      est = fmaf(kEps32, noise ? noise[idx] : gauss_noise(seed_lo, seed_hi, idx), est);
      if (est < e) { e = est; bsel = SMD_SEL_MASKED; }
    }
    err[idx] = e; sel[idx] = (uint8_t)bsel;
  }
  const float tot = wave_sum(e);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = tot;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

static inline int mask_mode_of(int flags, const float* mask) { return !mask ? 0 : ((flags & SMD_MASK_UNCERTAINTY) ? 2 : ((flags & SMD_MASK_EXPLAINABILITY) ? 1 : 0)); }

hipError_t launch_recon_reduce_fwd(const float* err_warp, const float* err_static, const float* mask, const float* noise, uint64_t seed,
                                   float* err, uint8_t* sel, float* loss, float* ws, int n, int B, int h, int w, int flags, hipStream_t st) {
  const unsigned total = (unsigned)B*h*w;
  const int nblk = ceil_div((int)total, kUfBlock);
  hipLaunchKernelGGL(k_recon_reduce_fwd, dim3(nblk), dim3(kUfBlock), 0, st, err_warp, (flags & SMD_USE_AUTOMASK) ? err_static : nullptr, noise,
                     (uint32_t)seed, (uint32_t)(seed >> 32), err, sel, ws, n, total, (flags & SMD_USE_MIN) ? 1 : 0, mask, mask_mode_of(flags, mask), (unsigned)(h*w));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_sum_partials(ws, nblk, 1.0/(double)total, loss, st);
}

// Backward.  Without a mask: the upstream gradient routed by `sel`.  With a mask the error that reached the loss was
// f(err, m) = err*m | err*exp(-m) + m, so the routed gradient is scaled by df/derr and the mask receives df/dm — also where the
// STATIC error won (sel == masked): the identity error carries no gradient to the images but it does to the mask
// (reconstruction.py:70-71 evaluates `compute_photo(source, target, mask=mask)`); its winning support is re-derived here.
__global__ __launch_bounds__(kUfBlock) void k_recon_reduce_bwd(const uint8_t* __restrict__ sel, const float* __restrict__ g_loss,
                                                               float* __restrict__ g_err_warp, int n, unsigned total, int use_min,
                                                               const float* __restrict__ err_warp, const float* __restrict__ err_static,
                                                               const float* __restrict__ mask, float* __restrict__ g_mask, int mask_mode, unsigned hw) {
  const unsigned idx = blockIdx.x*kUfBlock + threadIdx.x;
  if (idx >= total) return;
  const float g = g_loss[0]/(float)total;
  const uint8_t s = sel[idx];
  const unsigned bi = idx/hw, pix = idx - bi*hw;
  int jstat = -1;                       // static winner (min-reprojection over the masked identity errors), when the automask removed the pixel
  if (mask_mode && s == (uint8_t)SMD_SEL_MASKED && use_min && err_static) {
    float sb = 0.f;
    for (int i = 0; i < n; ++i) {
      const float v = masked_err(err_static[(size_t)i*total + idx], mask[((size_t)bi*n + i)*hw + pix], mask_mode);
      if (i == 0 || v < sb) { sb = v; jstat = i; }
    }
  }
  for (int i = 0; i < n; ++i) {
    float v = 0.f;
    if (use_min) v = (s == (uint8_t)i) ? g : 0.f;
    else v = (s != (uint8_t)SMD_SEL_MASKED) ? g/(float)n : 0.f;
    if (mask_mode) {
      const size_t mi = ((size_t)bi*n + i)*hw + pix;
      const float m = mask[mi];
      float gs = 0.f;                   // gradient reaching the masked STATIC error of support i
      if (s == (uint8_t)SMD_SEL_MASKED && err_static) gs = use_min ? (i == jstat ? g : 0.f) : g/(float)n;
      const float ew = err_warp[(size_t)i*total + idx], es = err_static ? err_static[(size_t)i*total + idx] : 0.f;
      if (mask_mode == 1) { g_mask[mi] = v*ew + gs*es; v *= m; }
      else { const float em = __expf(-m); g_mask[mi] = v*fmaf(-ew, em, 1.f) + gs*fmaf(-es, em, 1.f); v *= em; }
    }
    g_err_warp[(size_t)i*total + idx] = v;
  }
}

hipError_t launch_recon_reduce_bwd(const uint8_t* sel, const float* g_loss, float* g_err_warp, const float* err_warp, const float* err_static,
                                   const float* mask, float* g_mask, int n, int B, int h, int w, int flags, hipStream_t st) {
  const unsigned total = (unsigned)B*h*w;
  hipLaunchKernelGGL(k_recon_reduce_bwd, dim3(ceil_div((int)total, kUfBlock)), dim3(kUfBlock), 0, st, sel, g_loss, g_err_warp, n, total,
                     (flags & SMD_USE_MIN) ? 1 : 0, err_warp, (flags & SMD_USE_AUTOMASK) ? err_static : nullptr, mask, g_mask, mask_mode_of(flags, mask), (unsigned)(h*w));
  return hipGetLastError();
}

}  // namespace smd
