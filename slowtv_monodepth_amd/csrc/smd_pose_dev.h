// smd_pose_dev.h — device-side bodies of the pose / intrinsics adjoints, shared between the kernels of smd_pose.hip and the guest epilogue of
// the fused loss path (smd_depth.hip), where the chain rule runs on from dL/dT, dL/dK, dL/dK^-1 to the pose network's outputs in the same wave.
#pragma once
#include "smd_common.h"
#include "smd_kernels.h"

namespace smd {

__device__ __forceinline__ void rodrigues(const float a[3], float R[9], float& th, float& c, float n[3], float& s, float& k) {
  th = sqrtf(a[0]*a[0] + a[1]*a[1] + a[2]*a[2]);
  c = fmaxf(th, kEps32);
  n[0] = a[0]/c; n[1] = a[1]/c; n[2] = a[2]/c;
  s = sinf(th); k = 1.f - cosf(th);
  const float W[9] = {0.f, -n[2], n[1], n[2], 0.f, -n[0], -n[1], n[0], 0.f};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float w2 = W[i*3]*W[j] + W[i*3 + 1]*W[3 + j] + W[i*3 + 2]*W[6 + j];
      R[i*3 + j] = ((i == j) ? 1.f : 0.f) + s*W[i*3 + j] + k*w2;
    }
}

// Adjoint of matrix i: g = dL/dT_i (16 floats, row-major; any address space) -> g_aa[3i..], g_t[3i..].  One lane.
inline __device__ void pose_bwd_one(const float* __restrict__ aa, const float* __restrict__ t, const uint8_t* __restrict__ invert, int i,
                                    const float* g, float* __restrict__ g_aa, float* __restrict__ g_t) {
  const float a[3] = {aa[i*3], aa[i*3 + 1], aa[i*3 + 2]}, tv[3] = {t[i*3], t[i*3 + 1], t[i*3 + 2]};
  float R[9], th, c, n[3], s, k;
  rodrigues(a, R, th, c, n, s, k);
  float G[9], gt[3];   // dL/dR, dL/dt
  if (invert && invert[i]) {
    const float u[3] = {g[3], g[7], g[11]};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int q = 0; q < 3; ++q) G[r*3 + q] = g[q*4 + r] - tv[r]*u[q];    // (G'_R)^T - t u^T
      gt[r] = -(R[r*3]*u[0] + R[r*3 + 1]*u[1] + R[r*3 + 2]*u[2]);         // -R u
    }
  } else {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int q = 0; q < 3; ++q) G[r*3 + q] = g[r*4 + q];
      gt[r] = g[r*4 + 3];
    }
  }
  const float W[9] = {0.f, -n[2], n[1], n[2], 0.f, -n[0], -n[1], n[0], 0.f};
  float W2[9], gs = 0.f, gk = 0.f;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      W2[r*3 + q] = W[r*3]*W[q] + W[r*3 + 1]*W[3 + q] + W[r*3 + 2]*W[6 + q];
      gs += G[r*3 + q]*W[r*3 + q]; gk += G[r*3 + q]*W2[r*3 + q];
    }
  float gW[9];   // s G + k (G W^T + W^T G)
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      float gwt = 0.f, wtg = 0.f;
#pragma unroll
      for (int m = 0; m < 3; ++m) { gwt += G[r*3 + m]*W[q*3 + m]; wtg += W[m*3 + r]*G[m*3 + q]; }
      gW[r*3 + q] = s*G[r*3 + q] + k*(gwt + wtg);
    }
  const float gn[3] = {gW[7] - gW[5], gW[2] - gW[6], gW[3] - gW[1]};
  const float inv_th = (th > 0.f) ? 1.f/th : 0.f;
  const float gth = gs*cosf(th) + gk*sinf(th);
  const float gna = gn[0]*a[0] + gn[1]*a[1] + gn[2]*a[2];
  const float clip_pass = (th >= kEps32) ? 1.f : 0.f;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    g_aa[i*3 + m] = gn[m]/c - clip_pass*gna/(c*c)*a[m]*inv_th + gth*a[m]*inv_th;
    g_t[i*3 + m] = gt[m];
  }
}

// Adjoint of sample i's pinhole K / K^-1: gk = dL/dK_i, gi = dL/dK_i^-1 (16 floats each, any address space) -> g_fs[2i..], g_cs[2i..].  One lane.
inline __device__ void intrinsics_bwd_one(const float* __restrict__ fs, const float* __restrict__ cs, int i, int h, int w,
                                          const float* gk, const float* gi, float* __restrict__ g_fs, float* __restrict__ g_cs) {
  const float F = fs[i*2]*(float)w, G = fs[i*2 + 1]*(float)h, C = cs[i*2]*(float)w, D = cs[i*2 + 1]*(float)h;
  const float gF = gk[0] - gi[0]/(F*F) + gi[2]*C/(F*F), gC = gk[2] - gi[2]/F;
  const float gG = gk[5] - gi[5]/(G*G) + gi[6]*D/(G*G), gD = gk[6] - gi[6]/G;
  g_fs[i*2] = gF*(float)w; g_fs[i*2 + 1] = gG*(float)h; g_cs[i*2] = gC*(float)w; g_cs[i*2 + 1] = gD*(float)h;
}

}  // namespace smd
