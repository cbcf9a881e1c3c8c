// smd_pose.hip — pose / intrinsics prologue of the loss path as two tiny kernels with analytic adjoints.
//
// Replaces the ~45 eager launches per step of `T_from_AAt` (src/tools/geometry.py:181-209), `T.inverse()` for
// backward-in-time supports (src/core/trainer.py:253), `PoseNet.build_K` + `resize_K` (src/networks/pose.py:60-73,
// geometry.py:249-263) and `K.inverse()` (geometry.py:383, a batched LU in the reference).  One thread per matrix.
//   * Rodrigues: R = I + sin(th) W + (1 - cos(th)) W^2,  W = skew(aa / clip(|aa|, eps))
//   * inverted pose: the inverse of a rigid [R t; 0 1] is [R^T, -R^T t]; its adjoint restricted to rotations equals
//     the adjoint of the general inverse the reference differentiates (both act on tangent directions R*skew only)
//   * pinhole K = [[fx w, 0, cx w], [0, fy h, cy h], [0, 0, 1]] and its closed-form inverse; for a caller-supplied K
//     the 3x3 block is inverted by its adjugate.
#include "smd_common.h"
#include "smd_kernels.h"
#include "smd_pose_dev.h"

namespace smd {

__global__ void k_pose_fwd(const float* __restrict__ aa, const float* __restrict__ t, const uint8_t* __restrict__ invert, int N, float* __restrict__ T) {
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float a[3] = {aa[i*3], aa[i*3 + 1], aa[i*3 + 2]}, tv[3] = {t[i*3], t[i*3 + 1], t[i*3 + 2]};
  float R[9], th, c, n[3], s, k;
  rodrigues(a, R, th, c, n, s, k);
  float* o = T + (size_t)i*16;
  if (invert && invert[i]) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int q = 0; q < 3; ++q) o[r*4 + q] = R[q*3 + r];
      o[r*4 + 3] = -(R[r]*tv[0] + R[3 + r]*tv[1] + R[6 + r]*tv[2]);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int q = 0; q < 3; ++q) o[r*4 + q] = R[r*3 + q];
      o[r*4 + 3] = tv[r];
    }
  }
  o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
}

__global__ void k_pose_bwd(const float* __restrict__ aa, const float* __restrict__ t, const uint8_t* __restrict__ invert, int N,
                           const float* __restrict__ g_T, float* __restrict__ g_aa, float* __restrict__ g_t) {
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i < N) pose_bwd_one(aa, t, invert, i, g_T + (size_t)i*16, g_aa, g_t);
}

// mode 0: K (b,4,4) given -> Kinv;  mode 1: (fs, cs) normalised -> K resized to (h, w) and Kinv
__global__ void k_intrinsics_fwd(const float* __restrict__ fs, const float* __restrict__ cs, const float* __restrict__ Kin, int b, int h, int w,
                                 float* __restrict__ K, float* __restrict__ Kinv) {
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= b) return;
  float* ki = Kinv + (size_t)i*16;
  for (int q = 0; q < 16; ++q) ki[q] = (q % 5 == 0) ? 1.f : 0.f;
  if (fs) {
    const float F = fs[i*2]*(float)w, G = fs[i*2 + 1]*(float)h, C = cs[i*2]*(float)w, D = cs[i*2 + 1]*(float)h;
    float* ko = K + (size_t)i*16;
    for (int q = 0; q < 16; ++q) ko[q] = (q % 5 == 0) ? 1.f : 0.f;
    ko[0] = F; ko[2] = C; ko[5] = G; ko[6] = D;
    ki[0] = 1.f/F; ki[2] = -C/F; ki[5] = 1.f/G; ki[6] = -D/G;
  } else {
    const float* m = Kin + (size_t)i*16;
    const float a = m[0], bb = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], hh = m[9], k = m[10];
    const float A = e*k - f*hh, B = -(d*k - f*g), Cc = d*hh - e*g;
    const float idet = 1.f/(a*A + bb*B + c*Cc);
    ki[0] = A*idet; ki[1] = -(bb*k - c*hh)*idet; ki[2] = (bb*f - c*e)*idet;
    ki[4] = B*idet; ki[5] = (a*k - c*g)*idet;    ki[6] = -(a*f - c*d)*idet;
    ki[8] = Cc*idet; ki[9] = -(a*hh - bb*g)*idet; ki[10] = (a*e - bb*d)*idet;
  }
}

__global__ void k_intrinsics_bwd(const float* __restrict__ fs, const float* __restrict__ cs, int b, int h, int w,
                                 const float* __restrict__ g_K, const float* __restrict__ g_Kinv, float* __restrict__ g_fs, float* __restrict__ g_cs) {
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i < b) intrinsics_bwd_one(fs, cs, i, h, w, g_K + (size_t)i*16, g_Kinv + (size_t)i*16, g_fs, g_cs);
}

hipError_t launch_pose_fwd(const float* aa, const float* t, const uint8_t* invert, int N, float* T, hipStream_t st) {
  hipLaunchKernelGGL(k_pose_fwd, dim3(ceil_div(N, 64)), dim3(64), 0, st, aa, t, invert, N, T);
  return hipGetLastError();
}
hipError_t launch_pose_bwd(const float* aa, const float* t, const uint8_t* invert, int N, const float* g_T, float* g_aa, float* g_t, hipStream_t st) {
  hipLaunchKernelGGL(k_pose_bwd, dim3(ceil_div(N, 64)), dim3(64), 0, st, aa, t, invert, N, g_T, g_aa, g_t);
  return hipGetLastError();
}
hipError_t launch_intrinsics_fwd(const float* fs, const float* cs, const float* Kin, int b, int h, int w, float* K, float* Kinv, hipStream_t st) {
  hipLaunchKernelGGL(k_intrinsics_fwd, dim3(ceil_div(b, 64)), dim3(64), 0, st, fs, cs, Kin, b, h, w, K, Kinv);
  return hipGetLastError();
}
hipError_t launch_intrinsics_bwd(const float* fs, const float* cs, int b, int h, int w, const float* g_K, const float* g_Kinv,
                                 float* g_fs, float* g_cs, hipStream_t st) {
  hipLaunchKernelGGL(k_intrinsics_bwd, dim3(ceil_div(b, 64)), dim3(64), 0, st, fs, cs, b, h, w, g_K, g_Kinv, g_fs, g_cs);
  return hipGetLastError();
}

}  // namespace smd
