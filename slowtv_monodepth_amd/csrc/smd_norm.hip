// smd_norm.hip — training-mode BatchNorm2d fused with the residual add and ReLU that follow it in the ResNet encoders
// (timm `resnet18/34/50`, built at src/networks/depth.py:95-98 and src/networks/pose.py:39-41: conv -> BN -> ReLU and
// conv -> BN -> (+identity) -> ReLU).  The producer side of the loss path: after the loss kernels, BatchNorm and the
// element-wise ops around it were the largest non-convolution item of the step (MIOpen's spatial BN kernels launch one
// workgroup per channel: 64 workgroups on a 256-CU device for the stem).
//
//   forward : stats sweep (shifted sums, many blocks per channel) -> finalize (fp64 combine, running stats) -> apply sweep
//   backward: reduce sweep (sum dz, sum dz*(x-mean)) -> finalize (g_gamma, g_beta, coefficients) -> apply sweep
// where dz = g_y masked by (y > 0) when the ReLU is fused.  NCHW fp32; float4 path when HW % 4 == 0.
#include "smd_common.h"
#include "smd_kernels.h"

namespace smd {

constexpr int kBnBlock = 256;
constexpr int kBnItemsPerBlock = 8192;   // floats a block sweeps in the reduction kernels
constexpr int kBnMaxChunks = 128;

int bn_chunks(int N, int HW) {
  const long long total = (long long)N*HW;
  long long c = (total + kBnItemsPerBlock - 1)/kBnItemsPerBlock;
  return (int)(c < 1 ? 1 : (c > kBnMaxChunks ? kBnMaxChunks : c));
}

__device__ __forceinline__ void block_sum2(float a, float b, float* red, float& ra, float& rb) {
  a = wave_sum(a); b = wave_sum(b);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wv*2] = a; red[wv*2 + 1] = b; }
  __syncthreads();
  ra = 0.f; rb = 0.f;
#pragma unroll
  for (int k = 0; k < kBnBlock/64; ++k) { ra += red[k*2]; rb += red[k*2 + 1]; }
}

// Range of the flattened (n, p) index space of one channel swept by chunk k (multiple of 4 when VEC).
__device__ __forceinline__ void chunk_range(long long total, int chunks, int k, long long& lo, long long& hi) {
  long long len = (total + chunks - 1)/chunks;
  len = (len + 3) & ~3ll;
  lo = (long long)k*len; hi = lo + len < total ? lo + len : total;
}

template <bool VEC>
__global__ __launch_bounds__(kBnBlock) void k_bn_stats(const float* __restrict__ x, int N, int C, int HW, int chunks, float* __restrict__ partial) {
  __shared__ float red[2*kBnBlock/64];
  const int c = blockIdx.y, k = blockIdx.x;
  const long long total = (long long)N*HW;
  long long lo, hi;
  chunk_range(total, chunks, k, lo, hi);
  const float shift = x[(size_t)c*HW];   // first element of the channel: sums of (x - shift) stay well conditioned
  float s1 = 0.f, s2 = 0.f;
  if (VEC) {
    for (long long i = lo + (long long)threadIdx.x*4; i < hi; i += kBnBlock*4) {
      const int n = (int)(i/HW), p = (int)(i - (long long)n*HW);
      const f4 v = *(const f4*)(x + ((size_t)n*C + c)*HW + p);
#pragma unroll
      for (int q = 0; q < 4; ++q) { const float d = v[q] - shift; s1 += d; s2 = fmaf(d, d, s2); }
    }
  } else {
    for (long long i = lo + threadIdx.x; i < hi; i += kBnBlock) {
      const int n = (int)(i/HW), p = (int)(i - (long long)n*HW);
      const float d = x[((size_t)n*C + c)*HW + p] - shift; s1 += d; s2 = fmaf(d, d, s2);
    }
  }
  float r1, r2;
  block_sum2(s1, s2, red, r1, r2);
  if (threadIdx.x == 0) { partial[((size_t)c*chunks + k)*2] = r1; partial[((size_t)c*chunks + k)*2 + 1] = r2; }
}

__global__ __launch_bounds__(64) void k_bn_finalize(const float* __restrict__ x, const float* __restrict__ partial, int N, int C, int HW, int chunks,
                                                    float momentum, float eps, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                    float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  const int c = blockIdx.x*64 + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < chunks; ++k) { s1 += (double)partial[((size_t)c*chunks + k)*2]; s2 += (double)partial[((size_t)c*chunks + k)*2 + 1]; }
  const double M = (double)N*HW, m1 = s1/M;
  const double mean = (double)x[(size_t)c*HW] + m1;
  double var = s2/M - m1*m1;
  if (var < 0.0) var = 0.0;
  save_mean[c] = (float)mean;
  save_invstd[c] = (float)(1.0/sqrt(var + (double)eps));
  if (running_mean) running_mean[c] = (float)((1.0 - momentum)*(double)running_mean[c] + momentum*mean);
  if (running_var) running_var[c] = (float)((1.0 - momentum)*(double)running_var[c] + momentum*var*(M > 1.0 ? M/(M - 1.0) : 1.0));
}

template <bool VEC>
__global__ __launch_bounds__(kBnBlock) void k_bn_apply(const float* __restrict__ x, const float* __restrict__ residual, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       int C, int HW, size_t total, int relu, float* __restrict__ y) {
  constexpr int W = VEC ? 4 : 1;
  for (size_t e = ((size_t)blockIdx.x*kBnBlock + threadIdx.x)*W; e < total; e += (size_t)gridDim.x*kBnBlock*W) {
    const int c = (int)((e/HW) % C);
    const float sc = gamma[c]*invstd[c], sh = fmaf(-mean[c], sc, beta[c]);
    if (VEC) {
      f4 v = *(const f4*)(x + e);
      f4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = fmaf(v[q], sc, sh);
      if (residual) { const f4 r = *(const f4*)(residual + e); o += r; }
      if (relu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = fmaxf(o[q], 0.f);
      }
      *(f4*)(y + e) = o;
    } else {
      float o = fmaf(x[e], sc, sh);
      if (residual) o += residual[e];
      y[e] = relu ? fmaxf(o, 0.f) : o;
    }
  }
}

template <bool VEC>
__global__ __launch_bounds__(kBnBlock) void k_bn_bwd_reduce(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g_y,
                                                            const float* __restrict__ mean, int N, int C, int HW, int chunks, int relu,
                                                            float* __restrict__ partial) {
  __shared__ float red[2*kBnBlock/64];
  const int c = blockIdx.y, k = blockIdx.x;
  const long long total = (long long)N*HW;
  long long lo, hi;
  chunk_range(total, chunks, k, lo, hi);
  const float mu = mean[c];
  float s1 = 0.f, s2 = 0.f;
  if (VEC) {
    for (long long i = lo + (long long)threadIdx.x*4; i < hi; i += kBnBlock*4) {
      const int n = (int)(i/HW), p = (int)(i - (long long)n*HW);
      const size_t off = ((size_t)n*C + c)*HW + p;
      const f4 xv = *(const f4*)(x + off), gv = *(const f4*)(g_y + off);
      f4 yv = {1.f, 1.f, 1.f, 1.f};
      if (relu) yv = *(const f4*)(y + off);
#pragma unroll
      for (int q = 0; q < 4; ++q) { const float dz = (yv[q] > 0.f) ? gv[q] : 0.f; s1 += dz; s2 = fmaf(dz, xv[q] - mu, s2); }
    }
  } else {
    for (long long i = lo + threadIdx.x; i < hi; i += kBnBlock) {
      const int n = (int)(i/HW), p = (int)(i - (long long)n*HW);
      const size_t off = ((size_t)n*C + c)*HW + p;
      const float dz = (!relu || y[off] > 0.f) ? g_y[off] : 0.f;
      s1 += dz; s2 = fmaf(dz, x[off] - mu, s2);
    }
  }
  float r1, r2;
  block_sum2(s1, s2, red, r1, r2);
  if (threadIdx.x == 0) { partial[((size_t)c*chunks + k)*2] = r1; partial[((size_t)c*chunks + k)*2 + 1] = r2; }
}

// coef[c] = {gamma*invstd, sum(dz)/M, invstd^2 * sum(dz*(x-mean))/M}
__global__ __launch_bounds__(64) void k_bn_bwd_finalize(const float* __restrict__ partial, const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                        int N, int C, int HW, int chunks, float* __restrict__ g_gamma, float* __restrict__ g_beta,
                                                        float* __restrict__ coef) {
  const int c = blockIdx.x*64 + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < chunks; ++k) { s1 += (double)partial[((size_t)c*chunks + k)*2]; s2 += (double)partial[((size_t)c*chunks + k)*2 + 1]; }
  const double M = (double)N*HW, is = (double)invstd[c];
  g_beta[c] = (float)s1; g_gamma[c] = (float)(s2*is);
  coef[c*3] = (float)((double)gamma[c]*is); coef[c*3 + 1] = (float)(s1/M); coef[c*3 + 2] = (float)(s2*is*is/M);
}

template <bool VEC>
__global__ __launch_bounds__(kBnBlock) void k_bn_bwd_apply(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g_y,
                                                           const float* __restrict__ mean, const float* __restrict__ coef, int C, int HW, size_t total,
                                                           int relu, float* __restrict__ g_x, float* __restrict__ g_res) {
  constexpr int W = VEC ? 4 : 1;
  for (size_t e = ((size_t)blockIdx.x*kBnBlock + threadIdx.x)*W; e < total; e += (size_t)gridDim.x*kBnBlock*W) {
    const int c = (int)((e/HW) % C);
    const float a = coef[c*3], b = coef[c*3 + 1], k2 = coef[c*3 + 2], mu = mean[c];
    if (VEC) {
      const f4 xv = *(const f4*)(x + e), gv = *(const f4*)(g_y + e);
      f4 yv = {1.f, 1.f, 1.f, 1.f};
      if (relu) yv = *(const f4*)(y + e);
      f4 dz, dx;
#pragma unroll
      for (int q = 0; q < 4; ++q) { dz[q] = (yv[q] > 0.f) ? gv[q] : 0.f; dx[q] = a*(dz[q] - b - (xv[q] - mu)*k2); }
      *(f4*)(g_x + e) = dx;
      if (g_res) *(f4*)(g_res + e) = dz;
    } else {
      const float dz = (!relu || y[e] > 0.f) ? g_y[e] : 0.f;
      g_x[e] = a*(dz - b - (x[e] - mu)*k2);
      if (g_res) g_res[e] = dz;
    }
  }
}

hipError_t launch_bn_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float* running_mean, float* running_var,
                         float momentum, float eps, int relu, float* y, float* save_mean, float* save_invstd, float* ws, int N, int C, int HW,
                         hipStream_t st) {
  const int chunks = bn_chunks(N, HW);
  const bool vec = (HW % 4) == 0;
  const size_t total = (size_t)N*C*HW;
  const unsigned ngrid = (unsigned)((total/(vec ? 4 : 1) + kBnBlock - 1)/kBnBlock < 16384 ? (total/(vec ? 4 : 1) + kBnBlock - 1)/kBnBlock : 16384);
  if (vec) hipLaunchKernelGGL(k_bn_stats<true>, dim3(chunks, C), dim3(kBnBlock), 0, st, x, N, C, HW, chunks, ws);
  else hipLaunchKernelGGL(k_bn_stats<false>, dim3(chunks, C), dim3(kBnBlock), 0, st, x, N, C, HW, chunks, ws);
  hipLaunchKernelGGL(k_bn_finalize, dim3(ceil_div(C, 64)), dim3(64), 0, st, x, ws, N, C, HW, chunks, momentum, eps, running_mean, running_var,
                     save_mean, save_invstd);
  if (vec) hipLaunchKernelGGL(k_bn_apply<true>, dim3(ngrid), dim3(kBnBlock), 0, st, x, residual, gamma, beta, save_mean, save_invstd, C, HW, total, relu, y);
  else hipLaunchKernelGGL(k_bn_apply<false>, dim3(ngrid), dim3(kBnBlock), 0, st, x, residual, gamma, beta, save_mean, save_invstd, C, HW, total, relu, y);
  return hipGetLastError();
}

hipError_t launch_bn_bwd(const float* x, const float* y, const float* g_y, const float* gamma, const float* save_mean, const float* save_invstd,
                         int relu, float* g_x, float* g_res, float* g_gamma, float* g_beta, float* ws, int N, int C, int HW, hipStream_t st) {
  const int chunks = bn_chunks(N, HW);
  const bool vec = (HW % 4) == 0;
  const size_t total = (size_t)N*C*HW;
  const unsigned ngrid = (unsigned)((total/(vec ? 4 : 1) + kBnBlock - 1)/kBnBlock < 16384 ? (total/(vec ? 4 : 1) + kBnBlock - 1)/kBnBlock : 16384);
  float* coef = ws + (size_t)C*chunks*2;
  if (vec) hipLaunchKernelGGL(k_bn_bwd_reduce<true>, dim3(chunks, C), dim3(kBnBlock), 0, st, x, y, g_y, save_mean, N, C, HW, chunks, relu, ws);
  else hipLaunchKernelGGL(k_bn_bwd_reduce<false>, dim3(chunks, C), dim3(kBnBlock), 0, st, x, y, g_y, save_mean, N, C, HW, chunks, relu, ws);
  hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(ceil_div(C, 64)), dim3(64), 0, st, ws, gamma, save_invstd, N, C, HW, chunks, g_gamma, g_beta, coef);
  if (vec) hipLaunchKernelGGL(k_bn_bwd_apply<true>, dim3(ngrid), dim3(kBnBlock), 0, st, x, y, g_y, save_mean, coef, C, HW, total, relu, g_x, g_res);
  else hipLaunchKernelGGL(k_bn_bwd_apply<false>, dim3(ngrid), dim3(kBnBlock), 0, st, x, y, g_y, save_mean, coef, C, HW, total, relu, g_x, g_res);
  return hipGetLastError();
}

}  // namespace smd
