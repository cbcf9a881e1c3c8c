// smd_norm.hip — training-mode BatchNorm2d fused with the residual add and ReLU that follow it in the ResNet encoders
// (timm `resnet18/34/50`, built at src/networks/depth.py:95-98 and src/networks/pose.py:39-41: conv -> BN -> ReLU and
// conv -> BN -> (+identity) -> ReLU).  The producer side of the loss path: after the loss kernels, BatchNorm and the
// element-wise ops around it were the largest non-convolution item of the step (MIOpen's spatial BN kernels launch one
// workgroup per channel: 64 workgroups on a 256-CU device for the stem).
//
//   forward : stats sweep (shifted sums, many blocks per channel) -> apply sweep (each block combines the channel's partials in fp64)
//   backward: reduce sweep (sum dz, sum dz*(x-mean))               -> apply sweep (same; block 0 of a channel writes g_gamma, g_beta)
//   layers with N*HW <= 16384 (layer3/4 of the encoders): ONE launch per direction, one block per channel
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

// fp64 combine of a channel's partial pairs by the whole block (every block of the apply sweeps does this for itself:
// <= 128 pairs, so the separate one-thread-per-channel finalize launch is gone and nothing needs an atomic).
__device__ __forceinline__ void combine_partials(const float* __restrict__ partial, int c, int chunks, double* red, double& s1, double& s2) {
  double a = 0.0, b = 0.0;
  for (int k = threadIdx.x; k < chunks; k += kBnBlock) { a += (double)partial[((size_t)c*chunks + k)*2]; b += (double)partial[((size_t)c*chunks + k)*2 + 1]; }
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
  const int wv = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[wv*2] = a; red[wv*2 + 1] = b; }
  __syncthreads();
  s1 = 0.0; s2 = 0.0;
#pragma unroll
  for (int k = 0; k < kBnBlock/64; ++k) { s1 += red[k*2]; s2 += red[k*2 + 1]; }
}

struct BnStat { float mean, invstd; };
__device__ __forceinline__ BnStat finish_stats(double s1, double s2, double shift, int N, int HW, float momentum, float eps, int c, bool writer,
                                               float* running_mean, float* running_var, float* save_mean, float* save_invstd) {
  const double M = (double)N*HW, m1 = s1/M, mean = shift + m1;
  double var = s2/M - m1*m1;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0/sqrt(var + (double)eps);
  if (writer) {
    save_mean[c] = (float)mean; save_invstd[c] = (float)invstd;
    if (running_mean) running_mean[c] = (float)((1.0 - momentum)*(double)running_mean[c] + momentum*mean);
    if (running_var) running_var[c] = (float)((1.0 - momentum)*(double)running_var[c] + momentum*var*(M > 1.0 ? M/(M - 1.0) : 1.0));
  }
  return {(float)mean, (float)invstd};
}

template <bool VEC>
__device__ __forceinline__ void apply_range(const float* __restrict__ x, const float* __restrict__ residual, float* __restrict__ y, int C, int HW, int c,
                                            long long lo, long long hi, float sc, float sh, int relu) {
  constexpr int W = VEC ? 4 : 1;
  for (long long i = lo + (long long)threadIdx.x*W; i < hi; i += kBnBlock*W) {
    const int n = (int)(i/HW), p = (int)(i - (long long)n*HW);
    const size_t off = ((size_t)n*C + c)*HW + p;
    if (VEC) {
      const f4 v = *(const f4*)(x + off);
      f4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = fmaf(v[q], sc, sh);
      if (residual) { const f4 r = *(const f4*)(residual + off); o += r; }
      if (relu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = fmaxf(o[q], 0.f);
      }
      *(f4*)(y + off) = o;
    } else {
      float o = fmaf(x[off], sc, sh);
      if (residual) o += residual[off];
      y[off] = relu ? fmaxf(o, 0.f) : o;
    }
  }
}

// Apply sweep of the chunked path: grid (chunks, C), the same ranges as the stats sweep.
template <bool VEC>
__global__ __launch_bounds__(kBnBlock) void k_bn_apply(const float* __restrict__ x, const float* __restrict__ residual, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ partial, int N, int C, int HW, int chunks,
                                                       float momentum, float eps, int relu, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                       float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ y) {
  __shared__ double red[2*kBnBlock/64];
  const int c = blockIdx.y, k = blockIdx.x;
  double s1, s2;
  combine_partials(partial, c, chunks, red, s1, s2);
  const BnStat st = finish_stats(s1, s2, (double)x[(size_t)c*HW], N, HW, momentum, eps, c, k == 0 && threadIdx.x == 0, running_mean, running_var,
                                 save_mean, save_invstd);
  const float sc = gamma[c]*st.invstd, sh = fmaf(-st.mean, sc, beta[c]);
  long long lo, hi;
  chunk_range((long long)N*HW, chunks, k, lo, hi);
  apply_range<VEC>(x, residual, y, C, HW, c, lo, hi, sc, sh, relu);
}

// Single-launch path for small channels (N*HW <= kBnSmall): one block per channel does stats, finalize and apply (second read from L2).
template <bool VEC>
__global__ __launch_bounds__(kBnBlock) void k_bn_fwd_small(const float* __restrict__ x, const float* __restrict__ residual, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int N, int C, int HW, float momentum, float eps, int relu,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ y) {
  __shared__ float red[2*kBnBlock/64];
  const int c = blockIdx.x;
  const long long total = (long long)N*HW;
  const float shift = x[(size_t)c*HW];
  constexpr int W = VEC ? 4 : 1;
  float s1 = 0.f, s2 = 0.f;
  for (long long i = (long long)threadIdx.x*W; i < total; i += kBnBlock*W) {
    const int n = (int)(i/HW), p = (int)(i - (long long)n*HW);
    const size_t off = ((size_t)n*C + c)*HW + p;
    if (VEC) {
      const f4 v = *(const f4*)(x + off);
#pragma unroll
      for (int q = 0; q < 4; ++q) { const float d = v[q] - shift; s1 += d; s2 = fmaf(d, d, s2); }
    } else { const float d = x[off] - shift; s1 += d; s2 = fmaf(d, d, s2); }
  }
  float r1, r2;
  block_sum2(s1, s2, red, r1, r2);
  const BnStat st = finish_stats((double)r1, (double)r2, (double)shift, N, HW, momentum, eps, c, threadIdx.x == 0, running_mean, running_var,
                                 save_mean, save_invstd);
  const float sc = gamma[c]*st.invstd, sh = fmaf(-st.mean, sc, beta[c]);
  apply_range<VEC>(x, residual, y, C, HW, c, 0, total, sc, sh, relu);
}

template <bool VEC>
__device__ __forceinline__ void bwd_reduce_range(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g_y, int C, int HW, int c,
                                                 long long lo, long long hi, float mu, int relu, float& s1, float& s2) {
  constexpr int W = VEC ? 4 : 1;
  for (long long i = lo + (long long)threadIdx.x*W; i < hi; i += kBnBlock*W) {
    const int n = (int)(i/HW), p = (int)(i - (long long)n*HW);
    const size_t off = ((size_t)n*C + c)*HW + p;
    if (VEC) {
      const f4 xv = *(const f4*)(x + off), gv = *(const f4*)(g_y + off);
      f4 yv = {1.f, 1.f, 1.f, 1.f};
      if (relu) yv = *(const f4*)(y + off);
#pragma unroll
      for (int q = 0; q < 4; ++q) { const float dz = (yv[q] > 0.f) ? gv[q] : 0.f; s1 += dz; s2 = fmaf(dz, xv[q] - mu, s2); }
    } else {
      const float dz = (!relu || y[off] > 0.f) ? g_y[off] : 0.f;
      s1 += dz; s2 = fmaf(dz, x[off] - mu, s2);
    }
  }
}

template <bool VEC>
__device__ __forceinline__ void bwd_apply_range(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g_y, float* __restrict__ g_x,
                                                float* __restrict__ g_res, int C, int HW, int c, long long lo, long long hi, float mu, float a, float b,
                                                float k2, int relu) {
  constexpr int W = VEC ? 4 : 1;
  for (long long i = lo + (long long)threadIdx.x*W; i < hi; i += kBnBlock*W) {
    const int n = (int)(i/HW), p = (int)(i - (long long)n*HW);
    const size_t off = ((size_t)n*C + c)*HW + p;
    if (VEC) {
      const f4 xv = *(const f4*)(x + off), gv = *(const f4*)(g_y + off);
      f4 yv = {1.f, 1.f, 1.f, 1.f};
      if (relu) yv = *(const f4*)(y + off);
      f4 dz, dx;
#pragma unroll
      for (int q = 0; q < 4; ++q) { dz[q] = (yv[q] > 0.f) ? gv[q] : 0.f; dx[q] = a*(dz[q] - b - (xv[q] - mu)*k2); }
      *(f4*)(g_x + off) = dx;
      if (g_res) *(f4*)(g_res + off) = dz;
    } else {
      const float dz = (!relu || y[off] > 0.f) ? g_y[off] : 0.f;
      g_x[off] = a*(dz - b - (x[off] - mu)*k2);
      if (g_res) g_res[off] = dz;
    }
  }
}

template <bool VEC>
__global__ __launch_bounds__(kBnBlock) void k_bn_bwd_reduce(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g_y,
                                                            const float* __restrict__ mean, int N, int C, int HW, int chunks, int relu,
                                                            float* __restrict__ partial) {
  __shared__ float red[2*kBnBlock/64];
  const int c = blockIdx.y, k = blockIdx.x;
  long long lo, hi;
  chunk_range((long long)N*HW, chunks, k, lo, hi);
  float s1 = 0.f, s2 = 0.f;
  bwd_reduce_range<VEC>(x, y, g_y, C, HW, c, lo, hi, mean[c], relu, s1, s2);
  float r1, r2;
  block_sum2(s1, s2, red, r1, r2);
  if (threadIdx.x == 0) { partial[((size_t)c*chunks + k)*2] = r1; partial[((size_t)c*chunks + k)*2 + 1] = r2; }
}

template <bool VEC>
__global__ __launch_bounds__(kBnBlock) void k_bn_bwd_apply(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g_y,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ partial, int N, int C, int HW, int chunks, int relu,
                                                           float* __restrict__ g_x, float* __restrict__ g_res, float* __restrict__ g_gamma,
                                                           float* __restrict__ g_beta) {
  __shared__ double red[2*kBnBlock/64];
  const int c = blockIdx.y, k = blockIdx.x;
  double s1, s2;
  combine_partials(partial, c, chunks, red, s1, s2);
  const double M = (double)N*HW, is = (double)invstd[c];
  if (k == 0 && threadIdx.x == 0) { g_beta[c] = (float)s1; g_gamma[c] = (float)(s2*is); }
  long long lo, hi;
  chunk_range((long long)N*HW, chunks, k, lo, hi);
  bwd_apply_range<VEC>(x, y, g_y, g_x, g_res, C, HW, c, lo, hi, mean[c], (float)((double)gamma[c]*is), (float)(s1/M), (float)(s2*is*is/M), relu);
}

template <bool VEC>
__global__ __launch_bounds__(kBnBlock) void k_bn_bwd_small(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g_y,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           int N, int C, int HW, int relu, float* __restrict__ g_x, float* __restrict__ g_res,
                                                           float* __restrict__ g_gamma, float* __restrict__ g_beta) {
  __shared__ float red[2*kBnBlock/64];
  const int c = blockIdx.x;
  const long long total = (long long)N*HW;
  float s1 = 0.f, s2 = 0.f;
  bwd_reduce_range<VEC>(x, y, g_y, C, HW, c, 0, total, mean[c], relu, s1, s2);
  float r1, r2;
  block_sum2(s1, s2, red, r1, r2);
  const double M = (double)total, is = (double)invstd[c];
  if (threadIdx.x == 0) { g_beta[c] = r1; g_gamma[c] = (float)((double)r2*is); }
  bwd_apply_range<VEC>(x, y, g_y, g_x, g_res, C, HW, c, 0, total, mean[c], (float)((double)gamma[c]*is), (float)((double)r1/M),
                       (float)((double)r2*is*is/M), relu);
}

constexpr long long kBnSmall = 16384;   // N*HW up to which one block per channel does the whole layer in one launch

hipError_t launch_bn_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float* running_mean, float* running_var,
                         float momentum, float eps, int relu, float* y, float* save_mean, float* save_invstd, float* ws, int N, int C, int HW,
                         hipStream_t st) {
  const bool vec = (HW % 4) == 0;
  if ((long long)N*HW <= kBnSmall) {
    if (vec) hipLaunchKernelGGL(k_bn_fwd_small<true>, dim3(C), dim3(kBnBlock), 0, st, x, residual, gamma, beta, N, C, HW, momentum, eps, relu, running_mean, running_var, save_mean, save_invstd, y);
    else hipLaunchKernelGGL(k_bn_fwd_small<false>, dim3(C), dim3(kBnBlock), 0, st, x, residual, gamma, beta, N, C, HW, momentum, eps, relu, running_mean, running_var, save_mean, save_invstd, y);
    return hipGetLastError();
  }
  const int chunks = bn_chunks(N, HW);
  if (vec) {
    hipLaunchKernelGGL(k_bn_stats<true>, dim3(chunks, C), dim3(kBnBlock), 0, st, x, N, C, HW, chunks, ws);
    hipLaunchKernelGGL(k_bn_apply<true>, dim3(chunks, C), dim3(kBnBlock), 0, st, x, residual, gamma, beta, ws, N, C, HW, chunks, momentum, eps, relu,
                       running_mean, running_var, save_mean, save_invstd, y);
  } else {
    hipLaunchKernelGGL(k_bn_stats<false>, dim3(chunks, C), dim3(kBnBlock), 0, st, x, N, C, HW, chunks, ws);
    hipLaunchKernelGGL(k_bn_apply<false>, dim3(chunks, C), dim3(kBnBlock), 0, st, x, residual, gamma, beta, ws, N, C, HW, chunks, momentum, eps, relu,
                       running_mean, running_var, save_mean, save_invstd, y);
  }
  return hipGetLastError();
}

hipError_t launch_bn_bwd(const float* x, const float* y, const float* g_y, const float* gamma, const float* save_mean, const float* save_invstd,
                         int relu, float* g_x, float* g_res, float* g_gamma, float* g_beta, float* ws, int N, int C, int HW, hipStream_t st) {
  const bool vec = (HW % 4) == 0;
  if ((long long)N*HW <= kBnSmall) {
    if (vec) hipLaunchKernelGGL(k_bn_bwd_small<true>, dim3(C), dim3(kBnBlock), 0, st, x, y, g_y, gamma, save_mean, save_invstd, N, C, HW, relu, g_x, g_res, g_gamma, g_beta);
    else hipLaunchKernelGGL(k_bn_bwd_small<false>, dim3(C), dim3(kBnBlock), 0, st, x, y, g_y, gamma, save_mean, save_invstd, N, C, HW, relu, g_x, g_res, g_gamma, g_beta);
    return hipGetLastError();
  }
  const int chunks = bn_chunks(N, HW);
  if (vec) {
    hipLaunchKernelGGL(k_bn_bwd_reduce<true>, dim3(chunks, C), dim3(kBnBlock), 0, st, x, y, g_y, save_mean, N, C, HW, chunks, relu, ws);
    hipLaunchKernelGGL(k_bn_bwd_apply<true>, dim3(chunks, C), dim3(kBnBlock), 0, st, x, y, g_y, gamma, save_mean, save_invstd, ws, N, C, HW, chunks, relu,
                       g_x, g_res, g_gamma, g_beta);
  } else {
    hipLaunchKernelGGL(k_bn_bwd_reduce<false>, dim3(chunks, C), dim3(kBnBlock), 0, st, x, y, g_y, save_mean, N, C, HW, chunks, relu, ws);
    hipLaunchKernelGGL(k_bn_bwd_apply<false>, dim3(chunks, C), dim3(kBnBlock), 0, st, x, y, g_y, gamma, save_mean, save_invstd, ws, N, C, HW, chunks, relu,
                       g_x, g_res, g_gamma, g_beta);
  }
  return hipGetLastError();
}

}  // namespace smd

// ---------------------------------------------------------------------------------------------
// MaxPool2d(kernel 3, stride 2, padding 1) of the ResNet stem.  Forward keeps the winning window position (0..8, first
// maximum in row-major scan order, as ATen's max_pool2d_with_indices) in one byte instead of an int64 flat index; the
// backward is a gather over the <= 4 windows that cover an input pixel, so g_input is written exactly once and never
// zero-filled or scattered into.
// ---------------------------------------------------------------------------------------------
namespace smd {

__global__ __launch_bounds__(256) void k_maxpool_fwd(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx, int H, int W, int Ho, int Wo,
                                                     unsigned chunks) {
  const unsigned plane = blockIdx.x/chunks, chunk = blockIdx.x - plane*chunks;
  const float* xp = x + (size_t)plane*H*W;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int o = chunk*1024 + k*256 + threadIdx.x;
    if (o >= Ho*Wo) break;
    const int oh = o/Wo, ow = o - oh*Wo;
    float best = -INFINITY; int bi = 0;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int h = 2*oh - 1 + dh;
      if (h < 0 || h >= H) continue;
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int w = 2*ow - 1 + dw;
        if (w < 0 || w >= W) continue;
        const float v = xp[h*W + w];
        if (v > best || v != v) { best = v; bi = dh*3 + dw; }
      }
    }
    y[(size_t)plane*Ho*Wo + o] = best; idx[(size_t)plane*Ho*Wo + o] = (uint8_t)bi;
  }
}

// One thread per 2x2 input block (rows 2k, 2k+1; columns 2j, 2j+1): the four windows A=(k,j), B=(k,j+1), C=(k+1,j),
// D=(k+1,j+1) are the only ones that can have selected any of its pixels, each at a fixed window position:
//   (2k,  2j)   <- A@4            (2k,  2j+1) <- A@5, B@3
//   (2k+1,2j)   <- A@7, C@1       (2k+1,2j+1) <- A@8, B@6, C@2, D@0
__global__ __launch_bounds__(256) void k_maxpool_bwd(const float* __restrict__ g_y, const uint8_t* __restrict__ idx, float* __restrict__ g_x, int H, int W,
                                                     int Ho, int Wo, unsigned chunks) {
  const unsigned plane = blockIdx.x/chunks, chunk = blockIdx.x - plane*chunks;
  const float* gp = g_y + (size_t)plane*Ho*Wo; const uint8_t* ip = idx + (size_t)plane*Ho*Wo;
  float* out = g_x + (size_t)plane*H*W;
  const int Hb = (H + 1)/2, Wb = (W + 1)/2;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int q = chunk*1024 + t*256 + threadIdx.x;
    if (q >= Hb*Wb) break;
    const int k = q/Wb, j = q - k*Wb;
    const bool hasB = j + 1 < Wo, hasC = k + 1 < Ho;
    const int ia = k*Wo + j;
    const int sa = ip[ia], sb = hasB ? ip[ia + 1] : 255, sc = hasC ? ip[ia + Wo] : 255, sd = (hasB && hasC) ? ip[ia + Wo + 1] : 255;
    const float ga = gp[ia], gb = hasB ? gp[ia + 1] : 0.f, gc = hasC ? gp[ia + Wo] : 0.f, gd = (hasB && hasC) ? gp[ia + Wo + 1] : 0.f;
    const float o00 = (sa == 4) ? ga : 0.f;
    const float o01 = ((sa == 5) ? ga : 0.f) + ((sb == 3) ? gb : 0.f);
    const float o10 = ((sa == 7) ? ga : 0.f) + ((sc == 1) ? gc : 0.f);
    const float o11 = (((sa == 8) ? ga : 0.f) + ((sb == 6) ? gb : 0.f)) + (((sc == 2) ? gc : 0.f) + ((sd == 0) ? gd : 0.f));
    const int h0 = 2*k, w0 = 2*j;
    out[h0*W + w0] = o00;
    if (w0 + 1 < W) out[h0*W + w0 + 1] = o01;
    if (h0 + 1 < H) { out[(h0 + 1)*W + w0] = o10; if (w0 + 1 < W) out[(h0 + 1)*W + w0 + 1] = o11; }
  }
}

hipError_t launch_maxpool_fwd(const float* x, float* y, uint8_t* idx, size_t planes, int H, int W, hipStream_t st) {
  const int Ho = (H - 1)/2 + 1, Wo = (W - 1)/2 + 1;
  const unsigned chunks = ceil_div(Ho*Wo, 1024);
  hipLaunchKernelGGL(k_maxpool_fwd, dim3((unsigned)(planes*chunks)), dim3(256), 0, st, x, y, idx, H, W, Ho, Wo, chunks);
  return hipGetLastError();
}
hipError_t launch_maxpool_bwd(const float* g_y, const uint8_t* idx, float* g_x, size_t planes, int H, int W, hipStream_t st) {
  const int Ho = (H - 1)/2 + 1, Wo = (W - 1)/2 + 1;
  const unsigned chunks = ceil_div(((H + 1)/2)*((W + 1)/2), 1024);
  hipLaunchKernelGGL(k_maxpool_bwd, dim3((unsigned)(planes*chunks)), dim3(256), 0, st, g_y, idx, g_x, H, W, Ho, Wo, chunks);
  return hipGetLastError();
}

}  // namespace smd
