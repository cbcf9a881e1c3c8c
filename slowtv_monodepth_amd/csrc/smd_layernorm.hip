// smd_layernorm.hip — LayerNorm over the CHANNEL dimension of an NCHW tensor (ConvNeXt's block norm and `LayerNorm2d`).
//
// timm evaluates it as permute(NCHW -> NHWC) -> F.layer_norm -> [MLP on NHWC] -> permute back, i.e. two full-tensor layout
// copies per block and direction around ATen's layer-norm kernels (12 % + 14 % of a cfg-5 step,
// profiles/r01_bench_cfg5_before_ln_steady_state_summary.txt).  Keeping the block in NCHW (the depthwise kernel's layout;
// the MLP runs as 1x1 convolutions on the same Linear weights) leaves a norm whose reduction axis is strided by H*W:
// one lane per pixel walks the channels, so every load/store is coalesced across the wave and nothing is transposed.
// A block = 64 pixels x 4 waves that split the channels (deep stages have few pixels but many channels).
//   forward : pass 1 sum / sum of squares over C (shifted by channel 0), pass 2 normalise + affine (second read from L2)
//   backward: dx kernel (pass 1 sum_c g*gamma and sum_c g*gamma*xhat, pass 2 dx) and, separately, d gamma / d beta as a
//             BatchNorm-style per-channel reduction over the pixels (grid (chunks, C)) with an fp64 finalize.
#include "smd_common.h"
#include "smd_kernels.h"

namespace smd {

// The normalised output feeds a bf16 1x1 convolution under autocast and its gradient arrives in bf16: writing / reading those
// two tensors in bf16 directly removes a full-tensor cast kernel per block and direction.  Arithmetic stays fp32.
constexpr int kLnWaves = 4;                 // waves of a block share 64 pixels and split the channels
constexpr int kLnBlock = 64*kLnWaves;
constexpr int kLnRedItems = 8192;           // pixels a block sweeps per channel in the gamma/beta reduction
constexpr int kLnMaxChunks = 64;

// Combine per-wave partial pairs of the 64 pixels of a block through LDS; every wave gets the totals of its lane's pixel.
__device__ __forceinline__ void combine_waves(float& a, float& b, float (*red)[64][2]) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  red[wv][lane][0] = a; red[wv][lane][1] = b;
  __syncthreads();
  a = 0.f; b = 0.f;
#pragma unroll
  for (int k = 0; k < kLnWaves; ++k) { a += red[k][lane][0]; b += red[k][lane][1]; }
}

template <typename TY>
__global__ __launch_bounds__(kLnBlock) void k_ln_cf_fwd(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        TY* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                        int C, int HW, size_t npix, float eps) {
  __shared__ float red[kLnWaves][64][2];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t pix = (size_t)blockIdx.x*64 + lane;   // index over (n, p)
  const bool ok = pix < npix;
  const size_t q = ok ? pix : 0, n = q/HW, p = q - n*HW;
  const float* __restrict__ xp = x + n*(size_t)C*HW + p;
  const float shift = xp[0];
  float s1 = 0.f, s2 = 0.f;
  for (int c = wv; c < C; c += kLnWaves) { const float d = xp[(size_t)c*HW] - shift; s1 += d; s2 = fmaf(d, d, s2); }
  combine_waves(s1, s2, red);
  const float m1 = s1/(float)C;
  const float var = fmaxf(s2/(float)C - m1*m1, 0.f);
  const float mu = shift + m1, rs = rsqrtf(var + eps);
  if (!ok) return;
  if (wv == 0) { mean[pix] = mu; rstd[pix] = rs; }
  TY* __restrict__ yp = y + n*(size_t)C*HW + p;
  for (int c = wv; c < C; c += kLnWaves) st_from_float<TY>(yp, (size_t)c*HW, fmaf((xp[(size_t)c*HW] - mu)*rs, gamma[c], beta[c]));
}

// dx: per pixel a = sum_c g*gamma, b = sum_c g*gamma*xhat;  dx = rstd*(g*gamma - (a + xhat*b)/C).
template <typename TG>
__global__ __launch_bounds__(kLnBlock) void k_ln_cf_bwd_dx(const float* __restrict__ x, const TG* __restrict__ g_y, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           float* __restrict__ g_x, int C, int HW, size_t npix) {
  __shared__ float red[kLnWaves][64][2];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t pix = (size_t)blockIdx.x*64 + lane;
  const bool ok = pix < npix;
  const size_t q = ok ? pix : 0, n = q/HW, p = q - n*HW;
  const float* __restrict__ xp = x + n*(size_t)C*HW + p;
  const TG* __restrict__ gp = g_y + n*(size_t)C*HW + p;
  const float mu = mean[q], rs = rstd[q];
  float a = 0.f, b = 0.f;
  for (int c = wv; c < C; c += kLnWaves) {
    const float g = ld_as_float<TG>(gp, (size_t)c*HW)*gamma[c];
    a += g; b = fmaf(g, (xp[(size_t)c*HW] - mu)*rs, b);
  }
  combine_waves(a, b, red);
  if (!ok) return;
  const float rc = 1.f/(float)C;
  float* __restrict__ dp = g_x + n*(size_t)C*HW + p;
  for (int c = wv; c < C; c += kLnWaves) {
    const float xh = (xp[(size_t)c*HW] - mu)*rs;
    dp[(size_t)c*HW] = rs*(ld_as_float<TG>(gp, (size_t)c*HW)*gamma[c] - rc*(a + xh*b));
  }
}

// d gamma / d beta: per channel sums over all pixels of g*xhat and g — the BatchNorm-style reduction (grid (chunks, C)).
template <typename TG>
__global__ __launch_bounds__(256) void k_ln_cf_bwd_wb(const float* __restrict__ x, const TG* __restrict__ g_y, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, int N, int C, int HW, int chunks, float* __restrict__ partial) {
  __shared__ float red[8];
  const int c = blockIdx.y, k = blockIdx.x;
  const long long total = (long long)N*HW;
  const long long len = (total + chunks - 1)/chunks, lo = (long long)k*len, hi = lo + len < total ? lo + len : total;
  float s1 = 0.f, s2 = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const long long n = i/HW, p = i - n*HW;
    const size_t off = ((size_t)n*C + c)*HW + p;
    const float g = ld_as_float<TG>(g_y, off);
    s1 = fmaf(g, (x[off] - mean[i])*rstd[i], s1); s2 += g;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wv*2] = s1; red[wv*2 + 1] = s2; }
  __syncthreads();
  if (threadIdx.x < 2) partial[((size_t)c*chunks + k)*2 + threadIdx.x] = (red[threadIdx.x] + red[2 + threadIdx.x]) + (red[4 + threadIdx.x] + red[6 + threadIdx.x]);
}

__global__ __launch_bounds__(64) void k_ln_cf_bwd_finalize(const float* __restrict__ partial, int chunks, int C, float* __restrict__ g_gamma, float* __restrict__ g_beta) {
  const int c = blockIdx.x*64 + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int k = 0; k < chunks; ++k) { a += (double)partial[((size_t)c*chunks + k)*2]; b += (double)partial[((size_t)c*chunks + k)*2 + 1]; }
  g_gamma[c] = (float)a; g_beta[c] = (float)b;
}

int ln_cf_chunks(size_t npix) {
  const size_t c = (npix + kLnRedItems - 1)/kLnRedItems;
  return (int)(c < 1 ? 1 : (c > (size_t)kLnMaxChunks ? kLnMaxChunks : c));
}

hipError_t launch_ln_cf_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_bf16, float* mean, float* rstd, int N, int C, int HW,
                            float eps, hipStream_t st) {
  const size_t npix = (size_t)N*HW;
  const dim3 grid((unsigned)((npix + 63)/64));
  if (y_bf16) hipLaunchKernelGGL(k_ln_cf_fwd<bf16>, grid, dim3(kLnBlock), 0, st, x, gamma, beta, (bf16*)y, mean, rstd, C, HW, npix, eps);
  else hipLaunchKernelGGL(k_ln_cf_fwd<float>, grid, dim3(kLnBlock), 0, st, x, gamma, beta, (float*)y, mean, rstd, C, HW, npix, eps);
  return hipGetLastError();
}
hipError_t launch_ln_cf_bwd(const float* x, const void* g_y, int g_bf16, const float* gamma, const float* mean, const float* rstd, float* g_x, float* g_gamma,
                            float* g_beta, float* ws, int N, int C, int HW, hipStream_t st) {
  const size_t npix = (size_t)N*HW;
  const int chunks = ln_cf_chunks(npix);
  const dim3 grid((unsigned)((npix + 63)/64));
  if (g_bf16) {
    hipLaunchKernelGGL(k_ln_cf_bwd_dx<bf16>, grid, dim3(kLnBlock), 0, st, x, (const bf16*)g_y, gamma, mean, rstd, g_x, C, HW, npix);
    hipLaunchKernelGGL(k_ln_cf_bwd_wb<bf16>, dim3(chunks, C), dim3(256), 0, st, x, (const bf16*)g_y, mean, rstd, N, C, HW, chunks, ws);
  } else {
    hipLaunchKernelGGL(k_ln_cf_bwd_dx<float>, grid, dim3(kLnBlock), 0, st, x, (const float*)g_y, gamma, mean, rstd, g_x, C, HW, npix);
    hipLaunchKernelGGL(k_ln_cf_bwd_wb<float>, dim3(chunks, C), dim3(256), 0, st, x, (const float*)g_y, mean, rstd, N, C, HW, chunks, ws);
  }
  hipLaunchKernelGGL(k_ln_cf_bwd_finalize, dim3(ceil_div(C, 64)), dim3(64), 0, st, ws, chunks, C, g_gamma, g_beta);
  return hipGetLastError();
}

}  // namespace smd
