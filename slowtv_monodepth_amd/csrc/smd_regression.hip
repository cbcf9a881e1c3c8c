// smd_regression.hip — `RegressionLoss.forward` and its adjoint (src/losses/regression.py:11-37, 69-75).
//
//   loss = sum(mask * crit(p, t)) / sum(mask),   crit in {l1, log_l1, berhu},   (p, t) optionally mapped through to_inv first.
// berHu uses the dynamic threshold delta = 0.2 * max|p - t| over the WHOLE tensor (masked-out elements included, :32-33),
// and autograd differentiates through that max: the elements attaining it receive d loss/d delta, split evenly (ATen's
// full-reduction max backward).  Both directions are therefore two sweeps with a one-block fp64 reduction in between;
// nothing is atomically accumulated, so results are run-to-run deterministic.
//
// stats (8 floats, written by forward, read by backward): [0] max diff, [1] sum(mask), [2] #elements attaining the max,
// [3] d loss/d delta scratch (backward).
#include "smd_common.h"
#include "smd_kernels.h"

namespace smd {

constexpr int kRegrBlock = 256;
constexpr int kRegrPerThread = 4;
enum { kRegrL1 = 0, kRegrLogL1 = 1, kRegrBerhu = 2 };

__device__ __forceinline__ float inv_map(float x, bool invert) { return invert ? ((x > 0.f) ? 1.f/fmaxf(x, kEps32) : 0.f) : x; }
// d to_inv(x)/dx: -1/x^2 on the pass-through branch, 0 where x <= 0 or the clamp is active
__device__ __forceinline__ float inv_map_grad(float x, bool invert) { return invert ? ((x >= kEps32) ? -1.f/(x*x) : 0.f) : 1.f; }

template <typename T> __device__ __forceinline__ void block_reduce(T val, T* red, bool is_max, T& out) {
  red[threadIdx.x] = val;
  __syncthreads();
  for (int sft = kRegrBlock/2; sft > 0; sft >>= 1) {
    if ((int)threadIdx.x < sft) red[threadIdx.x] = is_max ? (red[threadIdx.x] > red[threadIdx.x + sft] ? red[threadIdx.x] : red[threadIdx.x + sft])
                                                            : red[threadIdx.x] + red[threadIdx.x + sft];
    __syncthreads();
  }
  out = red[0];
  __syncthreads();
}

// Sweep 1 (all modes): per-block max of diff and sum of mask; for l1 / log_l1 also the masked error sum (and err itself).
__global__ __launch_bounds__(kRegrBlock) void k_regr_sweep1(const float* __restrict__ pred, const float* __restrict__ target,
                                                            const uint8_t* __restrict__ mask, size_t N, int mode, int invert,
                                                            float* __restrict__ err, float* __restrict__ partial) {
  __shared__ float red[kRegrBlock];
  float mx = 0.f, ms = 0.f, es = 0.f;
  const size_t base = ((size_t)blockIdx.x*kRegrBlock + threadIdx.x)*kRegrPerThread;
#pragma unroll
  for (int k = 0; k < kRegrPerThread; ++k) {
    const size_t i = base + k;
    if (i < N) {
      const float diff = fabsf(inv_map(pred[i], invert) - inv_map(target[i], invert));
      const float m = mask ? (mask[i] ? 1.f : 0.f) : 1.f;
      mx = fmaxf(mx, diff); ms += m;
      if (mode != kRegrBerhu) {
        const float e = m*(mode == kRegrL1 ? diff : logf(1.f + diff));
        es += e;
        if (err) err[i] = e;
      }
    }
  }
  float o;
  block_reduce(mx, red, true, o);  if (threadIdx.x == 0) partial[blockIdx.x*3] = o;
  block_reduce(ms, red, false, o); if (threadIdx.x == 0) partial[blockIdx.x*3 + 1] = o;
  block_reduce(es, red, false, o); if (threadIdx.x == 0) partial[blockIdx.x*3 + 2] = o;
}

// One block: max / sums of the sweep-1 partials -> stats[0..1], and for l1 / log_l1 the loss.
__global__ __launch_bounds__(kRegrBlock) void k_regr_finalize1(const float* __restrict__ partial, int nblk, int mode, float* __restrict__ stats,
                                                               float* __restrict__ loss) {
  __shared__ double red[kRegrBlock];
  double mx = 0.0, ms = 0.0, es = 0.0;
  for (int i = threadIdx.x; i < nblk; i += kRegrBlock) { mx = fmax(mx, (double)partial[i*3]); ms += (double)partial[i*3 + 1]; es += (double)partial[i*3 + 2]; }
  double o;
  block_reduce(mx, red, true, o);  const double tmx = o;
  block_reduce(ms, red, false, o); const double tms = o;
  block_reduce(es, red, false, o);
  if (threadIdx.x == 0) {
    stats[0] = (float)tmx; stats[1] = (float)tms;
    if (mode != kRegrBerhu) loss[0] = (float)(o/tms);
  }
}

__device__ __forceinline__ float berhu_value(float diff, float delta) {
  return (diff <= delta) ? diff : (diff*diff + delta*delta)/(2.f*delta + kEps32);
}

// Sweep 2 (berHu): errors with the now-known delta; per-block masked error sum and count of elements attaining the max.
__global__ __launch_bounds__(kRegrBlock) void k_regr_sweep2(const float* __restrict__ pred, const float* __restrict__ target,
                                                            const uint8_t* __restrict__ mask, size_t N, int invert, const float* __restrict__ stats,
                                                            float* __restrict__ err, float* __restrict__ partial) {
  __shared__ float red[kRegrBlock];
  const float mxd = stats[0], delta = 0.2f*mxd;
  float es = 0.f, ties = 0.f;
  const size_t base = ((size_t)blockIdx.x*kRegrBlock + threadIdx.x)*kRegrPerThread;
#pragma unroll
  for (int k = 0; k < kRegrPerThread; ++k) {
    const size_t i = base + k;
    if (i < N) {
      const float diff = fabsf(inv_map(pred[i], invert) - inv_map(target[i], invert));
      const float m = mask ? (mask[i] ? 1.f : 0.f) : 1.f;
      const float e = m*berhu_value(diff, delta);
      es += e; ties += (diff == mxd) ? 1.f : 0.f;
      if (err) err[i] = e;
    }
  }
  float o;
  block_reduce(es, red, false, o);   if (threadIdx.x == 0) partial[blockIdx.x*2] = o;
  block_reduce(ties, red, false, o); if (threadIdx.x == 0) partial[blockIdx.x*2 + 1] = o;
}

__global__ __launch_bounds__(kRegrBlock) void k_regr_finalize2(const float* __restrict__ partial, int nblk, float* __restrict__ stats, float* __restrict__ loss) {
  __shared__ double red[kRegrBlock];
  double es = 0.0, ties = 0.0;
  for (int i = threadIdx.x; i < nblk; i += kRegrBlock) { es += (double)partial[i*2]; ties += (double)partial[i*2 + 1]; }
  double o;
  block_reduce(es, red, false, o);   const double tes = o;
  block_reduce(ties, red, false, o);
  if (threadIdx.x == 0) { stats[2] = (float)o; loss[0] = (float)(tes/(double)stats[1]); }
}

// Backward sweep A (berHu): per-block sum of g_e * d berhu/d delta over the quadratic branch.
__global__ __launch_bounds__(kRegrBlock) void k_regr_bwd_delta(const float* __restrict__ pred, const float* __restrict__ target,
                                                               const uint8_t* __restrict__ mask, size_t N, int invert, const float* __restrict__ stats,
                                                               float* __restrict__ partial) {
  __shared__ float red[kRegrBlock];
  const float delta = 0.2f*stats[0], q = 2.f*delta + kEps32;
  float acc = 0.f;
  const size_t base = ((size_t)blockIdx.x*kRegrBlock + threadIdx.x)*kRegrPerThread;
#pragma unroll
  for (int k = 0; k < kRegrPerThread; ++k) {
    const size_t i = base + k;
    if (i < N) {
      const float diff = fabsf(inv_map(pred[i], invert) - inv_map(target[i], invert));
      const float m = mask ? (mask[i] ? 1.f : 0.f) : 1.f;
      if (diff > delta) acc += m*(2.f*delta*q - 2.f*(diff*diff + delta*delta))/(q*q);
    }
  }
  float o;
  block_reduce(acc, red, false, o);
  if (threadIdx.x == 0) partial[blockIdx.x] = o;
}

// Backward sweep B (all modes): g_pred / g_target.
__global__ __launch_bounds__(kRegrBlock) void k_regr_bwd(const float* __restrict__ pred, const float* __restrict__ target,
                                                         const uint8_t* __restrict__ mask, size_t N, int mode, int invert,
                                                         const float* __restrict__ stats, const float* __restrict__ g_loss,
                                                         float* __restrict__ g_pred, float* __restrict__ g_target) {
  const float gs = g_loss[0]/stats[1];
  const float mxd = stats[0], delta = 0.2f*mxd, q = 2.f*delta + kEps32;
  const float g_tie = (mode == kRegrBerhu) ? 0.2f*gs*stats[3]/stats[2] : 0.f;   // d loss/d delta through the max, split over ties
  const size_t base = ((size_t)blockIdx.x*kRegrBlock + threadIdx.x)*kRegrPerThread;
#pragma unroll
  for (int k = 0; k < kRegrPerThread; ++k) {
    const size_t i = base + k;
    if (i < N) {
      const float p = pred[i], t = target[i];
      const float d = inv_map(p, invert) - inv_map(t, invert), diff = fabsf(d);
      const float ge = gs*(mask ? (mask[i] ? 1.f : 0.f) : 1.f);
      float gd;
      if (mode == kRegrL1) gd = ge;
      else if (mode == kRegrLogL1) gd = ge/(1.f + diff);
      else gd = ((diff <= delta) ? ge : ge*2.f*diff/q) + ((diff == mxd) ? g_tie : 0.f);
      const float gv = gd*((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f));
      if (g_pred) g_pred[i] = gv*inv_map_grad(p, invert);
      if (g_target) g_target[i] = -gv*inv_map_grad(t, invert);
    }
  }
}

static inline int regr_mode(int flags) { return (flags & SMD_REGR_BERHU) ? kRegrBerhu : ((flags & SMD_REGR_LOG_L1) ? kRegrLogL1 : kRegrL1); }
int regr_blocks(size_t N) { return (int)((N + (size_t)kRegrBlock*kRegrPerThread - 1)/((size_t)kRegrBlock*kRegrPerThread)); }

hipError_t launch_regression_fwd(const float* pred, const float* target, const uint8_t* mask, size_t N, int flags, float* loss, float* err,
                                 float* stats, float* ws, hipStream_t st) {
  const int mode = regr_mode(flags), inv = (flags & SMD_REGR_INVERT) ? 1 : 0, nblk = regr_blocks(N);
  hipLaunchKernelGGL(k_regr_sweep1, dim3(nblk), dim3(kRegrBlock), 0, st, pred, target, mask, N, mode, inv, err, ws);
  hipLaunchKernelGGL(k_regr_finalize1, dim3(1), dim3(kRegrBlock), 0, st, ws, nblk, mode, stats, loss);
  if (mode == kRegrBerhu) {
    hipLaunchKernelGGL(k_regr_sweep2, dim3(nblk), dim3(kRegrBlock), 0, st, pred, target, mask, N, inv, stats, err, ws);
    hipLaunchKernelGGL(k_regr_finalize2, dim3(1), dim3(kRegrBlock), 0, st, ws, nblk, stats, loss);
  }
  return hipGetLastError();
}

hipError_t launch_regression_bwd(const float* pred, const float* target, const uint8_t* mask, size_t N, int flags, float* stats,
                                 const float* g_loss, float* g_pred, float* g_target, float* ws, hipStream_t st) {
  const int mode = regr_mode(flags), inv = (flags & SMD_REGR_INVERT) ? 1 : 0, nblk = regr_blocks(N);
  if (mode == kRegrBerhu) {
    hipLaunchKernelGGL(k_regr_bwd_delta, dim3(nblk), dim3(kRegrBlock), 0, st, pred, target, mask, N, inv, stats, ws);
    hipError_t e = launch_sum_partials(ws, nblk, 1.0, stats + 3, st);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_regr_bwd, dim3(nblk), dim3(kRegrBlock), 0, st, pred, target, mask, N, mode, inv, stats, g_loss, g_pred, g_target);
  return hipGetLastError();
}

}  // namespace smd
