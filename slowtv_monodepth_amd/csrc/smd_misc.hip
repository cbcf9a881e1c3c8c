// smd_misc.hip — small shared kernels: the deterministic second stage of the scalar reductions, the STREAM-style
// bandwidth probe quoted by bench.py, and the cross-lane self-test.
#include "smd_common.h"
#include "smd_kernels.h"

namespace smd {

// ---------------------------------------------------------------------------------------------
// Deterministic second stage of every scalar reduction: one block, fp64 accumulation.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_sum_partials(const float* __restrict__ partial, int count, double scale, float* out) {
  __shared__ double red[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double acc = 0.0;
  int i = threadIdx.x;
  for (; i + 3*1024 < count; i += 4*1024) {      // four independent loads in flight; the order of the additions is fixed
    const float a = partial[i], b = partial[i + 1024], c = partial[i + 2048], d = partial[i + 3072];
    acc += ((double)a + (double)b) + ((double)c + (double)d);
  }
  for (; i < count; i += 1024) acc += (double)partial[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) red[wv] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += red[k];
    out[0] = (float)(tot*scale);
  }
}

hipError_t launch_sum_partials(const float* partial, int count, double scale, float* out, hipStream_t st) {
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, st, partial, count, scale, out);
  return hipGetLastError();
}

// STREAM-style device copy (4 independent 16-B loads per lane in flight, grid-stride): the measured HBM ceiling quoted
// beside the datasheet peak.  `mode` 0 = copy (read + write), 1 = read-only sum (writes one value per block).
__global__ __launch_bounds__(256) void k_stream_copy(const f4* __restrict__ src, f4* __restrict__ dst, size_t n16, int mode) {
  const size_t stride = (size_t)gridDim.x*256;
  size_t i = (size_t)blockIdx.x*256 + threadIdx.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (; i + 3*stride < n16; i += 4*stride) {
    const f4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
    const f4 c = __builtin_nontemporal_load(src + i + 2*stride), d = __builtin_nontemporal_load(src + i + 3*stride);
    if (mode == 0) {
      __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + stride);
      __builtin_nontemporal_store(c, dst + i + 2*stride); __builtin_nontemporal_store(d, dst + i + 3*stride);
    } else acc += (a + b) + (c + d);
  }
  for (; i < n16; i += stride) { const f4 a = src[i]; if (mode == 0) dst[i] = a; else acc += a; }
  if (mode != 0 && (acc[0] + acc[1] + acc[2] + acc[3]) == 12345.678f) dst[blockIdx.x] = acc;   // keeps the loads alive; practically never taken
}
// Second variant: eight independent 16-byte loads per lane in flight, plain (cached) accesses, one block per CU-slot.
__global__ __launch_bounds__(256) void k_stream_copy8(const f4* __restrict__ src, f4* __restrict__ dst, size_t n16, int mode) {
  const size_t stride = (size_t)gridDim.x*256;
  size_t i = (size_t)blockIdx.x*256 + threadIdx.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (; i + 7*stride < n16; i += 8*stride) {
    f4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = src[i + k*stride];
    if (mode == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) dst[i + k*stride] = v[k];
    } else acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  for (; i < n16; i += stride) { const f4 a = src[i]; if (mode == 0) dst[i] = a; else acc += a; }
  if (mode != 0 && (acc[0] + acc[1] + acc[2] + acc[3]) == 12345.678f) dst[blockIdx.x] = acc;   // keeps the loads alive; practically never taken
}
// Third variant: each block owns one contiguous chunk and walks it 8 x 4 KB at a time with non-temporal accesses (every
// wave touches whole DRAM pages; measured best on MI355X: 5.9 TB/s at 1 GiB, scripts/dev/stream_probe.hip).
__global__ __launch_bounds__(256) void k_stream_chunk8(const f4* __restrict__ src, f4* __restrict__ dst, size_t n16, int mode) {
  const size_t per_block = (n16 + gridDim.x - 1)/gridDim.x;
  const size_t lo = (size_t)blockIdx.x*per_block, hi = lo + per_block < n16 ? lo + per_block : n16;
  size_t i = lo + threadIdx.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (; i + 7*256 < hi; i += 8*256) {
    f4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(src + i + k*256);
    if (mode == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(v[k], dst + i + k*256);
    } else acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  for (; i < hi; i += 256) { const f4 a = src[i]; if (mode == 0) dst[i] = a; else acc += a; }
  if (mode != 0 && (acc[0] + acc[1] + acc[2] + acc[3]) == 12345.678f) dst[blockIdx.x] = acc;
}
hipError_t launch_stream_copy(const void* src, void* dst, size_t nbytes, int mode, hipStream_t st) {
  if (mode & 4) hipLaunchKernelGGL(k_stream_chunk8, dim3(256*16), dim3(256), 0, st, (const f4*)src, (f4*)dst, nbytes/16, mode & 1);
  else if (mode & 2) hipLaunchKernelGGL(k_stream_copy8, dim3(256*16), dim3(256), 0, st, (const f4*)src, (f4*)dst, nbytes/16, mode & 1);
  else hipLaunchKernelGGL(k_stream_copy, dim3(256*8), dim3(256), 0, st, (const f4*)src, (f4*)dst, nbytes/16, mode);
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
