// smd_decoder.hip — glue between the 3x3 convolutions of the Monodepth decoder.
//
// The reference decoder (src/networks/decoders/monodepth.py:71-89, decoders/utils.py:44-54) runs, per stage,
//   reflect-pad -> conv3x3 -> ELU -> nearest x2 -> cat(skip) -> reflect-pad -> conv3x3 -> ELU -> [reflect-pad -> conv3x3 -> sigmoid]
// as separate ATen kernels, each a full read + write of the activation.  The convolutions stay with MIOpen; everything
// between them collapses into two gather kernels that write the NEXT convolution's already-padded input:
//   k_elu_pad         out = reflect_pad1(elu(x + bias))                              (B,C,h,w)           -> (B,C,h+2,w+2)
//   k_elu_up_cat_pad  out = reflect_pad1(cat(nearest_x2(elu(a + bias)), skip))       (B,Ca,h,w),(B,Cs,2h,2w) -> (B,Ca+Cs,2h+2,2w+2)
// and two adjoint gathers (deterministic, no atomics).  The convolution's bias is added here (the convolution itself runs
// bias-free), so MIOpen's separate bias pass and ATen's bias-gradient reduction over the full tensor both disappear: the
// adjoint gathers already hold the pre-activation gradient and emit its per-block channel sums.  ELU is recomputed from the saved pre-activation in the backward
// (elu'(x) = x > 0 ? 1 : exp(x)), so no activated tensor is kept.
#include "smd_common.h"
#include "smd_kernels.h"

namespace smd {

constexpr int kDecBlock = 256;
constexpr int kDecPerThread = 4;   // block-strided elements per thread: 4 independent load->store chains, 4x fewer blocks
constexpr int kDecChunk = kDecBlock*kDecPerThread;

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : __expf(x) - 1.f; }
__device__ __forceinline__ float elu1_grad(float x) { return x > 0.f ? 1.f : __expf(x); }
__device__ __forceinline__ int unpad_reflect(int p, int n) { const int r = p - 1; return r < 0 ? -r : (r >= n ? 2*(n - 1) - r : r); }

__device__ __forceinline__ void block_store_sum(float v, float* red, float* dst) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) { float t = 0.f; for (int k = 0; k < kDecBlock/64; ++k) t += red[k]; *dst = t; }
}

// g_bias[c] = sum over samples and chunks of the per-block partial sums of the pre-activation gradient (fp64, fixed order).
__global__ __launch_bounds__(64) void k_bias_finalize(const float* __restrict__ partial, int B, int C, unsigned chunks, float* __restrict__ g_bias) {
  const int c = blockIdx.x;
  double acc = 0.0;
  const unsigned per = chunks, n = (unsigned)B*per;
  for (unsigned i = threadIdx.x; i < n; i += 64) { const unsigned b = i/per, k = i - b*per; acc += (double)partial[((size_t)b*C + c)*per + k]; }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (threadIdx.x == 0) g_bias[c] = (float)acc;
}

// Sum of g over the padded positions that read un-padded index r along one axis of length n: p = r+1, plus the mirrored
// border cell when r is the second / second-to-last element.
#define SMD_PAD_ADJ_POS(r, n, p0, p1, p2) const int p0 = (r) + 1, p1 = ((r) == 1) ? 0 : -1, p2 = ((r) == (n) - 2) ? (n) + 1 : -1

template <typename TA, typename TO>
__global__ __launch_bounds__(kDecBlock) void k_elu_pad_fwd(const TA* __restrict__ x, const float* __restrict__ bias, TO* __restrict__ out, int C, int h, int w, int apply_elu,
                                                           unsigned chunks) {
  const unsigned plane = blockIdx.x/chunks, chunk = blockIdx.x - plane*chunks;
  const int H = h + 2, W = w + 2;
  const float bc = bias ? bias[plane % C] : 0.f;
#pragma unroll
  for (int k = 0; k < kDecPerThread; ++k) {
    const int idx = chunk*kDecChunk + k*kDecBlock + threadIdx.x;
    if (idx >= H*W) break;
    const int py = idx/W, px = idx - py*W;
    const float v = ld_as_float<TA>(x, (size_t)plane*h*w + unpad_reflect(py, h)*w + unpad_reflect(px, w)) + bc;
    st_from_float<TO>(out, (size_t)plane*H*W + idx, apply_elu ? elu1(v) : v);
  }
}

template <typename TA, typename TO>
__global__ __launch_bounds__(kDecBlock) void k_elu_pad_bwd(const TA* __restrict__ x, const float* __restrict__ bias, const TO* __restrict__ g_out,
                                                           TA* __restrict__ g_x, float* __restrict__ bias_partial, int C, int h, int w, int apply_elu,
                                                           unsigned chunks) {
  __shared__ float red[kDecBlock/64];
  const unsigned plane = blockIdx.x/chunks, chunk = blockIdx.x - plane*chunks;
  const int W = w + 2;
  const float bc = bias ? bias[plane % C] : 0.f;
  float bsum = 0.f;
  const TO* g = g_out + (size_t)plane*(h + 2)*W;
#pragma unroll
  for (int k = 0; k < kDecPerThread; ++k) {
    const int idx = chunk*kDecChunk + k*kDecBlock + threadIdx.x;
    if (idx >= h*w) break;
    const int i = idx/w, j = idx - i*w;
    float acc = ld_as_float<TO>(g, (i + 1)*W + j + 1);
    if (i == 1 || i == h - 2 || j == 1 || j == w - 2) {   // rare: mirrored border cells
      SMD_PAD_ADJ_POS(i, h, y0, y1, y2); SMD_PAD_ADJ_POS(j, w, x0, x1, x2);
      const int ys[3] = {y0, y1, y2}, xs[3] = {x0, x1, x2};
      acc = 0.f;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (ys[a] < 0) continue;
#pragma unroll
        for (int b = 0; b < 3; ++b) if (xs[b] >= 0) acc += ld_as_float<TO>(g, ys[a]*W + xs[b]);
      }
    }
    const float gv = apply_elu ? acc*elu1_grad(ld_as_float<TA>(x, (size_t)plane*h*w + idx) + bc) : acc;
    st_from_float<TA>(g_x, (size_t)plane*h*w + idx, gv); bsum += gv;
  }
  if (bias_partial) block_store_sum(bsum, red, bias_partial + blockIdx.x);
}

template <typename TA, typename TS, typename TO>
__global__ __launch_bounds__(kDecBlock) void k_elu_up_cat_pad_fwd(const TA* __restrict__ a, const float* __restrict__ bias, const TS* __restrict__ skip,
                                                                  TO* __restrict__ out, int Ca, int Cs, int h, int w, unsigned chunks) {
  const unsigned plane = blockIdx.x/chunks, chunk = blockIdx.x - plane*chunks;   // plane = b*(Ca+Cs) + c
  const int C = Ca + Cs, H2 = 2*h, W2 = 2*w, H = H2 + 2, W = W2 + 2;
  const unsigned b = plane/C, c = plane - b*C;
  const bool from_a = (int)c < Ca;
  const float bc = (from_a && bias) ? bias[c] : 0.f;
  const TA* src_a = a + ((size_t)b*Ca + (from_a ? c : 0))*h*w;
  const TS* src_s = from_a ? nullptr : skip + ((size_t)b*Cs + (c - Ca))*H2*W2;
#pragma unroll
  for (int k = 0; k < kDecPerThread; ++k) {
    const int idx = chunk*kDecChunk + k*kDecBlock + threadIdx.x;
    if (idx >= H*W) break;
    const int py = idx/W, px = idx - py*W;
    const int r = unpad_reflect(py, H2), q = unpad_reflect(px, W2);
    st_from_float<TO>(out, (size_t)plane*H*W + idx, from_a ? elu1(ld_as_float<TA>(src_a, (r >> 1)*w + (q >> 1)) + bc) : ld_as_float<TS>(src_s, r*W2 + q));
  }
}

// Adjoint w.r.t. `a` (low resolution): each source pixel feeds a 2x2 block of the up-sampled map, each cell of which
// feeds its padded position plus (on the second / second-to-last row or column) the mirrored border cell.
template <typename TA, typename TO>
__global__ __launch_bounds__(kDecBlock) void k_elu_up_cat_pad_bwd_a(const TA* __restrict__ a, const float* __restrict__ bias, const TO* __restrict__ g_out,
                                                                    TA* __restrict__ g_a, float* __restrict__ bias_partial, int Ca, int Cs, int h, int w,
                                                                    unsigned chunks) {
  __shared__ float red[kDecBlock/64];
  float bsum = 0.f;
  const unsigned plane = blockIdx.x/chunks, chunk = blockIdx.x - plane*chunks;   // plane = b*Ca + c
  const int C = Ca + Cs, H2 = 2*h, W2 = 2*w, W = W2 + 2;
  const unsigned b = plane/Ca, c = plane - b*Ca;
  const TO* g = g_out + ((size_t)b*C + c)*(H2 + 2)*W;
  const float bc = bias ? bias[c] : 0.f;
  for (int k = 0; k < kDecPerThread; ++k) {
  const int idx = chunk*kDecChunk + k*kDecBlock + threadIdx.x;
  if (idx >= h*w) break;
  const int i = idx/w, j = idx - i*w;
  float acc = 0.f;
  if (i > 0 && i < h - 1 && j > 0 && j < w - 1) {   // interior: the plain 2x2 block
    const TO* gp = g + (2*i + 1)*W + 2*j + 1;
    acc = (ld_as_float<TO>(gp, 0) + ld_as_float<TO>(gp, 1)) + (ld_as_float<TO>(gp, W) + ld_as_float<TO>(gp, W + 1));
  } else
#pragma unroll
  for (int dr = 0; dr < 2; ++dr) {
    const int r = 2*i + dr;
    SMD_PAD_ADJ_POS(r, H2, y0, y1, y2);
    const int ys[3] = {y0, y1, y2};
#pragma unroll
    for (int dq = 0; dq < 2; ++dq) {
      const int q = 2*j + dq;
      SMD_PAD_ADJ_POS(q, W2, x0, x1, x2);
      const int xs[3] = {x0, x1, x2};
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        if (ys[m] < 0) continue;
#pragma unroll
        for (int n = 0; n < 3; ++n) if (xs[n] >= 0) acc += ld_as_float<TO>(g, ys[m]*W + xs[n]);
      }
    }
  }
  const float gv = acc*elu1_grad(ld_as_float<TA>(a, (size_t)plane*h*w + idx) + bc);
  st_from_float<TA>(g_a, (size_t)plane*h*w + idx, gv); bsum += gv;
  }
  if (bias_partial) block_store_sum(bsum, red, bias_partial + blockIdx.x);
}

// Adjoint w.r.t. the skip tensor (full resolution): plain reflection-pad adjoint of its channel slice.
template <typename TS, typename TO>
__global__ __launch_bounds__(kDecBlock) void k_elu_up_cat_pad_bwd_skip(const TO* __restrict__ g_out, TS* __restrict__ g_skip,
                                                                       int Ca, int Cs, int h, int w, unsigned chunks) {
  const unsigned plane = blockIdx.x/chunks, chunk = blockIdx.x - plane*chunks;   // plane = b*Cs + c
  const int C = Ca + Cs, H2 = 2*h, W2 = 2*w, W = W2 + 2;
  const unsigned b = plane/Cs, c = plane - b*Cs;
  const TO* g = g_out + ((size_t)b*C + Ca + c)*(H2 + 2)*W;
#pragma unroll
  for (int k = 0; k < kDecPerThread; ++k) {
    const int idx = chunk*kDecChunk + k*kDecBlock + threadIdx.x;
    if (idx >= H2*W2) break;
    const int r = idx/W2, q = idx - r*W2;
    float acc = ld_as_float<TO>(g, (r + 1)*W + q + 1);
    if (r == 1 || r == H2 - 2 || q == 1 || q == W2 - 2) {
      SMD_PAD_ADJ_POS(r, H2, y0, y1, y2); SMD_PAD_ADJ_POS(q, W2, x0, x1, x2);
      const int ys[3] = {y0, y1, y2}, xs[3] = {x0, x1, x2};
      acc = 0.f;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        if (ys[m] < 0) continue;
#pragma unroll
        for (int n = 0; n < 3; ++n) if (xs[n] >= 0) acc += ld_as_float<TO>(g, ys[m]*W + xs[n]);
      }
    }
    st_from_float<TS>(g_skip, (size_t)plane*H2*W2 + idx, acc);
  }
}

// dtypes: bit 0 = x / a (and their gradients) are bf16, bit 1 = skip (and its gradient), bit 2 = out (and its gradient).
#define SMD_DT_A 1
#define SMD_DT_S 2
#define SMD_DT_O 4

hipError_t launch_elu_pad_fwd(const void* x, const float* bias, void* out, int B, int C, int h, int w, int apply_elu, int dt, hipStream_t st) {
  const unsigned chunks = ceil_div((h + 2)*(w + 2), kDecChunk);
  const dim3 grid((unsigned)((size_t)B*C*chunks)), blk(kDecBlock);
#define SMD_GO(TA_, TO_) hipLaunchKernelGGL((k_elu_pad_fwd<TA_, TO_>), grid, blk, 0, st, (const TA_*)x, bias, (TO_*)out, C, h, w, apply_elu, chunks)
  if (dt & SMD_DT_A) { if (dt & SMD_DT_O) SMD_GO(bf16, bf16); else SMD_GO(bf16, float); }
  else { if (dt & SMD_DT_O) SMD_GO(float, bf16); else SMD_GO(float, float); }
#undef SMD_GO
  return hipGetLastError();
}
size_t decoder_bias_partials(int B, int C, int h, int w) { return (size_t)B*C*ceil_div(h*w, kDecChunk); }
hipError_t launch_elu_pad_bwd(const void* x, const float* bias, const void* g_out, void* g_x, float* g_bias, float* ws, int B, int C, int h, int w,
                              int apply_elu, int dt, hipStream_t st) {
  const unsigned chunks = ceil_div(h*w, kDecChunk);
  const dim3 grid((unsigned)((size_t)B*C*chunks)), blk(kDecBlock);
#define SMD_GO(TA_, TO_) hipLaunchKernelGGL((k_elu_pad_bwd<TA_, TO_>), grid, blk, 0, st, (const TA_*)x, bias, (const TO_*)g_out, (TA_*)g_x, g_bias ? ws : nullptr, C, h, w, apply_elu, chunks)
  if (dt & SMD_DT_A) { if (dt & SMD_DT_O) SMD_GO(bf16, bf16); else SMD_GO(bf16, float); }
  else { if (dt & SMD_DT_O) SMD_GO(float, bf16); else SMD_GO(float, float); }
#undef SMD_GO
  if (g_bias) hipLaunchKernelGGL(k_bias_finalize, dim3(C), dim3(64), 0, st, ws, B, C, chunks, g_bias);
  return hipGetLastError();
}
hipError_t launch_elu_up_cat_pad_fwd(const void* a, const float* bias, const void* skip, void* out, int B, int Ca, int Cs, int h, int w, int dt,
                                     hipStream_t st) {
  const unsigned chunks = ceil_div((2*h + 2)*(2*w + 2), kDecChunk);
  const dim3 grid((unsigned)((size_t)B*(Ca + Cs)*chunks)), blk(kDecBlock);
#define SMD_GO(TA_, TS_, TO_) hipLaunchKernelGGL((k_elu_up_cat_pad_fwd<TA_, TS_, TO_>), grid, blk, 0, st, (const TA_*)a, bias, (const TS_*)skip, (TO_*)out, Ca, Cs, h, w, chunks)
  switch (dt & 7) {
    case 0: SMD_GO(float, float, float); break;  case 1: SMD_GO(bf16, float, float); break;
    case 2: SMD_GO(float, bf16, float); break;   case 3: SMD_GO(bf16, bf16, float); break;
    case 4: SMD_GO(float, float, bf16); break;   case 5: SMD_GO(bf16, float, bf16); break;
    case 6: SMD_GO(float, bf16, bf16); break;    default: SMD_GO(bf16, bf16, bf16); break;
  }
#undef SMD_GO
  return hipGetLastError();
}
hipError_t launch_elu_up_cat_pad_bwd(const void* a, const float* bias, const void* g_out, void* g_a, void* g_skip, float* g_bias, float* ws,
                                     int B, int Ca, int Cs, int h, int w, int dt, hipStream_t st) {
  if (g_a) {
    const unsigned chunks = ceil_div(h*w, kDecChunk);
    const dim3 grid((unsigned)((size_t)B*Ca*chunks)), blk(kDecBlock);
#define SMD_GO(TA_, TO_) hipLaunchKernelGGL((k_elu_up_cat_pad_bwd_a<TA_, TO_>), grid, blk, 0, st, (const TA_*)a, bias, (const TO_*)g_out, (TA_*)g_a, g_bias ? ws : nullptr, Ca, Cs, h, w, chunks)
    if (dt & SMD_DT_A) { if (dt & SMD_DT_O) SMD_GO(bf16, bf16); else SMD_GO(bf16, float); }
    else { if (dt & SMD_DT_O) SMD_GO(float, bf16); else SMD_GO(float, float); }
#undef SMD_GO
    if (g_bias) hipLaunchKernelGGL(k_bias_finalize, dim3(Ca), dim3(64), 0, st, ws, B, Ca, chunks, g_bias);
  }
  if (g_skip && Cs > 0) {
    const unsigned chunks = ceil_div(4*h*w, kDecChunk);
    const dim3 grid((unsigned)((size_t)B*Cs*chunks)), blk(kDecBlock);
#define SMD_GO(TS_, TO_) hipLaunchKernelGGL((k_elu_up_cat_pad_bwd_skip<TS_, TO_>), grid, blk, 0, st, (const TO_*)g_out, (TS_*)g_skip, Ca, Cs, h, w, chunks)
    if (dt & SMD_DT_S) { if (dt & SMD_DT_O) SMD_GO(bf16, bf16); else SMD_GO(bf16, float); }
    else { if (dt & SMD_DT_O) SMD_GO(float, bf16); else SMD_GO(float, float); }
#undef SMD_GO
  }
  return hipGetLastError();
}

}  // namespace smd
