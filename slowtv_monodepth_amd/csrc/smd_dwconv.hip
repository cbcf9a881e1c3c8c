// smd_dwconv.hip — depthwise 7x7 convolution (stride 1, padding 3, groups = C) of the ConvNeXt blocks
// (timm `convnext_*` encoders built at src/networks/depth.py:95-98; BASELINE configs 2-4), forward, data gradient and
// weight gradient.  MIOpen routes this layer to a batched-GEMM weight-gradient kernel that takes 48 % of a cfg-3 training
// step (52 ms; profiles/r01_bench_cfg3_steady_state_summary.txt); it is a 49-tap stencil, i.e. LDS-tile + register work.
//
// Tile: 32 x 64 outputs per 256-thread block, every thread a 1 x 8 strip.  The (32+6) x (64+6) input window is staged in
// LDS once (zero-filled outside the image); a thread reads, per kernel row, the 14 inputs its strip needs as four 16-byte
// LDS loads and does 7 x 8 FMAs on them: 28 LDS loads per 392 FMAs, so the VALU, not the LDS pipe, is the limit.  The 49
// weights of the block's channel are wave-uniform (scalar registers).
//   forward / data gradient : same kernel, the data gradient reads the weights mirrored (w[48 - k]).
//   weight gradient         : a block owns (channel, tile) and loops over the N samples, keeping 49 + 1 accumulators in
//                             registers; one block-wide reduction at the end; per-tile partials summed by a finalize pass.
#include "smd_common.h"
#include "smd_kernels.h"

namespace smd {

constexpr int kDwTH = 32, kDwTW = 64, kDwK = 7, kDwR = 3;
constexpr int kDwLH = kDwTH + 2*kDwR;          // 38 staged rows
constexpr int kDwLW = 72;                      // 70 staged columns, padded to a multiple of 4 floats

__device__ __forceinline__ void stage_tile(const float* __restrict__ plane, int H, int W, int y0, int x0, float* __restrict__ lds) {
  for (int i = threadIdx.x; i < kDwLH*kDwLW; i += 256) {
    const int r = i/kDwLW, c = i - r*kDwLW;
    const int y = y0 + r - kDwR, x = x0 + c - kDwR;
    lds[i] = (c < kDwTW + 2*kDwR && y >= 0 && y < H && x >= 0 && x < W) ? plane[(size_t)y*W + x] : 0.f;
  }
}

template <bool FLIP>
__global__ __launch_bounds__(256) void k_dwconv7_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                     float* __restrict__ y, int C, int H, int W, int tiles_x, int tiles) {
  __shared__ __attribute__((aligned(16))) float lds[kDwLH*kDwLW];
  const int plane = blockIdx.x/tiles, tile = blockIdx.x - plane*tiles;
  const int ty = tile/tiles_x, tx = tile - ty*tiles_x;
  const int c = plane % C;
  const int y0 = ty*kDwTH, x0 = tx*kDwTW;
  stage_tile(x + (size_t)plane*H*W, H, W, y0, x0, lds);
  __syncthreads();
  const int row = threadIdx.x >> 3, cs = (threadIdx.x & 7)*8;
  const float* __restrict__ wc = w + (size_t)c*49;
  float acc[8];
  const float b0 = bias ? bias[c] : 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = b0;
#pragma unroll
  for (int di = 0; di < kDwK; ++di) {
    float seg[16];
    const f4* lp = (const f4*)(lds + (row + di)*kDwLW + cs);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const f4 v = lp[q]; seg[q*4] = v[0]; seg[q*4 + 1] = v[1]; seg[q*4 + 2] = v[2]; seg[q*4 + 3] = v[3]; }
#pragma unroll
    for (int dj = 0; dj < kDwK; ++dj) {
      const float wv = FLIP ? wc[48 - (di*7 + dj)] : wc[di*7 + dj];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = fmaf(wv, seg[k + dj], acc[k]);
    }
  }
  const int oy = y0 + row;
  if (oy < H) {
    float* __restrict__ yp = y + (size_t)plane*H*W + (size_t)oy*W + x0 + cs;
    if (x0 + cs + 8 <= W && (W & 3) == 0) {
      *(f4*)yp = f4{acc[0], acc[1], acc[2], acc[3]}; *(f4*)(yp + 4) = f4{acc[4], acc[5], acc[6], acc[7]};
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) if (x0 + cs + k < W) yp[k] = acc[k];
    }
  }
}

// Weight gradient: gw[c][di][dj] = sum_{n,y,x} gy[n,c,y,x] * x[n,c,y+di-3,x+dj-3];  gb[c] = sum gy.
__global__ __launch_bounds__(256) void k_dwconv7_wrw(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ partial,
                                                     int N, int C, int H, int W, int tiles_x, int tiles) {
  __shared__ __attribute__((aligned(16))) float lds[kDwLH*kDwLW];
  __shared__ float red[4][52];
  const int c = blockIdx.x/tiles, tile = blockIdx.x - c*tiles;
  const int ty = tile/tiles_x, tx = tile - ty*tiles_x;
  const int y0 = ty*kDwTH, x0 = tx*kDwTW;
  const int row = threadIdx.x >> 3, cs = (threadIdx.x & 7)*8;
  const int oy = y0 + row;
  float acc[50];
#pragma unroll
  for (int k = 0; k < 50; ++k) acc[k] = 0.f;
  for (int n = 0; n < N; ++n) {
    const size_t po = ((size_t)n*C + c)*H*W;
    __syncthreads();
    stage_tile(x + po, H, W, y0, x0, lds);
    float g[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = (oy < H && x0 + cs + k < W) ? gy[po + (size_t)oy*W + x0 + cs + k] : 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[49] += g[k];
#pragma unroll
    for (int di = 0; di < kDwK; ++di) {
      float seg[16];
      const f4* lp = (const f4*)(lds + (row + di)*kDwLW + cs);
#pragma unroll
      for (int q = 0; q < 4; ++q) { const f4 v = lp[q]; seg[q*4] = v[0]; seg[q*4 + 1] = v[1]; seg[q*4 + 2] = v[2]; seg[q*4 + 3] = v[3]; }
#pragma unroll
      for (int dj = 0; dj < kDwK; ++dj) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[di*7 + dj] = fmaf(g[k], seg[k + dj], acc[di*7 + dj]);
      }
    }
  }
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 50; ++k) {
    const float t = wave_sum(acc[k]);
    if ((threadIdx.x & 63) == 0) red[wv][k] = t;
  }
  __syncthreads();
  if (threadIdx.x < 50) partial[((size_t)c*tiles + tile)*50 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(64) void k_dwconv7_wrw_finalize(const float* __restrict__ partial, int tiles, float* __restrict__ gw, float* __restrict__ gb) {
  const int c = blockIdx.x, k = threadIdx.x;
  if (k >= 50) return;
  double s = 0.0;
  for (int t = 0; t < tiles; ++t) s += (double)partial[((size_t)c*tiles + t)*50 + k];
  if (k < 49) gw[(size_t)c*49 + k] = (float)s; else if (gb) gb[c] = (float)s;
}

int dwconv_tiles(int H, int W) { return ceil_div(H, kDwTH)*ceil_div(W, kDwTW); }

hipError_t launch_dwconv7(const float* x, const float* w, const float* bias, float* y, int N, int C, int H, int W, int flip, hipStream_t st) {
  const int tiles_x = ceil_div(W, kDwTW), tiles = dwconv_tiles(H, W);
  const unsigned grid = (unsigned)((size_t)N*C*tiles);
  if (flip) hipLaunchKernelGGL(k_dwconv7_fwd<true>, dim3(grid), dim3(256), 0, st, x, w, bias, y, C, H, W, tiles_x, tiles);
  else hipLaunchKernelGGL(k_dwconv7_fwd<false>, dim3(grid), dim3(256), 0, st, x, w, bias, y, C, H, W, tiles_x, tiles);
  return hipGetLastError();
}

hipError_t launch_dwconv7_wrw(const float* x, const float* gy, float* gw, float* gb, float* ws, int N, int C, int H, int W, hipStream_t st) {
  const int tiles_x = ceil_div(W, kDwTW), tiles = dwconv_tiles(H, W);
  hipLaunchKernelGGL(k_dwconv7_wrw, dim3((unsigned)((size_t)C*tiles)), dim3(256), 0, st, x, gy, ws, N, C, H, W, tiles_x, tiles);
  hipLaunchKernelGGL(k_dwconv7_wrw_finalize, dim3(C), dim3(64), 0, st, ws, tiles, gw, gb);
  return hipGetLastError();
}

}  // namespace smd
