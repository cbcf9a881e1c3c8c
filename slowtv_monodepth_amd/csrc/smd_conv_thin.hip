// smd_conv_thin.hip — the Monodepth decoder's thin last stage on the f32 MFMA: conv3x3(32 -> 16) at half resolution and conv3x3(16 -> 16) at full resolution
// (reference: src/networks/decoders/monodepth.py:45-50, 80-84 — `self.up0['0'] = ConvELU(32, 16)`, `self.up1['0'] = ConvELU(16, 16)`;
// decoders/utils.py:44-54).
//
// With sixteen output channels a 3x3 convolution has 2304-4608 multiply-adds per pixel on 128-192 bytes: compute-bound in fp32, but far too thin for the tiles
// of MIOpen's wide-layer kernels (cfg 2, b = 12: 16 -> 16 at 192x640 196 us forward and 534 us backward, 35 / 25 TFLOP/s, where the fat layers of the same
// decoder reach 90).  Here: `v_mfma_f32_16x16x4_f32` (exact f32 products and accumulation at the vector ALU's peak rate, 157 TFLOP/s, without its issue costs),
// operands staged through LDS (the matrix core's operand layout is the transpose of the memory's).  Forward and data gradient are one kernel form; the weight
// gradient is a GEMM with K = pixels, per-block sums and a fixed-order fp64 second stage.  Round 6: smd_conv_mfma.hip serves the same layers on the bf16
// matrix cores with three-way split operands (fp32-class results at 6/16 of this instruction's time: forward 88 -> 52 us, data gradient 100 -> 63 at cfg 2);
// `functional._conv_route` keeps whichever form wins this box's A/B per operator — this file's weight gradient still does.
#include "smd_common.h"
#include "smd_kernels.h"

namespace smd {

constexpr int kThinCO = 16;                    // output channels of the layer

// ---- forward and data gradient (v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate, the vector ALU's peak rate without its issue costs) ----
// out[px][co] = sum_K A[px][K] Bw[K][co] with K = (tap, input channel): a wave owns 16 pixels of a row x TR rows (one 16 x 16 accumulator tile per row and
// 16 output channels), K runs in steps of four input channels of one tap.  Lane l = (i = l & 15, q = l >> 4) feeds A[pixel i][channel 4 cg + q] and
// Bw[4 cg + q][co = i];  D[row = 4 q + v][col = i]: lane l ends with pixels 4 q .. 4 q + 3 of output channel i.
// Everything goes through LDS, because the matrix core's operand layout is the opposite of the memory's: a lane group's 16 pixels are 64 bytes of four
// different channel planes (first version: operands straight from global memory, 120 us at cfg 2 — the loads' request count, not the MFMAs, not the
// bytes).  A block of four waves stages its 64 x TR tile of every input channel with its halo (rows of 66 floats, coalesced), the weights filed by
// (K step, lane), and at the end the output tile, so that memory sees whole rows.  Channel stride of the input tile = 16 mod 32 floats: the four channel
// groups of a ds_read land on the two halves of the banks.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CI, int NT, bool BWD>
struct ThinTile {
  static constexpr int TR = (CI*NT <= 16) ? 2 : 4;   // rows of a block's tile.  16 -> 16 at cfg 2: 8 rows (53 KB of LDS, 3 blocks per CU) 105-120 us forward, 4 rows 88-95,
                                                      // 2 rows (27 KB, 6 blocks) 93 forward and 101 instead of 112-119 for the data gradient: more blocks to overlap the phases of
  static constexpr int LW = 66;                                             // 64 columns + halo
  static constexpr int CS = (((TR + 2)*LW + 15)/32)*32 + 16;                // channel stride of the input tile (floats), = 16 mod 32, >= (TR + 2) LW
  static constexpr int OS = TR*64 + 4;                                      // channel stride of the output tile
  static constexpr int kIn = CI*CS, kOut = NT*16*OS;
  static constexpr int kTile = kIn > kOut ? kIn : kOut;
  static constexpr int kW = 9*(CI/4)*NT*64;
};

template <int CI, int NT, bool BWD>
__global__ __launch_bounds__(256) void k_thin_mfma(const float* __restrict__ in, const float* __restrict__ wgt, float* __restrict__ out,
                                                   int hi, int wi, int ho, int wo) {
  using TT = ThinTile<CI, NT, BWD>;
  constexpr int KC = CI/4, TR = TT::TR, LW = TT::LW, CS = TT::CS, OS = TT::OS, off = BWD ? 2 : 0;
  static_assert(CS >= (TR + 2)*LW && CS % 32 == 16, "channel stride of the input tile");
  __shared__ float tile[TT::kTile];
  __shared__ float wl[TT::kW];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 15, q = lane >> 4;   // (wv on the scalar unit: row addresses are wave-uniform)
  const int xb = blockIdx.x*64, y0 = blockIdx.y*TR, b = blockIdx.z;
  // weights: forward w[co][c][tap]; data gradient (in = dL/dy with CI = the layer's output channels, out = the layer's NT*16 input channels) w[c][co][8 - tap]
  // (both staging loops request a batch of loads before they file it: one load per trip would cost a block forty memory latencies in a row — measured,
  // 164 us instead of 120)
  constexpr int kW = 9*CI*NT*16, kTrips = (kW + 255)/256;
  float wv_[kTrips];
#pragma unroll
  for (int t = 0; t < kTrips; ++t) { const int e = t*256 + threadIdx.x; wv_[t] = e < kW ? wgt[e] : 0.f; }
  // input tile: rows y0 - off .. y0 - off + TR + 1, columns xb - off .. xb - off + 65 of every channel; a wave per row of 66, kBatch rows in flight
  {
    const size_t plane = (size_t)hi*wi;
    const float* src = in + (size_t)b*CI*plane;
    constexpr int kRows = CI*(TR + 2), kPerWave = kRows/4, kBatch = (kPerWave <= 24) ? kPerWave : kPerWave/2;
    static_assert(kRows % 4 == 0 && kPerWave % kBatch == 0, "rows of the input tile per wave");
    for (int r0 = 0; r0 < kPerWave; r0 += kBatch) {
      float v0[kBatch], v1[kBatch];
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        const int rr = (r0 + k)*4 + wv;
        const int c = rr/(TR + 2), r = rr - c*(TR + 2);
        const int yy = y0 + r - off;
        const int xx0 = xb + lane - off, xx1 = xx0 + 64;
        if (BWD) {
          const bool yok = yy >= 0 && yy < hi;
          v0[k] = (yok && xx0 >= 0 && xx0 < wi) ? src[(size_t)c*plane + (size_t)yy*wi + xx0] : 0.f;
          v1[k] = (lane < 2 && yok && xx1 >= 0 && xx1 < wi) ? src[(size_t)c*plane + (size_t)yy*wi + xx1] : 0.f;
        } else {                                   // (beyond the image: any valid address, those outputs are not stored)
          const float* rowp = src + (size_t)c*plane + (size_t)min(yy, hi - 1)*wi;
          v0[k] = rowp[min(xx0, wi - 1)];
          v1[k] = lane < 2 ? rowp[min(xx1, wi - 1)] : 0.f;
        }
      }
      if (r0 == 0) {                                // the weights were requested before the tile's rows: one latency for both
#pragma unroll
        for (int t = 0; t < kTrips; ++t) {
          const int e = t*256 + threadIdx.x;
          if (e < kW) {
            const int tap_m = e % 9, r1 = e/9;
            int c, co;
            if (BWD) { co = r1 % (NT*16); c = r1/(NT*16); } else { c = r1 % CI; co = r1/CI; }
            const int tap = BWD ? 8 - tap_m : tap_m;
            wl[((tap*KC + (c >> 2))*NT + (co >> 4))*64 + (c & 3)*16 + (co & 15)] = wv_[t];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        const int rr = (r0 + k)*4 + wv;
        const int c = rr/(TR + 2), r = rr - c*(TR + 2);
        tile[c*CS + r*LW + lane] = v0[k];
        if (lane < 2) tile[c*CS + r*LW + 64 + lane] = v1[k];
      }
    }
  }
  __syncthreads();
  f32x4 acc[TR][NT];
#pragma unroll
  for (int r = 0; r < TR; ++r)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[r][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* ta = tile + q*CS + wv*16 + i;
#pragma unroll
  for (int cg = 0; cg < KC; ++cg) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      float a[TR + 2];
#pragma unroll
      for (int r = 0; r < TR + 2; ++r) a[r] = ta[cg*4*CS + r*LW + kx];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        float bwv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bwv[nt] = wl[(((ky*3 + kx)*KC + cg)*NT + nt)*64 + lane];
#pragma unroll
        for (int r = 0; r < TR; ++r)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r + ky], bwv[nt], acc[r][nt], 0, 0, 0);
      }
    }
  }
  if (!BWD && (wo & 3) == 0 && xb + 64 <= wo) {     // whole tile inside an image whose rows are 16-byte multiples: a lane's four pixels leave as one 16-byte store
#pragma unroll
    for (int r = 0; r < TR; ++r) {
      if (y0 + r >= ho) break;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        *reinterpret_cast<f32x4*>(out + (((size_t)b*(NT*16) + nt*16 + i)*ho + y0 + r)*wo + xb + wv*16 + q*4) = acc[r][nt];
    }
    return;
  }
  // (the data gradient's rows of w + 2 floats are not 16-byte multiples; its pixels as four dword stores per lane: 117 us instead of 97 through LDS)
  __syncthreads();                                 // every wave is done with the input tile: it becomes the output tile
#pragma unroll
  for (int r = 0; r < TR; ++r)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int v = 0; v < 4; ++v) tile[(nt*16 + i)*OS + r*64 + wv*16 + q*4 + v] = acc[r][nt][v];
  __syncthreads();
  for (int rr = wv; rr < NT*16*TR; rr += 4) {      // a wave per output row of 64
    const int co = rr/TR, r = rr - co*TR;
    if (y0 + r < ho && xb + lane < wo) out[(((size_t)b*(NT*16) + co)*ho + y0 + r)*wo + xb + lane] = tile[co*OS + r*64 + lane];
  }
}

// ---- weight gradient: g_w[co][c][tap] = sum over samples and pixels g_y[co][px] xp[c][px + tap] = a GEMM with M = 16 output channels, N = 9 taps x CI
// input channels, K = pixels.  A K step is four consecutive pixels of a row: A[co = i][pixel q] = g_y, B[pixel q][c = i] = xp shifted by the tap, one
// 16 x 16 accumulator tile per (tap, group of 16 input channels): 9 CI / 16 tiles = 9 CI / 4 VGPRs.  A block stages a 64 x 4 tile of g_y and the matching
// 66 x 6 tile of every xp channel in LDS (channel strides = 4 mod 32 floats: the 16 channels x 4 pixels of a ds_read cover the banks twice), each wave
// takes one row of the tile (16 K steps, 16 x 9 CI / 16 MFMAs), and the block walks kThinWgtRows / 4 such tiles before its four waves' accumulators are
// added up in LDS (wave order) and leave as ONE set of 144 CI sums per block; k_thin_wgt_finalize adds the blocks' sums in fp64 in block order.
constexpr int kThinWgtRows = 24;               // image rows per block (6 tiles)
template <int CI>
__global__ __launch_bounds__(256) void k_thin_wgt_mfma(const float* __restrict__ xp, const float* __restrict__ gy, float* __restrict__ partial, int h, int w) {
  constexpr int CO = kThinCO, NG = CI/16, NTL = 9*NG, TR = 4, LW = 66;
  constexpr int GS = TR*64 + 4;                                  // channel stride of the g_y tile: 260 = 4 mod 32
  constexpr int XS = (((TR + 2)*LW + 27)/32)*32 + 4;             // channel stride of the xp tile: >= 396, = 4 mod 32
  static_assert(GS % 32 == 4 && XS % 32 == 4 && XS >= (TR + 2)*LW, "channel strides");
  constexpr int kStage = CO*GS + CI*XS, kRed = 4*NTL*4*64;
  __shared__ float lds[kStage > kRed ? kStage : kRed];
  float* const tg = lds;
  float* const tx = lds + CO*GS;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 15, q = lane >> 4;   // (wv on the scalar unit: row addresses are wave-uniform)
  const int xb = blockIdx.x*64, b = blockIdx.z, W = w + 2, H = h + 2;
  const int ylo = blockIdx.y*kThinWgtRows, yhi = min(ylo + kThinWgtRows, h);
  f32x4 acc[NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int y0 = ylo; y0 < yhi; y0 += TR) {
    {   // stage: a wave per row, every request before the first store
      constexpr int kG = CO*TR/4, kX = CI*(TR + 2)/4;
      float vg[kG], vx0[kX], vx1[kX];
#pragma unroll
      for (int k = 0; k < kG; ++k) {
        const int rr = k*4 + wv, co = rr/TR, r = rr - co*TR;
        const int yy = y0 + r, xx = xb + lane;
        vg[k] = (yy < h && xx < w) ? gy[(((size_t)b*CO + co)*h + yy)*w + xx] : 0.f;      // (beyond the image: zeros, those pixels add nothing)
      }
#pragma unroll
      for (int k = 0; k < kX; ++k) {
        const int rr = k*4 + wv, c = rr/(TR + 2), r = rr - c*(TR + 2);
        const float* rowp = xp + (((size_t)b*CI + c)*H + min(y0 + r, H - 1))*W;
        vx0[k] = rowp[min(xb + lane, W - 1)];
        vx1[k] = lane < 2 ? rowp[min(xb + 64 + lane, W - 1)] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < kG; ++k) { const int rr = k*4 + wv, co = rr/TR, r = rr - co*TR; tg[co*GS + r*64 + lane] = vg[k]; }
#pragma unroll
      for (int k = 0; k < kX; ++k) {
        const int rr = k*4 + wv, c = rr/(TR + 2), r = rr - c*(TR + 2);
        tx[c*XS + r*LW + lane] = vx0[k];
        if (lane < 2) tx[c*XS + r*LW + 64 + lane] = vx1[k];
      }
    }
    __syncthreads();
    const float* pa = tg + i*GS + wv*64 + q;
    const float* pb = tx + i*XS + wv*LW + q;
#pragma unroll 4
    for (int xs = 0; xs < 16; ++xs) {
      const float a = pa[xs*4];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int g = 0; g < NG; ++g)
          acc[tap*NG + g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, pb[(size_t)g*16*XS + (tap/3)*LW + xs*4 + tap % 3], acc[tap*NG + g], 0, 0, 0);
    }
    __syncthreads();                               // before the next tile overwrites this one
  }
  // (requesting the next tile's rows before this tile's MFMAs costs 64 more live registers: 252 VGPRs, one wave per SIMD, 140 us instead of 113)
  // the block's sums: D[row = 4 q + v -> co][col = i -> c] of tile (tap, g)
#pragma unroll
  for (int t = 0; t < NTL; ++t)
#pragma unroll
    for (int v = 0; v < 4; ++v) lds[((wv*NTL + t)*4 + v)*64 + lane] = acc[t][v];
  __syncthreads();
  const size_t nblk = (size_t)gridDim.x*gridDim.y*gridDim.z;
  const size_t blk = ((size_t)blockIdx.z*gridDim.y + blockIdx.y)*gridDim.x + blockIdx.x;
  for (int e = threadIdx.x; e < NTL*4*64; e += 256) {
    const float sum = (lds[e] + lds[NTL*256 + e]) + (lds[2*NTL*256 + e] + lds[3*NTL*256 + e]);
    const int l = e & 63, v = (e >> 6) & 3, t = e >> 8;
    const int co = (l >> 4)*4 + v, c = (t % NG)*16 + (l & 15), tap = t/NG;
    partial[(((size_t)co*CI + c)*9 + tap)*nblk + blk] = sum;
  }
}

// one wave per weight: the fp64 sum of its T partial sums in block order
__global__ __launch_bounds__(256) void k_thin_wgt_finalize(const float* __restrict__ partial, unsigned T, int n, float* __restrict__ g_w) {
  const int i = blockIdx.x*4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  double acc = 0.0;
  for (unsigned t = lane; t < T; t += 64) acc += (double)partial[(size_t)i*T + t];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) g_w[i] = (float)acc;
}

static inline unsigned thin_wgt_blocks(int B, int h, int w) { return (unsigned)(ceil_div(w, 64)*ceil_div(h, kThinWgtRows)*B); }
size_t conv_thin_partials(int B, int C, int h, int w) { return (size_t)kThinCO*C*9*thin_wgt_blocks(B, h, w); }

hipError_t launch_conv_thin_bwd_wgt(const float* xp, const float* gy, float* g_w, float* partial, int B, int C, int h, int w, hipStream_t st) {
  const dim3 grid(ceil_div(w, 64), ceil_div(h, kThinWgtRows), B);
  if (C == 16) hipLaunchKernelGGL((k_thin_wgt_mfma<16>), grid, dim3(256), 0, st, xp, gy, partial, h, w);
  else hipLaunchKernelGGL((k_thin_wgt_mfma<32>), grid, dim3(256), 0, st, xp, gy, partial, h, w);
  hipLaunchKernelGGL(k_thin_wgt_finalize, dim3(ceil_div(kThinCO*C*9, 4)), dim3(256), 0, st, partial, thin_wgt_blocks(B, h, w), kThinCO*C*9, g_w);
  return hipGetLastError();
}

hipError_t launch_conv_thin_fwd(const float* xp, const float* wgt, float* y, int B, int C, int h, int w, hipStream_t st) {
  if (C == 16) hipLaunchKernelGGL((k_thin_mfma<16, 1, false>), dim3(ceil_div(w, 64), ceil_div(h, ThinTile<16, 1, false>::TR), B), dim3(256), 0, st, xp, wgt, y, h + 2, w + 2, h, w);
  else hipLaunchKernelGGL((k_thin_mfma<32, 1, false>), dim3(ceil_div(w, 64), ceil_div(h, ThinTile<32, 1, false>::TR), B), dim3(256), 0, st, xp, wgt, y, h + 2, w + 2, h, w);
  return hipGetLastError();
}

// g_xp (B, C, h + 2, w + 2) from g_y (B, 16, h, w): C = 16 or 32
hipError_t launch_conv_thin_bwd_data(const float* gy, const float* wgt, float* g_xp, int B, int C, int h, int w, hipStream_t st) {
  if (C == 16) hipLaunchKernelGGL((k_thin_mfma<16, 1, true>), dim3(ceil_div(w + 2, 64), ceil_div(h + 2, ThinTile<16, 1, true>::TR), B), dim3(256), 0, st, gy, wgt, g_xp, h, w, h + 2, w + 2);
  else hipLaunchKernelGGL((k_thin_mfma<16, 2, true>), dim3(ceil_div(w + 2, 64), ceil_div(h + 2, ThinTile<16, 2, true>::TR), B), dim3(256), 0, st, gy, wgt, g_xp, h, w, h + 2, w + 2);
  return hipGetLastError();
}

}  // namespace smd
