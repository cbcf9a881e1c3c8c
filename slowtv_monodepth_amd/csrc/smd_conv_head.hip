// smd_conv_head.hip — the output heads of the Monodepth decoder: reflect-pad -> conv3x3(C -> 1) -> sigmoid (SURVEY.md §8f rank 4;
// reference: src/networks/decoders/monodepth.py:52, 86-87 — `self.out[i] = conv3x3(num_ch_dec[i], out_ch)`, `out[i] = self.act(self.out[i](x))`,
// conv3x3 = nn.Conv2d(cin, cout, 3, padding=1, padding_mode='reflect'), decoders/utils.py:44-46).
//
// A convolution with ONE output channel is a stencil, not a GEMM: per output pixel 9 C multiply-adds on 4 C bytes of input that nobody else needs
// again (36 B of traffic per 18 C flop) — HBM-bound by a wide margin.  MIOpen serves it with the kernels it has for wide layers: at cfg 2's full
// resolution (16 -> 1 at 192x640, b = 12) 177 us forward and 334 us backward for 101 MB, 0.6 TB/s (scripts/dev/decoder_conv_times.py); the four heads
// of the decoder together cost 0.33 ms forward and 0.65 ms backward per training step, 6 % of it.  Here the head is three streaming kernels that
// read the already padded activation the glue kernels of smd_decoder.hip leave (it is shared with the next stage's convolution) exactly once:
//   k_head_fwd      y = act(bias + sum_c sum_3x3 w[c,ky,kx] xp[c, i+ky, j+kx])          a thread owns one column of kHeadRows output rows
//   k_head_bwd_data g_xp[c, p, q] = sum_3x3 w[c,ky,kx] gp[p-ky, q-kx]                    a thread owns one padded position, all channels: store-bound
//   k_head_bwd_wgt  g_w[c,ky,kx] = sum_pixels gp[i,j] xp[c, i+ky, j+kx];  g_bias = sum gp  per-(channel, tile column, sample) partial sums, then a fixed-order fp64 sum
// with gp = g_y * act'(y) recomputed from the saved output (sigmoid: y (1 - y)).  Deterministic (no atomics), fp32 arithmetic in a fixed order.
#include "smd_common.h"
#include "smd_kernels.h"

namespace smd {

constexpr int kHeadRows = 4;                   // output rows per thread: (kHeadRows + 2) x 3 loads feed 9 kHeadRows multiply-adds per channel
constexpr int kHeadTileW = 64, kHeadTileH = 4*kHeadRows;   // a block of 256 threads: 64 columns x 4 row groups
constexpr long long kHeadEnoughWaves = 4096;   // four generations of waves on the chip: below that a launch is a chain of latencies, split the channels as well

__device__ __forceinline__ float head_act(float v, int act) { return act == 1 ? 1.f/(1.f + __expf(-v)) : v; }

// Three consecutive elements of a padded row for a thread's column.  bfloat16 with an even row pitch (every decoder level: w is even): the two ALIGNED dwords that
// hold them and a funnel shift by the column's parity, the aligned base and the parity computed once per thread (2-byte loads made the bf16 forward three times
// slower than the fp32 one on half the bytes: 152 vs 50 us at 16 -> 1, 384x640).
template <typename TX, bool DW> struct Row3 {             // DW: bfloat16 rows read as aligned dwords (chosen at launch: even pitch); no run-time branch around a load
  const TX* base; size_t e0; unsigned par;                // (offsets from the tensor's base, a 4-byte-aligned kernel argument: the pointer never passes through an integer —
                                                          // that made every load a flat_load with vmcnt(0) behind it and cost 222 registers)
  __device__ __forceinline__ Row3(const TX* base_, size_t e0_) : base(base_), e0(e0_), par((unsigned)(e0_ & 1)) {}
  __device__ __forceinline__ void next_channel(size_t elems) { e0 += elems; }    // (an even number of elements: the parity stays)
  // the loads of a channel's rows are all issued before the first is converted (left to the scheduler the dword form waited after every row: 23 waits per three
  // channels where the fp32 form has 5)
  __device__ __forceinline__ void request(size_t row_off, unsigned& d0, unsigned& d1) const {
    const unsigned* pw = reinterpret_cast<const unsigned*>(base) + (e0 >> 1) + (row_off >> 1);   // (the pitch is even: the lane's part and the wave-uniform row part separate)
    d0 = pw[0]; d1 = pw[1];
  }
  __device__ __forceinline__ void unpack(unsigned d0, unsigned d1, float& a, float& b, float& c) const {
    const unsigned r = __builtin_amdgcn_alignbit(d1, d0, par*16), t = d1 >> (par*16);
    a = __builtin_bit_cast(float, r << 16); b = __builtin_bit_cast(float, r & 0xffff0000u); c = __builtin_bit_cast(float, t << 16);
  }
  __device__ __forceinline__ void load(size_t row_off, float& a, float& b, float& c) const {   // row_off: elements from the thread's first row (a multiple of the pitch)
    if constexpr (DW) { unsigned d0, d1; request(row_off, d0, d1); unpack(d0, d1, a, b, c); }
    else { a = ld_as_float<TX>(base, e0 + row_off); b = ld_as_float<TX>(base, e0 + row_off + 1); c = ld_as_float<TX>(base, e0 + row_off + 2); }
  }
};

// R output rows per thread.  SPLIT = false: the block's four waves are four row groups (a tile of 64 x 4R outputs), every wave walks all channels.
// SPLIT = true (small images, many channels: the coarse pyramid levels): the four waves share ONE row group (64 x R outputs) and take the channels
// c = wave, wave + 4, ...; their partial sums meet in LDS and wave 0 adds them in wave order.
// TX: the padded activation's element type (float, or bf16 under bf16 autocast: the glue kernels then write bf16 — half the bytes of an HBM-bound kernel;
// weights, bias, the output and every sum stay fp32).
template <int R, bool SPLIT, typename TX, bool DW>
__global__ __launch_bounds__(256) void k_head_fwd(const TX* __restrict__ xp, const float* __restrict__ wgt, const float* __restrict__ bias, float* __restrict__ y,
                                                  int C, int h, int w, int act) {
  __shared__ float red[SPLIT ? 3 : 1][R][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int x = blockIdx.x*kHeadTileW + lane, y0 = SPLIT ? blockIdx.y*R : (blockIdx.y*4 + wv)*R, b = blockIdx.z;
  const bool live = x < w && y0 < h;
  const int W = w + 2, H = h + 2;
  const int rows = live ? min(R, h - y0) : 0;
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.f;
  if (live) {
    const int c0 = SPLIT ? wv : 0, cs = SPLIT ? 4 : 1;
    Row3<TX, DW> rp(xp, (((size_t)b*C + c0)*H + y0)*W + x);
#pragma unroll SPLIT ? 4 : 2   // (a split wave has few outputs and a long chain: more channels' loads in flight)
    for (int c = c0; c < C; c += cs, rp.next_channel((size_t)cs*H*W)) {
      const float* wc = wgt + c*9;               // wave-uniform: scalar loads
      const float w00 = wc[0], w01 = wc[1], w02 = wc[2], w10 = wc[3], w11 = wc[4], w12 = wc[5], w20 = wc[6], w21 = wc[7], w22 = wc[8];
      float v[R + 2][3];
      if constexpr (DW) {
        unsigned d0[R + 2], d1[R + 2];
#pragma unroll
        for (int r = 0; r < R + 2; ++r) rp.request((size_t)min(r, rows + 1)*W, d0[r], d1[r]);   // (rows beyond the image's last: the last one again, unused)
#pragma unroll
        for (int r = 0; r < R + 2; ++r) {
          rp.unpack(d0[r], d1[r], v[r][0], v[r][1], v[r][2]);
          if (r >= rows + 2) v[r][0] = v[r][1] = v[r][2] = 0.f;
        }
      } else {
#pragma unroll
        for (int r = 0; r < R + 2; ++r) {
          if (r < rows + 2) rp.load((size_t)r*W, v[r][0], v[r][1], v[r][2]);   // (rows beyond the image's last: not read)
          else v[r][0] = v[r][1] = v[r][2] = 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float s = acc[r];
        s = fmaf(w00, v[r][0], s); s = fmaf(w01, v[r][1], s); s = fmaf(w02, v[r][2], s);
        s = fmaf(w10, v[r + 1][0], s); s = fmaf(w11, v[r + 1][1], s); s = fmaf(w12, v[r + 1][2], s);
        s = fmaf(w20, v[r + 2][0], s); s = fmaf(w21, v[r + 2][1], s); s = fmaf(w22, v[r + 2][2], s);
        acc[r] = s;
      }
    }
  }
  if (SPLIT) {
    if (wv > 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) red[wv - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wv > 0) return;
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
  }
  const float bc = bias ? bias[0] : 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) if (r < rows) y[((size_t)b*h + y0 + r)*w + x] = head_act(acc[r] + bc, act);
}

// gp = g_y * act'(y) at (i, j) of sample b, zero outside the image
__device__ __forceinline__ float head_gp(const float* __restrict__ gy, const float* __restrict__ y, size_t base, int i, int j, int h, int w, int act) {
  if (i < 0 || i >= h || j < 0 || j >= w) return 0.f;
  const float g = gy[base + (size_t)i*w + j];
  if (act != 1) return g;
  const float s = y[base + (size_t)i*w + j];
  return g*s*(1.f - s);
}

// grid (ceil(W/64), ceil(H/4), B * G): channel group blockIdx.z % G takes channels [g * Cg, (g + 1) * Cg)
template <typename TX, bool DW>
__global__ __launch_bounds__(256) void k_head_bwd_data(const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ wgt, TX* __restrict__ g_xp,
                                                       int C, int h, int w, int G, int Cg, int act) {
  const int W = w + 2, H = h + 2;
  const int q = blockIdx.x*64 + (threadIdx.x & 63), p = blockIdx.y*4 + (threadIdx.x >> 6), b = blockIdx.z/G, g = blockIdx.z - b*G;
  constexpr bool pairs = DW;                                    // bfloat16, even pitch: an even lane stores its neighbour's value with its own, one dword
  if (q >= W || p >= H) return;                                 // (W even: the last column W - 1 is odd, its partner W - 2 is a live lane of the same wave)
  const size_t base = (size_t)b*h*w;
  float nb[3][3];                              // nb[ky][kx] = gp[p - ky, q - kx]: the outputs whose window holds padded position (p, q) at (ky, kx)
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) nb[ky][kx] = head_gp(gy, y, base, p - ky, q - kx, h, w, act);
  const int c0 = g*Cg, c1 = min(c0 + Cg, C);
  TX* o = g_xp + (((size_t)b*C + c0)*H + p)*W + q;
  for (int c = c0; c < c1; ++c, o += (size_t)H*W) {
    const float* wc = wgt + c*9;
    float s = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) s = fmaf(wc[ky*3 + kx], nb[ky][kx], s);
    if constexpr (sizeof(TX) == 2) {
      if (pairs) {                                              // (lanes 2 j, 2 j + 1 are columns 2 j, 2 j + 1 of one row: W is even, so a pair never straddles rows)
        const float nb = __shfl_down(s, 1, 64);
        if ((q & 1) == 0) {
          typedef float f2 __attribute__((ext_vector_type(2))); typedef __bf16 b2 __attribute__((ext_vector_type(2)));
          const f2 pr = {s, nb};
          *reinterpret_cast<unsigned*>(o) = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, b2));
        }
      } else st_from_float<TX>(o, 0, s);
    } else st_from_float<TX>(o, 0, s);
  }
}

// grid (tiles_x, B * chunks_y, C + 1): a block walks kHeadWgtTiles vertically adjacent 64 x 16 tiles of its (tile column, sample, channel) and leaves ONE
// set of nine sums (one tile per block: the block's reduction costs as much as its sums; a whole column: too few waves for the loads' latency); channel C
// is the bias' job (sum of gp).  partial[(c * T + blockIdx.y * tiles_x + blockIdx.x) * 9 + k], T = tiles_x * B * chunks_y.
constexpr int kHeadWgtTiles = 3;
template <typename TX, bool DW>
__global__ __launch_bounds__(256) void k_head_bwd_wgt(const TX* __restrict__ xp, const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ partial,
                                                      int C, int h, int w, int chunks_y, int act) {
  __shared__ float red[4][9];
  const int W = w + 2, H = h + 2;
  const int b = blockIdx.y/chunks_y, chunk = blockIdx.y - b*chunks_y, c = blockIdx.z;
  const int x = blockIdx.x*kHeadTileW + (threadIdx.x & 63);
  float acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.f;
  const size_t base = (size_t)b*h*w;
  if (x < w) {
    const int ylo = chunk*kHeadWgtTiles*kHeadTileH, yhi = min(ylo + kHeadWgtTiles*kHeadTileH, h);
    for (int y0 = ylo + (threadIdx.x >> 6)*kHeadRows; y0 < yhi; y0 += kHeadTileH) {
      const int rows = min(kHeadRows, h - y0);
      float g[kHeadRows];
#pragma unroll
      for (int r = 0; r < kHeadRows; ++r) g[r] = r < rows ? head_gp(gy, y, base, y0 + r, x, h, w, act) : 0.f;
      if (c == C) {
#pragma unroll
        for (int r = 0; r < kHeadRows; ++r) acc[0] += g[r];
      } else {
        const Row3<TX, DW> rp(xp, (((size_t)b*C + c)*H + y0)*W + x);
        float v[kHeadRows + 2][3];
        if constexpr (DW) {
          unsigned d0[kHeadRows + 2], d1[kHeadRows + 2];
#pragma unroll
          for (int r = 0; r < kHeadRows + 2; ++r) rp.request((size_t)min(r, rows + 1)*W, d0[r], d1[r]);
#pragma unroll
          for (int r = 0; r < kHeadRows + 2; ++r) {
            rp.unpack(d0[r], d1[r], v[r][0], v[r][1], v[r][2]);
            if (r >= rows + 2) v[r][0] = v[r][1] = v[r][2] = 0.f;
          }
        } else {
#pragma unroll
          for (int r = 0; r < kHeadRows + 2; ++r) {
            if (r < rows + 2) rp.load((size_t)r*W, v[r][0], v[r][1], v[r][2]);
            else v[r][0] = v[r][1] = v[r][2] = 0.f;
          }
        }
#pragma unroll
        for (int r = 0; r < kHeadRows; ++r)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[ky*3 + kx] = fmaf(g[r], v[r + ky][kx], acc[ky*3 + kx]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const float t = wave_sum(acc[k]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = t;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const size_t T = (size_t)gridDim.x*gridDim.y;
    const size_t tile = (size_t)blockIdx.y*gridDim.x + blockIdx.x;
    partial[((size_t)c*T + tile)*9 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

// g_w[c*9 + k] (c < C) and g_bias = fp64 sums of the blocks' partial sums in block order; one wave per channel
__global__ __launch_bounds__(64) void k_head_wgt_finalize(const float* __restrict__ partial, unsigned T, int C, float* __restrict__ g_w, float* __restrict__ g_bias) {
  const int c = blockIdx.x;
  const int nk = c == C ? 1 : 9;
  for (int k = 0; k < nk; ++k) {
    double acc = 0.0;
    for (unsigned t = threadIdx.x; t < T; t += 64) acc += (double)partial[((size_t)c*T + t)*9 + k];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (threadIdx.x == 0) {
      if (c == C) { if (g_bias) g_bias[0] = (float)acc; }
      else g_w[c*9 + k] = (float)acc;
    }
  }
}

static inline int head_wgt_chunks(int h) { return ceil_div(ceil_div(h, kHeadTileH), kHeadWgtTiles); }
size_t conv_head_partials(int B, int C, int h, int w) { return (size_t)(C + 1)*ceil_div(w, kHeadTileW)*B*head_wgt_chunks(h)*9; }

template <typename TX, bool DW>
static void head_fwd_t(const TX* xp, const float* wgt, const float* bias, float* y, int B, int C, int h, int w, int act, hipStream_t st) {
  const int tx = ceil_div(w, kHeadTileW);
  if ((long long)B*tx*ceil_div(h, kHeadTileH)*4 >= kHeadEnoughWaves || C < 8)
    hipLaunchKernelGGL((k_head_fwd<kHeadRows, false, TX, DW>), dim3(tx, ceil_div(h, kHeadTileH), B), dim3(256), 0, st, xp, wgt, bias, y, C, h, w, act);
  else if ((long long)B*tx*ceil_div(h, 2)*4 >= kHeadEnoughWaves)
    hipLaunchKernelGGL((k_head_fwd<2, true, TX, DW>), dim3(tx, ceil_div(h, 2), B), dim3(256), 0, st, xp, wgt, bias, y, C, h, w, act);
  else
    hipLaunchKernelGGL((k_head_fwd<1, true, TX, DW>), dim3(tx, h, B), dim3(256), 0, st, xp, wgt, bias, y, C, h, w, act);
}
hipError_t launch_conv_head_fwd(const void* xp, int x_bf16, const float* wgt, const float* bias, float* y, int B, int C, int h, int w, int act, hipStream_t st) {
  if (x_bf16 && (w & 1) == 0) head_fwd_t<bf16, true>((const bf16*)xp, wgt, bias, y, B, C, h, w, act, st);      // (every decoder level has an even width)
  else if (x_bf16) head_fwd_t<bf16, false>((const bf16*)xp, wgt, bias, y, B, C, h, w, act, st);
  else head_fwd_t<float, false>((const float*)xp, wgt, bias, y, B, C, h, w, act, st);
  return hipGetLastError();
}

template <typename TX, bool DW>
static void head_bwd_t(const TX* xp, const float* wgt, const float* y, const float* gy, TX* g_xp, float* g_w, float* g_bias, float* partial,
                       int B, int C, int h, int w, int act, hipStream_t st) {
  if (g_xp) {
    const long long waves = (long long)B*ceil_div(w + 2, 64)*ceil_div(h + 2, 4)*4;
    int G = (int)((kHeadEnoughWaves + waves - 1)/waves);
    if (G > C) G = C;
    if (G < 1) G = 1;
    if ((long long)B*G > 65535) G = 65535/B > 0 ? 65535/B : 1;
    const int Cg = ceil_div(C, G);
    G = ceil_div(C, Cg);
    hipLaunchKernelGGL((k_head_bwd_data<TX, DW>), dim3(ceil_div(w + 2, 64), ceil_div(h + 2, 4), B*G), dim3(256), 0, st, gy, y, wgt, g_xp, C, h, w, G, Cg, act);
  }
  if (g_w) {
    const int tx = ceil_div(w, kHeadTileW);
    const int cy = head_wgt_chunks(h);
    hipLaunchKernelGGL((k_head_bwd_wgt<TX, DW>), dim3(tx, B*cy, C + 1), dim3(256), 0, st, xp, gy, y, partial, C, h, w, cy, act);
    hipLaunchKernelGGL(k_head_wgt_finalize, dim3(C + 1), dim3(64), 0, st, partial, (unsigned)(tx*B*cy), C, g_w, g_bias);
  }
}
hipError_t launch_conv_head_bwd(const void* xp, int x_bf16, const float* wgt, const float* y, const float* gy, void* g_xp, float* g_w, float* g_bias, float* partial,
                                int B, int C, int h, int w, int act, hipStream_t st) {
  if (x_bf16 && (w & 1) == 0) head_bwd_t<bf16, true>((const bf16*)xp, wgt, y, gy, (bf16*)g_xp, g_w, g_bias, partial, B, C, h, w, act, st);
  else if (x_bf16) head_bwd_t<bf16, false>((const bf16*)xp, wgt, y, gy, (bf16*)g_xp, g_w, g_bias, partial, B, C, h, w, act, st);
  else head_bwd_t<float, false>((const float*)xp, wgt, y, gy, (float*)g_xp, g_w, g_bias, partial, B, C, h, w, act, st);
  return hipGetLastError();
}

}  // namespace smd
