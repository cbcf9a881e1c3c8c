// smd_conv_mfma.hip — the Monodepth decoder's wide 3x3 convolutions on the bf16 matrix cores with fp32-class results (SURVEY.md §8f rank 4, round 6;
// reference: src/networks/decoders/monodepth.py:40-50, 71-84 — `ConvELU(cin, cout)` = reflection-padded conv3x3 + ELU; decoders/utils.py:44-54).
//
// Why not the f32 MFMA (smd_conv_thin.hip): on gfx950 `v_mfma_f32_32x32x2_f32` runs at the vector rate, 157 TFLOP/s — 1/16 of the bf16 matrix rate — and
// there is no xf32/TF32 form.  MIOpen's fp32 Winograd reaches 87-93 TFLOP/s effective on these layers forward and 55-75 backward.  Here every fp32 operand
// is SPLIT EXACTLY into three bf16 pieces, x = x0 + x1 + x2 (8 + 8 + 8 significant bits: x0 = bf16(x), x1 = bf16(x - x0), x2 = x - x0 - x1, each
// difference exact in fp32), and a product a.b is formed as the six bf16 products with i + j <= 2,
//     a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0),
// each exact in the fp32 accumulator's product stage; what is dropped (a1 b2 + a2 b1 + a2 b2) is below 2^-25 |a b|, under the rounding of an fp32
// multiply-add.  Six `v_mfma_f32_32x32x16_bf16` per K step of 16 = 6/16 of the f32 MFMA's time for the same arithmetic: a ceiling of 2.67 x the fp32
// matrix peak with fp32-class error (tests: <= 2e-6 of the tensor's max against fp64 `conv2d`, the same bound the f32-MFMA kernels are held to).
// PIECES = 2 (three products, 16 significant bits, "better than TF32") exists as an experiment knob only; PIECES = 1 is plain bf16.
//
// Forward and data gradient are one kernel (implicit GEMM, M = output channels, N = pixels, K = (tap, input channel)); the weight gradient is a GEMM
// with K = pixels (M = output channels, N = input channels, one accumulator tile per tap).  Operand layouts, per `v_mfma_f32_32x32x16_bf16`:
// A: lane l holds row i = l & 31, K slots 8 (l >> 5) .. + 7; B: column j = l & 31, same K slots; D: column = l & 31, row = (r & 3) + 8 (r >> 2) + 4 (l >> 5).
#include "smd_common.h"
#include "smd_kernels.h"
#include <algorithm>
#include <type_traits>

namespace smd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// the products kept of (a0 + a1 + a2)(b0 + b1 + b2), smallest terms first
__host__ __device__ constexpr int n_products(int P) { return P == 3 ? 6 : P == 2 ? 3 : 1; }
__host__ __device__ constexpr int prod_a(int P, int t) { return P == 3 ? (t == 0 ? 2 : t == 1 ? 1 : t == 2 ? 0 : t == 3 ? 1 : 0) : P == 2 ? (t == 0 ? 1 : 0) : 0; }
__host__ __device__ constexpr int prod_b(int P, int t) { return P == 3 ? (t == 0 ? 0 : t == 1 ? 1 : t == 2 ? 2 : t == 3 ? 0 : t == 4 ? 1 : 0) : P == 2 ? (t == 1 ? 1 : 0) : 0; }

__device__ __forceinline__ unsigned pack_bf16_rne(float a, float b) { const f32x2v v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v)); }   // v_cvt_pk_bf16_f32
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
// two fp32 values -> P dwords of two bf16 each (low half = a's piece, high half = b's)
template <int P> __device__ __forceinline__ void split_pair(float a, float b, unsigned (&p)[P]) {
  p[0] = pack_bf16_rne(a, b);
  if constexpr (P > 1) { a -= bf16_lo(p[0]); b -= bf16_hi(p[0]); p[1] = pack_bf16_rne(a, b); }
  if constexpr (P > 2) { a -= bf16_lo(p[1]); b -= bf16_hi(p[1]); p[2] = pack_bf16_rne(a, b); }
}
__device__ __forceinline__ bf16x8 as_frag(const uint4& u) { return __builtin_bit_cast(bf16x8, u); }
// bfloat16 tensors (the decoder under bf16 autocast, `pieces` = 1): an element IS its one piece — loaded as 16 raw bits, two of them a dword
template <int P> __device__ __forceinline__ void split_pair(unsigned short a, unsigned short b, unsigned (&p)[P]) {
  static_assert(P == 1, "bfloat16 operands have one piece");
  p[0] = (unsigned)a | ((unsigned)b << 16);
}
template <typename T> struct RawOf { typedef float type; };                  // what a staging load leaves in a register
template <> struct RawOf<bf16> { typedef unsigned short type; };
template <typename T> __device__ __forceinline__ void store_out(T* p, size_t i, float v);
template <> __device__ __forceinline__ void store_out<float>(float* p, size_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void store_out<bf16>(bf16* p, size_t i, float v) { p[i] = __float2bfloat16(v); }

// ---- weights -> bf16 pieces in the A-operand order of both convolution forms (one launch per layer and step; the backward reads what the forward packed) ----
// forward form:       Wt[m = co][k = c][tap]      = w[co][c][tap]          (M = CO, CK = C)
// data-gradient form: Wt[m = c][k = co][tap]      = w[co][c][8 - tap]      (M = C, CK = CO): the convolution of the zero-extended dL/dy with the flipped kernel
// element (m, k, tap, piece) at ((((m >> 5) KC + (k >> 4)) 9 + tap) P + piece) 512 + (((k >> 3) & 1) 32 + (m & 31)) 8 + (k & 7): a wave's fragment is 1 KiB, lane-major
template <int P>
__global__ __launch_bounds__(256) void k_conv_pack_w(const float* __restrict__ w, unsigned short* __restrict__ wp_fwd, unsigned short* __restrict__ wp_bwd, int CO, int C) {
  const int idx = blockIdx.x*256 + threadIdx.x;
  if (idx >= CO*C*9) return;
  const int tap = idx % 9, c = (idx/9) % C, co = idx/(9*C);
  unsigned pk[P];
  split_pair<P>(w[idx], 0.f, pk);
  if (wp_fwd) {
    const size_t base = ((((size_t)(co >> 5)*(C >> 4) + (c >> 4))*9 + tap)*P)*512 + (((c >> 3) & 1)*32 + (co & 31))*8 + (c & 7);
#pragma unroll
    for (int p = 0; p < P; ++p) wp_fwd[base + (size_t)p*512] = (unsigned short)(pk[p] & 0xffffu);
  }
  if (wp_bwd) {
    const size_t base = ((((size_t)(c >> 5)*(CO >> 4) + (co >> 4))*9 + (8 - tap))*P)*512 + (((co >> 3) & 1)*32 + (c & 31))*8 + (co & 7);
#pragma unroll
    for (int p = 0; p < P; ++p) wp_bwd[base + (size_t)p*512] = (unsigned short)(pk[p] & 0xffffu);
  }
}

// ---- forward / data gradient ----
// out[b][m][y][x] = sum over k < CK and taps of in[b][k][y + ky - off][x + kx - off] Wt[m][k][tap]; off = 0 for the forward (in = the reflection-padded input,
// every read of a stored output is inside it), off = 2 for the data gradient (in = dL/dy, zero outside; out = the gradient of the PADDED input).
// A block of four waves owns 8 pixel tiles of 32 consecutive pixels (TC = 64: 4 rows x 64 columns, a wave per row; TC = 32: 8 rows x 32 columns, a wave
// per row pair) and one tile of 32 output channels; K runs in chunks of 16 input channels x 9 taps.  Per chunk the block stages its (TRB + 2) x (TC + 2)
// patch of the 16 channels in LDS — coalesced row pieces, split into the bf16 pieces ONCE per element (it is used by 9 taps x every output channel),
// filed pixel-major with the 16 channels of a pixel contiguous (32 B per piece): the B operand of a tap is then one ds_read_b128 per lane whatever the tap's
// shift.  The two 16-byte halves of a pixel are swapped where bit 3 of the pixel index is set: the 16-lane groups that serve a ds_read_b128
// ({0-3, 12-15, 20-27}, ...) then cover all 64 banks instead of colliding two ways.  Two patches: the next chunk's loads are requested after tap 4's fetch
// and filed behind the MFMAs of taps 6 - 8, one barrier per chunk.  The weights come as whole fragments from `k_conv_pack_w`'s image (1 KiB per wave and piece, lane-major:
// every block reads the same ones, L1 / L2 hits), five slots requested four taps ahead (see below why).
// What was measured on the way to this form (cfg 2's 96 -> 32 layer at 96x320, MIOpen 217 us; scripts/dev/conv_mfma_check.py, profiles/r06_conv_mfma_*):
// one load per loop trip, 134 us; every staging load before its first use + the ring, 135 (the pieces — MFMAs alone 59 us, operand reads 40, staging 79-97 —
// hardly overlap: a CU's one memory pipeline carries the staging loads AND four waves' copies of the weight fragments, 133 KB per chunk and block);
// weights through LDS as well (one patch + one weight image, two barriers per chunk), 171; the chunk's loads spread over the taps, 205 (284 registers: one
// wave per SIMD); two producer waves + four MFMA waves per block, 158-174; chunks of 8 channels (two taps per MFMA K step) so that patch AND weight fragments
// fit twice in 68 KB and the MFMA loop touches no vector memory, 158.  What the series says: the limiter is the bytes a CU pulls through its vector-memory path
// (~10 B/clk for L2 / HBM data: 25 KB of patch per chunk and block = 2.5 k cycles against 3.5 k of MFMAs for 32 output channels) — weight fragments fetched
// once per block from L2 cost more than four waves' L1-hit copies.  Later in the round (profiles/r06_conv_mfma_ablations.txt): the in-order load counter (a fragment wait
// drained the staging loads: fragments four taps ahead, staging request behind the chunk's last fetch, -5 %), a phase trace (scripts/dev/conv_trace.py), eight waves with
// the weights in LDS and persistent blocks (both slower).  The lever left is MFMA work per wave and staged byte: two or three channel tiles per patch.
#ifdef SMD_CONV_TRACE   // diagnosis builds only (scripts/dev/conv_trace.py): shader-clock stamps of wave 0 of every block of the forward / data-gradient form
__device__ unsigned long long g_conv_trace[8192][40];
#define SMD_CT(slot) do { if (wall == 0 && lid < 8192u && (slot) < 40) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_conv_trace[lid][(slot)] = t_; } } while (0)
#else
#define SMD_CT(slot) do { } while (0)
#endif
template <int TC> struct ConvTile {
  static constexpr int TRB = (TC == 64) ? 4 : 8;
  static constexpr int PW = TC + 2, PH = TRB + 2, NPIX = PW*PH;
};

// NM: channel tiles per block.  NM = 2 (knob conv_two_tiles, layers with a multiple of 64 output channels): EIGHT waves, waves 4 .. 7 multiply the same patch by
// the next 32 output channels' weights — the patch is staged once for 64 channels (by all 512 lanes).  Built to halve what a block pulls through the CU's
// vector-memory path per MFMA; measured neutral (128 -> 64 at 48x160: 103.0 vs 104.7 us forward, 108.0 vs 104.7 data gradient; 512 -> 256 at 12x40: 183 vs 171;
// 128 -> 64 at 24x80 data gradient 42.7 vs 52.3), so the default stays one tile per block.  Same bits either way.
template <int TC, int P, bool BWD, typename TI, typename TO, int NM>
__global__ __launch_bounds__(256*NM, 2/NM) void k_conv_mfma(const TI* __restrict__ in_, const uint4* __restrict__ wp, TO* __restrict__ out,
                                                   int CK, int M, int hi, int wi, int ho, int wo, int KS, int kc_per_split, size_t split_stride,
                                                   unsigned gx, unsigned gy, unsigned gz) {
  using T = ConvTile<TC>;
  constexpr int NPIX = T::NPIX, PW = T::PW, off = BWD ? 2 : 0, NPROD = n_products(P);
  constexpr int kBuf = P*NPIX*2;
  __shared__ uint4 tile[2*kBuf];                                  // two patches, [piece][pixel][half]
  const int lane = threadIdx.x & 63, wall = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wv = wall & 3, mt = wall >> 2, j = lane & 31, g = lane >> 5;
  constexpr int NT = 256*NM;
  // Block -> (channel tile, K split, tile column, tile row, sample), XCD-aware: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs, each
  // with its own L2.  Here XCD k works through the k-th eighth of the tile list in order, so the blocks in flight on an XCD are neighbours in the image —
  // the halo rows / columns two tiles share, and the one patch the channel tiles of a pixel tile all read, come from HBM once (the natural order sends every
  // neighbour to another L2: 329 MB fetched for a 145 MB input at cfg 2's 96 -> 32 layer).
  const int MG = M/(32*NM);
  const unsigned nblk = (unsigned)(MG*KS)*gx*gy*gz, per = (nblk + 7)/8;
  const unsigned lid = (blockIdx.x & 7)*per + (blockIdx.x >> 3);
  if (lid >= nblk) return;
  const int mg = (int)(lid % MG)*NM + mt, ks = (lid/MG) % KS;     // mg: this wave's tile of 32 output channels
  const unsigned tl = lid/(MG*KS);
  const int x0 = (int)(tl % gx)*TC, y0 = (int)((tl/gx) % gy)*T::TRB, b = (int)(tl/(gx*gy));
  const int KC = CK >> 4, kc0 = ks*kc_per_split, kc1 = min(KC, kc0 + kc_per_split);
  const size_t plane = (size_t)hi*wi;
  typedef typename RawOf<TI>::type R;
  const R* src = reinterpret_cast<const R*>(in_) + (size_t)b*CK*plane;

  // staging: an item = 8 channels of one patch pixel; its address inside a channel plane does not depend on the chunk
  constexpr int ITEMS = 2*NPIX, TRIPS = (ITEMS + NT - 1)/NT;
  int pofs[TRIPS];                                               // offset inside the plane, or -1: outside (data gradient: zero)
#pragma unroll
  for (int t = 0; t < TRIPS; ++t) {
    const int item = min(t*NT + (int)threadIdx.x, ITEMS - 1);
    const int half = item >= NPIX ? 1 : 0, pix = item - half*NPIX;
    const int r = pix/PW, cc = pix - r*PW;
    const int yy = y0 + r - off, xx = x0 + cc - off;
    if (BWD) pofs[t] = (yy >= 0 && yy < hi && xx >= 0 && xx < wi) ? yy*wi + xx : -1;
    else pofs[t] = min(yy, hi - 1)*wi + min(xx, wi - 1);         // (beyond the image: any valid address, those outputs are not stored)
  }
  R v[TRIPS][8];
  auto request = [&](int kc) {                                    // every load of a chunk is issued before anything waits for one
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
      const int item = min(t*NT + (int)threadIdx.x, ITEMS - 1);
      const int half = item >= NPIX ? 1 : 0;
      const R* p = src + (size_t)(kc*16 + half*8)*plane + (size_t)max(pofs[t], 0);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[t][e] = (!BWD || pofs[t] >= 0) ? p[(size_t)e*plane] : R(0);
    }
  };
  auto file_trip = [&](int buf, int t) {
    {
      const int item = t*NT + (int)threadIdx.x;
      if (item < ITEMS) {
        const int half = item >= NPIX ? 1 : 0, pix = item - half*NPIX;
        unsigned pk[4][P];
#pragma unroll
        for (int q = 0; q < 4; ++q) split_pair<P>(v[t][2*q], v[t][2*q + 1], pk[q]);
        const int slot = pix*2 + (half ^ ((pix >> 3) & 1));
#pragma unroll
        for (int p = 0; p < P; ++p) tile[buf*kBuf + p*NPIX*2 + slot] = uint4{pk[0][p], pk[1][p], pk[2][p], pk[3][p]};
      }
    }
  };
  auto file = [&](int buf) {
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) file_trip(buf, t);
  };

  // the leading product a0 b0 and the five small ones run in accumulators of their own: adding a term 2^-8 or 2^-16 the size of the sum costs a rounding of
  // the SUM's size, so six products in one accumulator carry six times the roundings of one (measured: 2.5e-6 of the output's max at K = 4608 against
  // MIOpen's 6e-7; split: 1.0e-6 against 6e-7 there, at or below MIOpen's elsewhere); the small accumulator's roundings are 2^-8 of that
  f32x16 acc[2], lo[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[nt][r] = 0.f; lo[nt][r] = 0.f; }

  // the weights' fragments: five slots, requested FOUR taps ahead (tap t of a chunk sits in slot t mod 5; a chunk counts as ten steps, the tenth empty, so the
  // positions repeat every chunk), and the next patch's staging loads requested after tap 4's fetch.  The counter of outstanding loads retires in order: a wait
  // for a fragment also waits for every load requested before it.  With a ring of three (two taps ahead) and the staging request at the chunk's start, tap 2's
  // fragments were requested after the staging loads and the wait for them drained those: the staging latency stood exposed in every chunk.  Now the fragments
  // of taps 4 .. 8 are requested before the staging loads and every later fetch belongs to the next chunk, whose first wait comes after the staging loads
  // have been filed anyway: they have four taps to land and are waited for only where they are filed.  (Six slots / nine: 256 registers and 400 / 72 bytes of
  // spills per lane.)  The patch's fragments (LDS) one tap ahead.
  bf16x8 A[5][P], Bf[2][2][P];
  const uint4* wq = wp + ((size_t)mg*KC*9*P)*64 + lane;
  auto fetch_a = [&](bf16x8 (&dst)[P], int kc, int tap) {
#pragma unroll
    for (int p = 0; p < P; ++p) dst[p] = as_frag(wq[(size_t)((kc*9 + tap)*P + p)*64]);
  };
  auto read_b = [&](bf16x8 (&dst)[2][P], int buf, int tap) {
    const int ky = tap/3, kx = tap % 3;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int r = (TC == 64) ? wv : 2*wv + nt, cb = (TC == 64) ? nt*32 : 0;
      const int pix = (r + ky)*PW + cb + j + kx;
      const int slot = pix*2 + (g ^ ((pix >> 3) & 1));
#pragma unroll
      for (int p = 0; p < P; ++p) dst[nt][p] = as_frag(tile[buf*kBuf + p*NPIX*2 + slot]);
    }
  };
  // one chunk; MORE is compile-time (the last chunk is peeled): no load sits behind a run-time branch, the compiler keeps count of what is outstanding
  auto chunk = [&](int kc, int cur, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    [[maybe_unused]] const int cslot = 3 + 4*(kc - kc0);
    SMD_CT(cslot);
    read_b(Bf[0], cur, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap < 8) read_b(Bf[(tap + 1) & 1], cur, tap + 1);
#pragma unroll
      for (int t = 0; t < NPROD; ++t)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          if (t == NPROD - 1) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[tap % 5][0], Bf[tap & 1][nt][0], acc[nt], 0, 0, 0);
          else lo[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[tap % 5][prod_a(P, t)], Bf[tap & 1][nt][prod_b(P, t)], lo[nt], 0, 0, 0);
        }
      if (tap <= 4) fetch_a(A[(tap + 4) % 5], kc, tap + 4);
      else if (MORE && tap >= 6) fetch_a(A[tap - 6], kc + 1, tap - 6);
      if (MORE && tap == 8) fetch_a(A[3], kc + 1, 3);             // (the empty tenth step's fetch; slot 3 is tap 8's, whose MFMAs have been issued)
      if (MORE && tap == 4) request(kc + 1);                      // after this chunk's last fetch: lands during taps 5 .. 8
      // the next patch is filed trip by trip behind the last taps' MFMAs (its splits run in the matrix pipe's shadow) instead of after them
      if (MORE && tap >= 6) {
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) if (t*3/TRIPS == tap - 6) file_trip(cur ^ 1, t);
      }
      // a tap's instructions stay in their tap: left free, the scheduler sinks the fetches of taps 0 - 2 (for taps 4 - 6) behind tap 3's MFMAs and every later
      // fragment is then requested one tap before its use and waited for with vmcnt(0) — the four taps of distance exist in the source only
      __builtin_amdgcn_sched_barrier(0);
    }
    SMD_CT(cslot + 1);
    SMD_CT(cslot + 2);
    __syncthreads();                                              // the other patch is complete, and nobody reads this one any more
    SMD_CT(cslot + 3);
  };

  SMD_CT(0);
  if (kc0 < kc1) {
    request(kc0);
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) fetch_a(A[tap], kc0, tap);
    file(0);
  }
  SMD_CT(1);
  __syncthreads();
  SMD_CT(2);
  int kc = kc0;
  for (; kc + 1 < kc1; ++kc) chunk(kc, (kc - kc0) & 1, std::true_type{});
  if (kc < kc1) chunk(kc, (kc - kc0) & 1, std::false_type{});
  // D[row = output channel][column = pixel]: a register is 32 consecutive pixels of one channel per half wave (128-byte runs)
  // (a K split's partial output is always fp32: `out` is then the workspace the splits' sum reads)
  float* dstf = reinterpret_cast<float*>(out) + (size_t)ks*split_stride;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int r = (TC == 64) ? wv : 2*wv + nt, cb = (TC == 64) ? nt*32 : 0;
    const int y = y0 + r, x = x0 + cb + j;
    if (y < ho && x < wo) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int m = mg*32 + (rr & 3) + 8*(rr >> 2) + 4*g;
        const size_t o = (((size_t)b*M + m)*ho + y)*wo + x;
        if (KS > 1) dstf[o] = acc[nt][rr] + lo[nt][rr];
        else store_out<TO>(out, o, acc[nt][rr] + lo[nt][rr]);
      }
    }
  }
  SMD_CT(39);
}

// ---- sixteen output channels (the decoder's thin last stage: 32 -> 16 at half resolution, 16 -> 16 at full resolution) ----
// The same split-bf16 arithmetic on `v_mfma_f32_16x16x32_bf16` (A: row i = l & 15, K slots 8 (l >> 4) .. + 7; B: column j = l & 15; D: column = l & 15,
// row = 4 (l >> 4) + v), pixels as rows, output channels as columns: a lane ends with four consecutive pixels of one channel — a 16-byte store.  A K step of
// 32 = TWO taps x 16 channels (lane group q: tap 2 s + (q >> 1), channels 8 (q & 1) .. + 7); the ninth tap's partner is a zero weight.  With 2304 multiply-adds
// per pixel on 128 bytes the layer is HBM-bound once its arithmetic costs 6/16 of the f32 MFMA's time (the f32-MFMA kernel of smd_conv_thin.hip: 82 us
// forward at cfg 2, its K loop at half the f32 matrix rate; 189 MB / 5 TB/s = 38 us).  The weights (<= 2 chunks x 5 K steps x P fragments) stay in registers;
// a wave owns 16 columns x 4 rows of a 64 x 4 tile (four accumulator pairs); the patch is staged and read exactly as in k_conv_mfma.
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte store that is only 4-byte aligned (the padded gradient's rows)

// thin operand image: element (chunk, K step s, piece, lane = 16 q + j, e) = Wt[m = j][k = 16 chunk + 8 (q & 1) + e][tap = 2 s + (q >> 1)] (zero for tap 9)
template <int P>
__global__ __launch_bounds__(256) void k_conv_pack_w16(const float* __restrict__ w, unsigned short* __restrict__ wp_fwd, unsigned short* __restrict__ wp_bwd, int C) {
  const int idx = blockIdx.x*256 + threadIdx.x;                   // w is (16, C, 3, 3)
  const int nslots = (C >> 4)*5*64*8;
  if (idx < nslots && wp_fwd) {                                   // forward form: m = co, k = c
    const int e = idx & 7, lane = (idx >> 3) & 63, s = (idx >> 9) % 5, ch = idx/(512*5);
    const int j = lane & 15, q = lane >> 4, tap = 2*s + (q >> 1), k = 16*ch + 8*(q & 1) + e;
    unsigned pk[P];
    split_pair<P>(tap < 9 ? w[((size_t)j*C + k)*9 + tap] : 0.f, 0.f, pk);
#pragma unroll
    for (int p = 0; p < P; ++p) wp_fwd[(((size_t)(ch*5 + s)*P + p)*64 + lane)*8 + e] = (unsigned short)(pk[p] & 0xffffu);
  }
  if (idx < 5*64*8 && wp_bwd && C == 16) {                        // data-gradient form (16 -> 16 only): m = c, k = co, flipped taps
    const int e = idx & 7, lane = (idx >> 3) & 63, s = idx >> 9;
    const int j = lane & 15, q = lane >> 4, tap = 2*s + (q >> 1), k = 8*(q & 1) + e;
    unsigned pk[P];
    split_pair<P>(tap < 9 ? w[((size_t)k*C + j)*9 + (8 - tap)] : 0.f, 0.f, pk);
#pragma unroll
    for (int p = 0; p < P; ++p) wp_bwd[(((size_t)s*P + p)*64 + lane)*8 + e] = (unsigned short)(pk[p] & 0xffffu);
  }
}

template <int NCH, int P, bool BWD, typename TI, typename TO>
__global__ __launch_bounds__(256) void k_conv16_mfma(const TI* __restrict__ in_, const uint4* __restrict__ wp, TO* __restrict__ out,
                                                     int hi, int wi, int ho, int wo, unsigned gx, unsigned gy, unsigned gz) {
  using T = ConvTile<64>;
  constexpr int NPIX = T::NPIX, PW = T::PW, off = BWD ? 2 : 0, NPROD = n_products(P), CK = 16*NCH;
  __shared__ uint4 tile[P*NPIX*2];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 15, q = lane >> 4;
  const unsigned nblk = gx*gy*gz, per = (nblk + 7)/8;             // XCD-aware order, as in k_conv_mfma
  const unsigned lid = (blockIdx.x & 7)*per + (blockIdx.x >> 3);
  if (lid >= nblk) return;
  const int x0 = (int)(lid % gx)*64, y0 = (int)((lid/gx) % gy)*4, b = (int)(lid/(gx*gy));
  const size_t plane = (size_t)hi*wi;
  typedef typename RawOf<TI>::type R;
  const R* src = reinterpret_cast<const R*>(in_) + (size_t)b*CK*plane;

  bf16x8 Wr[NCH][5][P];                                           // every weight fragment of the layer: requested first, used last
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
      for (int p = 0; p < P; ++p) Wr[ch][s][p] = as_frag(wp[((ch*5 + s)*P + p)*64 + lane]);

  constexpr int ITEMS = 2*NPIX, TRIPS = (ITEMS + 255)/256;
  int pofs[TRIPS];
#pragma unroll
  for (int t = 0; t < TRIPS; ++t) {
    const int item = min(t*256 + (int)threadIdx.x, ITEMS - 1);
    const int half = item >= NPIX ? 1 : 0, pix = item - half*NPIX;
    const int r = pix/PW, cc = pix - r*PW;
    const int yy = y0 + r - off, xx = x0 + cc - off;
    if (BWD) pofs[t] = (yy >= 0 && yy < hi && xx >= 0 && xx < wi) ? yy*wi + xx : -1;
    else pofs[t] = min(yy, hi - 1)*wi + min(xx, wi - 1);
  }
  int dpix[5];                                                    // this lane group's tap of K step s, as an offset inside the patch
#pragma unroll
  for (int s = 0; s < 5; ++s) { const int tq = min(2*s + (q >> 1), 8); dpix[s] = (tq/3)*PW + tq % 3; }

  f32x4v acc[4], lo[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { acc[r] = f32x4v{0.f, 0.f, 0.f, 0.f}; lo[r] = f32x4v{0.f, 0.f, 0.f, 0.f}; }

#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    R v[TRIPS][8];
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {                             // every load of the chunk before its first use
      const int item = min(t*256 + (int)threadIdx.x, ITEMS - 1);
      const int half = item >= NPIX ? 1 : 0;
      const R* p = src + (size_t)(ch*16 + half*8)*plane + (size_t)max(pofs[t], 0);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[t][e] = (!BWD || pofs[t] >= 0) ? p[(size_t)e*plane] : R(0);
    }
    if (ch > 0) __syncthreads();                                  // nobody reads the previous chunk any more
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
      const int item = t*256 + (int)threadIdx.x;
      if (item < ITEMS) {
        const int half = item >= NPIX ? 1 : 0, pix = item - half*NPIX;
        unsigned pk[4][P];
#pragma unroll
        for (int k = 0; k < 4; ++k) split_pair<P>(v[t][2*k], v[t][2*k + 1], pk[k]);
        const int slot = pix*2 + (half ^ ((pix >> 3) & 1));
#pragma unroll
        for (int p = 0; p < P; ++p) tile[p*NPIX*2 + slot] = uint4{pk[0][p], pk[1][p], pk[2][p], pk[3][p]};
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 5; ++s) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pix = r*PW + 16*wv + i + dpix[s];
        const int slot = pix*2 + ((q & 1) ^ ((pix >> 3) & 1));
        bf16x8 A[P];
#pragma unroll
        for (int p = 0; p < P; ++p) A[p] = as_frag(tile[p*NPIX*2 + slot]);
#pragma unroll
        for (int t = 0; t < NPROD; ++t) {
          if (t == NPROD - 1) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], Wr[ch][s][0], acc[r], 0, 0, 0);
          else lo[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[prod_a(P, t)], Wr[ch][s][prod_b(P, t)], lo[r], 0, 0, 0);
        }
      }
    }
  }
  // D[row = pixel 4 q + v][column = channel i]: four consecutive pixels of one channel per lane
  const int x = x0 + 16*wv + 4*q;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = y0 + r;
    if (y >= ho || x >= wo) continue;
    TO* dstp = out + (((size_t)b*16 + i)*ho + y)*wo + x;
    const f32x4v o = acc[r] + lo[r];
    if constexpr (sizeof(TO) == 4) {
      if (x + 3 < wo) *reinterpret_cast<f32x4u*>(dstp) = o;
      else { dstp[0] = o[0]; if (x + 1 < wo) dstp[1] = o[1]; if (x + 2 < wo) dstp[2] = o[2]; }
    } else {                                                      // bfloat16: dword stores where the row pitch keeps pixel pairs 4-byte aligned
      const unsigned lo2 = pack_bf16_rne(o[0], o[1]), hi2 = pack_bf16_rne(o[2], o[3]);
      if (x + 3 < wo && (wo & 1) == 0) { reinterpret_cast<unsigned*>(dstp)[0] = lo2; reinterpret_cast<unsigned*>(dstp)[1] = hi2; }
      else {
        unsigned short* d16 = reinterpret_cast<unsigned short*>(dstp);
        d16[0] = (unsigned short)(lo2 & 0xffffu);
        if (x + 1 < wo) d16[1] = (unsigned short)(lo2 >> 16);
        if (x + 2 < wo) d16[2] = (unsigned short)(hi2 & 0xffffu);
        if (x + 3 < wo) d16[3] = (unsigned short)(hi2 >> 16);
      }
    }
  }
}

// out = the sum of the K splits' partial outputs, in split order (the coarse decoder levels: few pixels, thousands of K — the splits are what fills the chip)
template <typename TO>
__global__ __launch_bounds__(256) void k_conv_split_sum(const float* __restrict__ part, TO* __restrict__ out, size_t n4, int KS) {
  const size_t i = (size_t)blockIdx.x*256 + threadIdx.x;
  if (i >= n4) return;
  const f4* p = reinterpret_cast<const f4*>(part);
  f4 s = p[i];
  for (int k = 1; k < KS; ++k) s += p[(size_t)k*n4 + i];
  if constexpr (sizeof(TO) == 4) reinterpret_cast<f4*>(out)[i] = s;
  else { unsigned* o = reinterpret_cast<unsigned*>(out) + 2*i; o[0] = pack_bf16_rne(s.x, s.y); o[1] = pack_bf16_rne(s.z, s.w); }
}

// ---- weight gradient ----
// g_w[co][c][tap] = sum over samples and pixels of g_y[co][y][x] xp[c][y + ky][x + kx]: per tap a GEMM with M = output channels, N = input channels,
// K = pixels.  Both operands want 8 consecutive K per lane = 8 consecutive pixels of a row of one channel: the tensors' own (NCHW) order.  A K step is 16
// pixels of a row (lane group g: pixels 8 g .. + 7); A = g_y, B = the padded input shifted by the tap — the shift by kx is a funnel shift of the five
// dwords a lane reads (kx = 0: dwords 0-3, kx = 2: dwords 1-4, kx = 1: v_alignbit of neighbours).  A wave owns ONE pair (tile of 32 output channels, tile
// of 32 input channels) and all nine taps: 9 x 16 accumulator registers.
// The row loop runs over INPUT rows: input row r meets g_y rows r, r - 1, r - 2 as the taps' rows ky = 0, 1, 2 — so the block keeps a ring of four g_y rows
// (small: 32 channels) and only TWO slots of the input row (64 channels: this row, and the next one being filed), an input row's fragments are read once
// and serve three ky (15 LDS reads per 54 MFMAs), and 60 KB of LDS leave room for two blocks per CU.  A block = 32 output x 64 input channels, four waves =
// 2 input-channel tiles x the 2 K steps of a 32-column strip; per row it requests one new row of each operand before the row's MFMAs and splits + files
// them after, one barrier per row.  The block leaves its sums as one set of partials [tap][co][c]; k_conv_wgrad_finalize adds the blocks' sets in fp64 in
// a fixed order (deterministic, as everywhere in this library).
// (First form, round 6: the ring held four INPUT rows of 64-128 channels — 93-143 KB, one block per CU, a lane's five dwords read per ky: 193 us at cfg 2's
// 96 -> 32 layer, the bf16 pipe 37 % busy, 64 % of the LDS cycles bank conflicts of the fifth-dword read.)
struct WgradTile {
  static constexpr int COB = 32, CB = 64;
  static constexpr int XROW = 20, XCH = 2*XROW + 4;     // dwords: a row slot = 40 bf16 (34 used), a channel = 2 slots + 16 bytes (176 B = 16 x 11: odd, the 16 lanes of a ds_read_b128 group cover all banks)
  static constexpr int GROW = 16, GCH = 4*GROW + 4;     // dwords: a row slot = 32 bf16, a channel = 4 slots + 16 bytes (272 B = 16 x 17)
};

template <int P, typename TI>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad_mfma(const TI* __restrict__ xp, const TI* __restrict__ gy, float* __restrict__ partial,
                                                         int C, int CO, int h, int w, int rows_per_block) {
  using T = WgradTile;
  constexpr int COB = T::COB, CB = T::CB, XROW = T::XROW, XCH = T::XCH, GROW = T::GROW, GCH = T::GCH, NPROD = n_products(P);
  constexpr int kXs = P*CB*XCH, kGs = P*COB*GCH;
  constexpr int kRed = 2*144*64;                          // the second wave of a pair parks its accumulators
  __shared__ __attribute__((aligned(16))) unsigned lds[(kXs + kGs) > kRed ? (kXs + kGs) : kRed];
  unsigned* const xs = lds;
  unsigned* const gs = lds + kXs;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 31, g = lane >> 5;
  const int ct = wv & 1, ks = wv >> 1;                    // this wave's input-channel tile and K step (columns 16 ks + 8 g .. + 7)
  const int CGRP = (C + CB - 1)/CB, COGRP = CO/COB;
  const int cg = blockIdx.z % CGRP, cog = (blockIdx.z/CGRP) % COGRP, b = blockIdx.z/(CGRP*COGRP);
  const int x0 = blockIdx.x*32, ybeg = blockIdx.y*rows_per_block, nrows = min(rows_per_block, h - ybeg);
  const int W = w + 2, H = h + 2;
  typedef typename RawOf<TI>::type R;
  const R* xsrc = reinterpret_cast<const R*>(xp) + (size_t)b*C*H*W;
  const R* gsrc = reinterpret_cast<const R*>(gy) + ((size_t)b*CO + (size_t)cog*COB)*h*w;

  constexpr int XITEMS = CB*17, XTRIPS = (XITEMS + 255)/256;     // an item = two adjacent columns of one channel's row
  constexpr int GITEMS = COB*16, GTRIPS = GITEMS/256;
  static_assert(GITEMS % 256 == 0, "g_y items per thread");
  R xv[XTRIPS][2], gv[GTRIPS][2];
  auto load_x = [&](int yy) {                                     // padded row yy (clamped: rows past the strip are requested but never used)
    yy = min(yy, H - 1);
#pragma unroll
    for (int t = 0; t < XTRIPS; ++t) {
      const int item = t*256 + (int)threadIdx.x;
      const int ch = min(item/17, CB - 1), pr = item % 17;
      const int c = min(cg*CB + ch, C - 1);
      const R* rowp = xsrc + ((size_t)c*H + yy)*W;
      xv[t][0] = rowp[min(x0 + 2*pr, W - 1)];
      xv[t][1] = rowp[min(x0 + 2*pr + 1, W - 1)];
    }
  };
  auto file_x = [&](int slot) {
#pragma unroll
    for (int t = 0; t < XTRIPS; ++t) {
      const int item = t*256 + (int)threadIdx.x;
      if (item < XITEMS) {
        const int ch = item/17, pr = item % 17;
        unsigned pk[P];
        split_pair<P>(xv[t][0], xv[t][1], pk);
#pragma unroll
        for (int p = 0; p < P; ++p) xs[(p*CB + ch)*XCH + slot*XROW + pr] = pk[p];
      }
    }
  };
  auto load_g = [&](int y) {                                      // beyond the image or the block's rows: zeros, those pixels add nothing
#pragma unroll
    for (int t = 0; t < GTRIPS; ++t) {
      const int item = t*256 + (int)threadIdx.x;
      const int co = item >> 4, pr = item & 15;
      const int xa = x0 + 2*pr;
      const bool yok = y < ybeg + nrows;
      const R* rowp = gsrc + ((size_t)co*h + (yok ? y : 0))*w;
      gv[t][0] = (yok && xa < w) ? rowp[xa] : R(0);
      gv[t][1] = (yok && xa + 1 < w) ? rowp[xa + 1] : R(0);
    }
  };
  auto file_g = [&](int slot) {
#pragma unroll
    for (int t = 0; t < GTRIPS; ++t) {
      const int item = t*256 + (int)threadIdx.x;
      const int co = item >> 4, pr = item & 15;
      unsigned pk[P];
      split_pair<P>(gv[t][0], gv[t][1], pk);
#pragma unroll
      for (int p = 0; p < P; ++p) gs[(p*COB + co)*GCH + slot*GROW + pr] = pk[p];
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // step i = 0 .. nrows + 1 works on padded input row ybeg + i (slot i & 1) against g_y rows ybeg + i - ky (ring slot (i - ky) & 3), ky = 0, 1, 2, where they
  // are rows of this block; g_y rows at or past ybeg + nrows are filed as zeros (load_g), so only the steps before the block's first rows need a guard
  load_x(ybeg); file_x(0);
  load_g(ybeg); file_g(0);
  __syncthreads();
  for (int i = 0; i < nrows + 2; ++i) {
    load_x(ybeg + i + 1);
    load_g(ybeg + i + 1);
    bf16x8 Bx[3][P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const uint4* q = reinterpret_cast<const uint4*>(&xs[(p*CB + ct*32 + j)*XCH + (i & 1)*XROW + ks*8 + g*4]);
      const uint4 d = q[0];
      const unsigned d4 = q[1].x;
      Bx[0][p] = as_frag(d);
      Bx[1][p] = as_frag(uint4{__builtin_amdgcn_alignbit(d.y, d.x, 16), __builtin_amdgcn_alignbit(d.z, d.y, 16), __builtin_amdgcn_alignbit(d.w, d.z, 16), __builtin_amdgcn_alignbit(d4, d.w, 16)});
      Bx[2][p] = as_frag(uint4{d.y, d.z, d.w, d4});
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      if (i - ky < 0) continue;                                   // (wave-uniform: the block's first two steps)
      bf16x8 A[P];
#pragma unroll
      for (int p = 0; p < P; ++p) A[p] = as_frag(*reinterpret_cast<const uint4*>(&gs[(p*COB + j)*GCH + ((i - ky) & 3)*GROW + ks*8 + g*4]));
#pragma unroll
      for (int t = 0; t < NPROD; ++t)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc[ky*3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[prod_a(P, t)], Bx[kx][prod_b(P, t)], acc[ky*3 + kx], 0, 0, 0);
    }
    file_x((i + 1) & 1);
    file_g((i + 1) & 3);
    __syncthreads();
  }

  // D[row = co][column = c] of tap t: the two K-step waves of a pair meet in LDS
  float* red = reinterpret_cast<float*>(lds);                     // (everybody is past the last barrier of the loop: the rings are free)
  if (ks == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(ct*144 + t*16 + r)*64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (ks == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] += red[(ct*144 + t*16 + r)*64 + lane];
    const size_t blk = ((size_t)b*gridDim.y + blockIdx.y)*gridDim.x + blockIdx.x;
    const int c = (cg*2 + ct)*32 + j;
    if (c < C) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cog*32 + (r & 3) + 8*(r >> 2) + 4*g;
          partial[((blk*9 + t)*CO + co)*C + c] = acc[t][r];
        }
    }
  }
}

// The same block for fp32 tensors with the rows brought in by LDS-DMA (see k_conv16_wgrad_dma below, where the form was built): a ring of four RAW rows —
// [64 input channels][36 dwords: 34 columns + 2][32 g_y channels][36: 32 columns + 4], 13.8 KB a slot — three rows in flight per block instead of one; a wave
// splits its own slice of a row at fragment read (10 columns of its input channel, 8 of its g_y channel, the latter also split by the other channel tile's wave)
// and keeps the g_y fragments of the two rows before in registers.
template <int P>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad_dma(const float* __restrict__ xp, const float* __restrict__ gy, float* __restrict__ partial,
                                                        int C, int CO, int h, int w, int rows_per_block) {
  constexpr int COB = 32, CB = 64, NW = 4, NPROD = n_products(P), CS = 36, D = 4;
  constexpr int XDW = CB*CS, GDW = COB*CS, NPX = XDW/64, NPG = GDW/64, NX = (NPX + NW - 1)/NW, NG = (NPG + NW - 1)/NW, NDMA = NX + NG, SLOT = XDW + GDW + 64;
  static_assert(XDW % 64 == 0 && GDW % 64 == 0 && (D - 2)*NDMA < 64 && D == 4, "DMA pieces and waits");
  constexpr int kRing = D*SLOT, kTab = NW*NDMA*64, kRed = 2*144*64;
  __shared__ __attribute__((aligned(16))) unsigned lds[(kRing + kTab) > kRed ? (kRing + kTab) : kRed];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 31, g = lane >> 5;
  const int ct = wv & 1, ks = wv >> 1;
  const int CGRP = (C + CB - 1)/CB, COGRP = CO/COB;
  const int cg = blockIdx.z % CGRP, cog = (blockIdx.z/CGRP) % COGRP, b = blockIdx.z/(CGRP*COGRP);
  const int x0 = blockIdx.x*32, ybeg = blockIdx.y*rows_per_block, nrows = min(rows_per_block, h - ybeg), nsteps = nrows + 2;
  const int W = w + 2, H = h + 2;
  const rsrc_t rs_x = make_rsrc(xp + (size_t)b*C*H*W, (size_t)C*H*W*4), rs_g = make_rsrc(gy + ((size_t)b*CO + (size_t)cog*COB)*h*w, (size_t)COB*h*w*4);
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)lds);
  auto dma = [&](const rsrc_t& rs, unsigned v, unsigned so, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v), "s"(rs), "s"(so), "s"(dst) : "memory");
  };
  // piece k of a region = its dwords k 64 .. + 63; this wave takes pieces n NW + wv (a piece that does not exist: zeros into the slot's spare 256 bytes).
  // A lane's offset of a piece does not depend on the row: computed once, kept in LDS behind the ring (the accumulators leave no registers for 14 of them, and
  // recomputing them costs every row more vector instructions than splitting its operands), read back by the lane that wrote it — no synchronisation.
  unsigned* const tab = lds + kRing + wv*(NDMA*64) + lane;
#pragma unroll
  for (int n = 0; n < NX; ++n) {
    const int k = n*NW + wv, d = k*64 + lane, ch = d/CS, col = d - ch*CS, c = cg*CB + ch;
    tab[n*64] = (k < NPX && col < 34 && x0 + col < W && c < C) ? (unsigned)((c*H)*W + x0 + col)*4u : 0x80000000u;
  }
#pragma unroll
  for (int n = 0; n < NG; ++n) {
    const int k = n*NW + wv, d = k*64 + lane, co = d/CS, col = d - co*CS;
    tab[(NX + n)*64] = (k < NPG && col < 32 && x0 + col < w) ? (unsigned)((co*h)*w + x0 + col)*4u : 0x80000000u;
  }
  auto issue = [&](int r) {
    const unsigned base = lds0 + (unsigned)((r % D)*SLOT*4);
    const unsigned sx = (unsigned)min(ybeg + r, H - 1)*(unsigned)W*4u, sg = (unsigned)min(ybeg + r, h - 1)*(unsigned)w*4u;
    unsigned v[NDMA];
#pragma unroll
    for (int n = 0; n < NDMA; ++n) v[n] = tab[n*64];
#pragma unroll
    for (int n = 0; n < NX; ++n) { const int k = n*NW + wv; dma(rs_x, v[n], sx, base + (unsigned)(k < NPX ? k*256 : (XDW + GDW)*4)); }
#pragma unroll
    for (int n = 0; n < NG; ++n) { const int k = n*NW + wv; dma(rs_g, v[NX + n], sg, base + (unsigned)(k < NPG ? XDW*4 + k*256 : (XDW + GDW)*4)); }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  bf16x8 A[3][P];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int p = 0; p < P; ++p) A[r][p] = as_frag(uint4{0u, 0u, 0u, 0u});

  auto step = [&](int i, bf16x8 (&a0)[P], const bf16x8 (&a1)[P], const bf16x8 (&a2)[P]) {
    const int after = min(D - 2, nsteps - 1 - i);
    if (after >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2*NDMA) : "memory");
    else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (i + D - 1 < nsteps) issue(i + D - 1);
    const float* slot = reinterpret_cast<const float*>(lds) + (i % D)*SLOT;
    const float* xr = slot + (ct*32 + j)*CS + ks*16 + g*8;
    const float* gr = slot + XDW + j*CS + ks*16 + g*8;
    const float4 x0v = *reinterpret_cast<const float4*>(xr), x1v = *reinterpret_cast<const float4*>(xr + 4);
    const float2 x2v = *reinterpret_cast<const float2*>(xr + 8);
    const float4 g0v = *reinterpret_cast<const float4*>(gr), g1v = *reinterpret_cast<const float4*>(gr + 4);
    unsigned px[5][P], pg[4][P];
    split_pair<P>(x0v.x, x0v.y, px[0]); split_pair<P>(x0v.z, x0v.w, px[1]); split_pair<P>(x1v.x, x1v.y, px[2]); split_pair<P>(x1v.z, x1v.w, px[3]); split_pair<P>(x2v.x, x2v.y, px[4]);
    split_pair<P>(g0v.x, g0v.y, pg[0]); split_pair<P>(g0v.z, g0v.w, pg[1]); split_pair<P>(g1v.x, g1v.y, pg[2]); split_pair<P>(g1v.z, g1v.w, pg[3]);
    bf16x8 Bx[3][P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      Bx[0][p] = as_frag(uint4{px[0][p], px[1][p], px[2][p], px[3][p]});
      Bx[1][p] = as_frag(uint4{__builtin_amdgcn_alignbit(px[1][p], px[0][p], 16), __builtin_amdgcn_alignbit(px[2][p], px[1][p], 16),
                               __builtin_amdgcn_alignbit(px[3][p], px[2][p], 16), __builtin_amdgcn_alignbit(px[4][p], px[3][p], 16)});
      Bx[2][p] = as_frag(uint4{px[1][p], px[2][p], px[3][p], px[4][p]});
      a0[p] = as_frag(i < nrows ? uint4{pg[0][p], pg[1][p], pg[2][p], pg[3][p]} : uint4{0u, 0u, 0u, 0u});
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const bf16x8 (&a)[P] = ky == 0 ? a0 : ky == 1 ? a1 : a2;
#pragma unroll
      for (int t = 0; t < NPROD; ++t)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc[ky*3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[prod_a(P, t)], Bx[kx][prod_b(P, t)], acc[ky*3 + kx], 0, 0, 0);
    }
  };
#pragma unroll
  for (int r = 0; r < D - 1; ++r) issue(r);
  for (int i = 0; i < nsteps; i += 3) {
    step(i, A[0], A[2], A[1]);
    if (i + 1 < nsteps) step(i + 1, A[1], A[0], A[2]);
    if (i + 2 < nsteps) step(i + 2, A[2], A[1], A[0]);
  }
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds);
  if (ks == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(ct*144 + t*16 + r)*64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (ks == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] += red[(ct*144 + t*16 + r)*64 + lane];
    const size_t blk = ((size_t)b*gridDim.y + blockIdx.y)*gridDim.x + blockIdx.x;
    const int c = (cg*2 + ct)*32 + j;
    if (c < C) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cog*32 + (r & 3) + 8*(r >> 2) + 4*g;
          partial[((blk*9 + t)*CO + co)*C + c] = acc[t][r];
        }
    }
  }
}

// Sixteen output channels (the thin last stage): the same structure on `v_mfma_f32_16x16x32_bf16` — a K step is 32 pixels of a row (lane group q: pixels
// 8 q .. + 7), A = g_y (16 channels), B = the padded input shifted by the tap, one 16 x 16 accumulator tile per tap (36 registers).  A block walks down a strip of
// 64 columns: 2 K steps per row x NC tiles of 16 input channels = 2 NC waves; ring of four g_y rows + two slots of the input row in LDS (40-54 KB: three or four
// blocks per CU, ~100 registers), one barrier per row.  HBM-bound (128-192 B per pixel for 2304-4608 multiply-adds at 6/16 of the f32 MFMA's time).
// (Earlier forms, round 6, 16 -> 16 at 192x640 / 32 -> 16 at 96x320: tiles of 32 x 4 pixels staged through LDS with two barriers per tile 198 / 277 us; fragments
// straight from memory with a rolling register window 138 / 77 — a quarter wave of a fragment load touches 16 channel rows; the f32-MFMA kernel 112 / 76.)
template <int NC, int P, typename TI>
__global__ __launch_bounds__(128*NC) void k_conv16_wgrad_mfma(const TI* __restrict__ xp, const TI* __restrict__ gy, float* __restrict__ partial, int h, int w, int rows_per_block) {
  constexpr int C = 16*NC, NT = 128*NC, NPROD = n_products(P);
  constexpr int XROW = 36, XCH = 2*XROW + 4;            // dwords: a row slot = 72 bf16 (66 used), a channel = 2 slots + 16 bytes (304 B = 16 x 19)
  constexpr int GROW = 32, GCH = 4*GROW + 4;            // dwords: a row slot = 64 bf16, a channel = 4 slots + 16 bytes (528 B = 16 x 33)
  constexpr int kXs = P*C*XCH, kGs = P*16*GCH, kRed = NC*36*64;
  __shared__ __attribute__((aligned(16))) unsigned lds[(kXs + kGs) > kRed ? (kXs + kGs) : kRed];
  unsigned* const xs = lds;
  unsigned* const gs = lds + kXs;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, q = lane >> 4;
  const int nc = wv >> 1, ks = wv & 1;                   // this wave's input-channel tile and K step (columns 32 ks + 8 q .. + 7)
  const int x0 = blockIdx.x*64, ybeg = blockIdx.y*rows_per_block, nrows = min(rows_per_block, h - ybeg), b = blockIdx.z;
  const int W = w + 2, H = h + 2;
  typedef typename RawOf<TI>::type R;
  const R* xsrc = reinterpret_cast<const R*>(xp) + (size_t)b*C*H*W;
  const R* gsrc = reinterpret_cast<const R*>(gy) + (size_t)b*16*h*w;

  constexpr int XITEMS = C*33, XTRIPS = (XITEMS + NT - 1)/NT;   // an item = two adjacent columns of one channel's row (66 columns)
  constexpr int GITEMS = 16*32, GTRIPS = GITEMS/NT;
  static_assert(GITEMS % NT == 0, "g_y items per thread");
  R xv[XTRIPS][2], gv[GTRIPS][2];
  auto load_x = [&](int yy) {
    yy = min(yy, H - 1);
#pragma unroll
    for (int t = 0; t < XTRIPS; ++t) {
      const int item = min(t*NT + (int)threadIdx.x, XITEMS - 1);
      const int c = item/33, pr = item - c*33;
      const R* rowp = xsrc + ((size_t)c*H + yy)*W;
      xv[t][0] = rowp[min(x0 + 2*pr, W - 1)];
      xv[t][1] = rowp[min(x0 + 2*pr + 1, W - 1)];
    }
  };
  auto file_x = [&](int slot) {
#pragma unroll
    for (int t = 0; t < XTRIPS; ++t) {
      const int item = t*NT + (int)threadIdx.x;
      if (item < XITEMS) {
        const int c = item/33, pr = item - c*33;
        unsigned pk[P];
        split_pair<P>(xv[t][0], xv[t][1], pk);
#pragma unroll
        for (int p = 0; p < P; ++p) xs[(p*C + c)*XCH + slot*XROW + pr] = pk[p];
      }
    }
  };
  auto load_g = [&](int y) {                                      // beyond the image or the block's rows: zeros, those pixels add nothing
#pragma unroll
    for (int t = 0; t < GTRIPS; ++t) {
      const int item = t*NT + (int)threadIdx.x;
      const int co = item >> 5, pr = item & 31;
      const int xa = x0 + 2*pr;
      const bool yok = y < ybeg + nrows;
      const R* rowp = gsrc + ((size_t)co*h + (yok ? y : 0))*w;
      gv[t][0] = (yok && xa < w) ? rowp[xa] : R(0);
      gv[t][1] = (yok && xa + 1 < w) ? rowp[xa + 1] : R(0);
    }
  };
  auto file_g = [&](int slot) {
#pragma unroll
    for (int t = 0; t < GTRIPS; ++t) {
      const int item = t*NT + (int)threadIdx.x;
      const int co = item >> 5, pr = item & 31;
      unsigned pk[P];
      split_pair<P>(gv[t][0], gv[t][1], pk);
#pragma unroll
      for (int p = 0; p < P; ++p) gs[(p*16 + co)*GCH + slot*GROW + pr] = pk[p];
    }
  };

  f32x4v acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};

  // step i = 0 .. nrows + 1 works on padded input row ybeg + i (slot i & 1) against g_y rows ybeg + i - ky (ring slot (i - ky) & 3), as in k_conv_wgrad_mfma
  load_x(ybeg); file_x(0);
  load_g(ybeg); file_g(0);
  __syncthreads();
  for (int i = 0; i < nrows + 2; ++i) {
    load_x(ybeg + i + 1);
    load_g(ybeg + i + 1);
    bf16x8 Bx[3][P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const uint4* qp = reinterpret_cast<const uint4*>(&xs[(p*C + nc*16 + j)*XCH + (i & 1)*XROW + ks*16 + q*4]);
      const uint4 d = qp[0];
      const unsigned d4 = qp[1].x;
      Bx[0][p] = as_frag(d);
      Bx[1][p] = as_frag(uint4{__builtin_amdgcn_alignbit(d.y, d.x, 16), __builtin_amdgcn_alignbit(d.z, d.y, 16), __builtin_amdgcn_alignbit(d.w, d.z, 16), __builtin_amdgcn_alignbit(d4, d.w, 16)});
      Bx[2][p] = as_frag(uint4{d.y, d.z, d.w, d4});
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      if (i - ky < 0) continue;                                   // (wave-uniform: the block's first two steps)
      bf16x8 A[P];
#pragma unroll
      for (int p = 0; p < P; ++p) A[p] = as_frag(*reinterpret_cast<const uint4*>(&gs[(p*16 + j)*GCH + ((i - ky) & 3)*GROW + ks*16 + q*4]));
#pragma unroll
      for (int t = 0; t < NPROD; ++t)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc[ky*3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[prod_a(P, t)], Bx[kx][prod_b(P, t)], acc[ky*3 + kx], 0, 0, 0);
    }
    file_x((i + 1) & 1);
    file_g((i + 1) & 3);
    __syncthreads();
  }
  // D[row = co 4 q + v][column = c j] of tap t: the two K-step waves of a channel tile meet in LDS
  float* red = reinterpret_cast<float*>(lds);                     // (everybody is past the last barrier of the loop: the rings are free)
  if (ks == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v) red[(nc*36 + t*4 + v)*64 + lane] = acc[t][v];
  }
  __syncthreads();
  if (ks == 0) {
    const size_t blk = ((size_t)b*gridDim.y + blockIdx.y)*gridDim.x + blockIdx.x;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int co = 4*q + v, c = nc*16 + j;
        partial[((blk*9 + t)*16 + co)*C + c] = acc[t][v] + red[(nc*36 + t*4 + v)*64 + lane];
      }
  }
}

// The thin weight gradient for fp32 tensors, fourth form: the rows reach LDS by LDS-DMA (`buffer_load_dword ... lds`: no staging registers, so a ring of D rows
// costs LDS only and D - 1 rows are in flight per block), RAW; a wave reads its own slice of a row — 10 columns of its input channel, 8 of its g_y channel — and
// splits it in registers (every element is split by exactly one wave of its channel tile; the g_y slice again by each of the NC tiles), keeps the g_y fragments
// of the two rows before in registers for ky = 1, 2.  The third form issued a row's loads at the top of a step and filed them at its bottom: one row (8 KB) in
// flight per block, 3.3 us per row step at 16 -> 16 (the MFMAs of a step are 0.4 us).  One barrier per row, no vector-memory wait but the in-order counter.
// A slot = [C input channels][68 dwords: 66 columns + 2] [16 g_y channels][68: 64 columns + 4] (+ one dummy piece where the pieces do not divide among the waves);
// channel stride 272 B = 16 x 17.  Pieces outside the image (columns past the row, g_y rows past the block) carry an out-of-range offset / an empty resource:
// the DMA writes zeros.
template <int NC, int P>
__global__ __launch_bounds__(128*NC) void k_conv16_wgrad_dma(const float* __restrict__ xp, const float* __restrict__ gy, float* __restrict__ partial, int h, int w, int rows_per_block) {
  constexpr int C = 16*NC, NW = 2*NC, NPROD = n_products(P), CS = 68, D = 4;
  constexpr int XDW = C*CS, GDW = 16*CS, NPX = XDW/64, NPG = GDW/64, NX = (NPX + NW - 1)/NW, NG = (NPG + NW - 1)/NW, NDMA = NX + NG, SLOT = XDW + GDW + 64;
  static_assert(XDW % 64 == 0 && GDW % 64 == 0, "whole DMA pieces per region");
  static_assert((D - 2)*NDMA < 64, "the waits below are immediates of six bits");
  constexpr int kRed = NC*36*64;
  __shared__ __attribute__((aligned(16))) unsigned lds[(D*SLOT) > kRed ? (D*SLOT) : kRed];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 15, q = lane >> 4;
  const int nc = wv >> 1, ks = wv & 1;
  const int x0 = blockIdx.x*64, ybeg = blockIdx.y*rows_per_block, nrows = min(rows_per_block, h - ybeg), b = blockIdx.z;
  const int W = w + 2, H = h + 2, nsteps = nrows + 2;
  const rsrc_t rs_x = make_rsrc(xp + (size_t)b*C*H*W, (size_t)C*H*W*4), rs_g = make_rsrc(gy + (size_t)b*16*h*w, (size_t)16*h*w*4);
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)lds);

  // this wave's pieces of a row: input piece n is piece k = n NW + wv of the slot's input region (dword k 64 + lane), likewise for g_y; a wave whose last
  // piece does not exist sends it (out-of-range offsets: zeros) to the slot's 256 spare bytes, so that every wave has NX + NG loads in flight per row
  unsigned vx[NX], vg[NG];
#pragma unroll
  for (int n = 0; n < NX; ++n) {
    const int k = n*NW + wv, d = k*64 + lane, c = d/CS, col = d - c*CS;
    vx[n] = (k < NPX && col < 66 && x0 + col < W) ? (unsigned)((c*H)*W + x0 + col)*4u : 0x80000000u;
  }
#pragma unroll
  for (int n = 0; n < NG; ++n) {
    const int k = n*NW + wv, d = k*64 + lane, co = d/CS, col = d - co*CS;
    vg[n] = (k < NPG && col < 64 && x0 + col < w) ? (unsigned)((co*h)*w + x0 + col)*4u : 0x80000000u;
  }
  auto dma = [&](const rsrc_t& rs, unsigned v, unsigned so, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v), "s"(rs), "s"(so), "s"(dst) : "memory");
  };
  auto issue = [&](int r) {                                      // row r of the block -> slot r % D   (g_y rows past the block's: a valid row, unused)
    const unsigned base = lds0 + (unsigned)((r % D)*SLOT*4);
    const unsigned sx = (unsigned)min(ybeg + r, H - 1)*(unsigned)W*4u, sg = (unsigned)min(ybeg + r, h - 1)*(unsigned)w*4u;
#pragma unroll
    for (int n = 0; n < NX; ++n) { const int k = n*NW + wv; dma(rs_x, vx[n], sx, base + (unsigned)(k < NPX ? k*256 : (XDW + GDW)*4)); }
#pragma unroll
    for (int n = 0; n < NG; ++n) { const int k = n*NW + wv; dma(rs_g, vg[n], sg, base + (unsigned)(k < NPG ? XDW*4 + k*256 : (XDW + GDW)*4)); }
  };

  f32x4v acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
  bf16x8 A[3][P];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int p = 0; p < P; ++p) A[r][p] = as_frag(uint4{0u, 0u, 0u, 0u});

  // step i: padded input row ybeg + i (slot i % D) against g_y rows ybeg + i - ky: this row's fragments (a0) and the two rows' before (a1, a2)
  auto step = [&](int i, bf16x8 (&a0)[P], const bf16x8 (&a1)[P], const bf16x8 (&a2)[P]) {
    // this wave's pieces of row i have landed (the rows requested after it may be in flight: D - 2 of them, fewer at the block's end)
    const int after = min(D - 2, nsteps - 1 - i);
    if (after >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2*NDMA) : "memory");
    else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                             // ... and everybody's; nobody reads slot (i - 1) % D any more
    asm volatile("" ::: "memory");
    if (i + D - 1 < nsteps) issue(i + D - 1);
    const float* slot = reinterpret_cast<const float*>(lds) + (i % D)*SLOT;
    const float* xr = slot + (nc*16 + j)*CS + ks*32 + q*8;
    const float* gr = slot + XDW + j*CS + ks*32 + q*8;
    const float4 x0v = *reinterpret_cast<const float4*>(xr), x1v = *reinterpret_cast<const float4*>(xr + 4);
    const float2 x2v = *reinterpret_cast<const float2*>(xr + 8);
    const float4 g0v = *reinterpret_cast<const float4*>(gr), g1v = *reinterpret_cast<const float4*>(gr + 4);
    unsigned px[5][P], pg[4][P];
    split_pair<P>(x0v.x, x0v.y, px[0]); split_pair<P>(x0v.z, x0v.w, px[1]); split_pair<P>(x1v.x, x1v.y, px[2]); split_pair<P>(x1v.z, x1v.w, px[3]); split_pair<P>(x2v.x, x2v.y, px[4]);
    split_pair<P>(g0v.x, g0v.y, pg[0]); split_pair<P>(g0v.z, g0v.w, pg[1]); split_pair<P>(g1v.x, g1v.y, pg[2]); split_pair<P>(g1v.z, g1v.w, pg[3]);
    bf16x8 Bx[3][P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      Bx[0][p] = as_frag(uint4{px[0][p], px[1][p], px[2][p], px[3][p]});
      Bx[1][p] = as_frag(uint4{__builtin_amdgcn_alignbit(px[1][p], px[0][p], 16), __builtin_amdgcn_alignbit(px[2][p], px[1][p], 16),
                               __builtin_amdgcn_alignbit(px[3][p], px[2][p], 16), __builtin_amdgcn_alignbit(px[4][p], px[3][p], 16)});
      Bx[2][p] = as_frag(uint4{px[1][p], px[2][p], px[3][p], px[4][p]});
      a0[p] = as_frag(i < nrows ? uint4{pg[0][p], pg[1][p], pg[2][p], pg[3][p]} : uint4{0u, 0u, 0u, 0u});   // (past the block's rows: nothing to add)
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const bf16x8 (&a)[P] = ky == 0 ? a0 : ky == 1 ? a1 : a2;
#pragma unroll
      for (int t = 0; t < NPROD; ++t)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc[ky*3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[prod_a(P, t)], Bx[kx][prod_b(P, t)], acc[ky*3 + kx], 0, 0, 0);
    }
  };

  static_assert(D == 4, "the waits above");
#pragma unroll
  for (int r = 0; r < D - 1; ++r) issue(r);                       // (nsteps >= 3)
  for (int i = 0; i < nsteps; i += 3) {
    step(i, A[0], A[2], A[1]);
    if (i + 1 < nsteps) step(i + 1, A[1], A[0], A[2]);
    if (i + 2 < nsteps) step(i + 2, A[2], A[1], A[0]);
  }
  __syncthreads();                                                // the ring is free
  float* red = reinterpret_cast<float*>(lds);
  if (ks == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v) red[(nc*36 + t*4 + v)*64 + lane] = acc[t][v];
  }
  __syncthreads();
  if (ks == 0) {
    const size_t blk = ((size_t)b*gridDim.y + blockIdx.y)*gridDim.x + blockIdx.x;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int co = 4*q + v, c = nc*16 + j;
        partial[((blk*9 + t)*16 + co)*C + c] = acc[t][v] + red[(nc*36 + t*4 + v)*64 + lane];
      }
  }
}

// partial[t][tap][co][c] -> g_w[co][c][tap], fp64, fixed order.  Many blocks' sums for few weights (the thin stage: T = 960 sets of 2304): two launches —
// (1) a block = 64 weights x one of G slices of the T sets, its four waves every fourth set of the slice, added in wave order -> slice[g][i] (fp64, behind the
// partials in the workspace); (2) the G slices in order.  (One launch of ceil(n / 64) blocks over all T sets — 36 blocks reading 8.8 MB — took 67 us beside a
// 47 us kernel.)  Few sets (T < 64: the coarse levels, up to 1.2 M weights): G = 1 and the first launch writes g_w itself.
template <bool DIRECT>
__global__ __launch_bounds__(256) void k_conv_wgrad_finalize(const float* __restrict__ partial, unsigned T, unsigned G, int CO, int C, double* __restrict__ slice, float* __restrict__ g_w) {
  __shared__ double part[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n = CO*C*9, i = blockIdx.x*64 + lane;
  const unsigned g = blockIdx.y, t0 = (unsigned)(((unsigned long long)T*g)/G), t1 = (unsigned)(((unsigned long long)T*(g + 1))/G);
  double s = 0.0;
  if (i < n) for (unsigned t = t0 + wv; t < t1; t += 4) s += (double)partial[(size_t)t*n + i];
  part[wv][lane] = s;
  __syncthreads();
  if (wv == 0 && i < n) {
    const double tot = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    if (DIRECT) { const int c = i % C, co = (i/C) % CO, tap = i/(C*CO); g_w[((size_t)co*C + c)*9 + tap] = (float)tot; }
    else slice[(size_t)g*n + i] = tot;
  }
}
__global__ __launch_bounds__(256) void k_conv_wgrad_finalize2(const double* __restrict__ slice, unsigned G, int CO, int C, float* __restrict__ g_w) {
  const int n = CO*C*9, i = blockIdx.x*256 + threadIdx.x;
  if (i >= n) return;
  double tot = 0.0;
  unsigned g = 0;
  for (; g + 8 <= G; g += 8) {                                     // eight loads in flight, added in order
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = slice[(size_t)(g + k)*n + i];
#pragma unroll
    for (int k = 0; k < 8; ++k) tot += v[k];
  }
  for (; g < G; ++g) tot += slice[(size_t)g*n + i];
  const int c = i % C, co = (i/C) % CO, tap = i/(C*CO);
  g_w[((size_t)co*C + c)*9 + tap] = (float)tot;
}
static unsigned wgrad_slices(unsigned T) { return T < 64 ? 1u : std::min(32u, T/16); }

// ---- launch shapes ----
static void wgrad_shape(int B, int C, int CO, int h, int w, dim3& grid, int& rows) {
  const int strips = ceil_div(w, 32), base = strips*B*ceil_div(C, 64)*(CO/32);
  const int groups = std::max(1, std::min(512/std::max(base, 1), ceil_div(h, 12)));  // two blocks per CU: about one generation of equal blocks where the layer allows; at least
                                                                                      // twelve rows per block (a block runs two steps more than it has rows, then reduces and writes 9 x 32 x 64 sums)
  rows = ceil_div(h, groups);
  grid = dim3(strips, ceil_div(h, rows), B*ceil_div(C, 64)*(CO/32));
}
static void wgrad16_shape(int B, int C, int h, int w, dim3& grid, int& rows) {
  const int strips = ceil_div(w, 64);
  const long long units = (long long)strips*B, slots = C == 16 ? 1024 : 768;   // strips of 64 columns; ONE generation of blocks (four / three per CU), at least twelve rows per block
  const int groups = (int)std::max(1ll, std::min<long long>(ceil_div(h, 12), slots/units));
  rows = ceil_div(h, groups);
  grid = dim3(strips, ceil_div(h, rows), B);
}
// floats of workspace: the blocks' partial sums, then the finalize's fp64 slices
size_t conv_mfma_wgrad_partials(int B, int C, int CO, int h, int w) {
  dim3 grid; int rows;
  if (CO == 16) wgrad16_shape(B, C, h, w, grid, rows); else wgrad_shape(B, C, CO, h, w, grid, rows);
  const size_t T = (size_t)grid.x*grid.y*B, n = (size_t)9*CO*C;
  return T*n + 2*(size_t)wgrad_slices((unsigned)T)*n;           // (T n is even: n = 9 CO C with CO even)
}
static size_t thin_packed_elems(int C, int pieces) { return (size_t)(C >> 4)*5*pieces*512; }
size_t conv_mfma_packed_elems(int C, int CO, int pieces) { return std::max((size_t)CO*C*9*pieces, CO == 16 ? thin_packed_elems(C, pieces) : (size_t)0); }

// ---- launches.  `pieces`: 3 (or the experiment's 2): fp32 tensors, split operands; 1: bfloat16 tensors in and out (fp32 weights, packed as their bf16 rounding;
// fp32 accumulation, fp32 weight gradient) — the decoder under bf16 autocast ----
#define SMD_BY_PIECES(pieces, CALL) do { if ((pieces) == 3) { CALL(3, float); } else if ((pieces) == 2) { CALL(2, float); } else { CALL(1, bf16); } } while (0)

// 16 output channels: the thin operand image (forward: C = 16 or 32; data gradient: the thin image for C = 16, the wide one — 32 rows — for C = 32)
template <int P, typename T>
static void launch_conv16(const void* in, const void* wp, void* out, int B, int CK, bool bwd, int hi, int wi, int ho, int wo, hipStream_t st) {
  const unsigned gx = ceil_div(wo, 64), gy = ceil_div(ho, 4), gz = B;
  const dim3 grid(8*(unsigned)ceil_div((long long)gx*gy*gz, 8ll));
  const uint4* wq = (const uint4*)wp;
  const T* i_ = (const T*)in; T* o_ = (T*)out;
  if (bwd) hipLaunchKernelGGL((k_conv16_mfma<1, P, true, T, T>), grid, dim3(256), 0, st, i_, wq, o_, hi, wi, ho, wo, gx, gy, gz);
  else if (CK == 16) hipLaunchKernelGGL((k_conv16_mfma<1, P, false, T, T>), grid, dim3(256), 0, st, i_, wq, o_, hi, wi, ho, wo, gx, gy, gz);
  else hipLaunchKernelGGL((k_conv16_mfma<2, P, false, T, T>), grid, dim3(256), 0, st, i_, wq, o_, hi, wi, ho, wo, gx, gy, gz);
}

hipError_t launch_conv_mfma_pack(const float* w, void* wp_fwd, void* wp_bwd, int C, int CO, int pieces, hipStream_t st) {
  if (CO == 16) {
    void* thin_bwd = (C == 16) ? wp_bwd : nullptr;
    if (wp_fwd || thin_bwd) {
      const dim3 g16(ceil_div((C >> 4)*5*512, 256));
#define SMD_CALL(P, T) hipLaunchKernelGGL((k_conv_pack_w16<P>), g16, dim3(256), 0, st, w, (unsigned short*)wp_fwd, (unsigned short*)thin_bwd, C)
      SMD_BY_PIECES(pieces, SMD_CALL);
#undef SMD_CALL
    }
    if (C == 16 || !wp_bwd) return hipGetLastError();
    wp_fwd = nullptr;                                             // C = 32: the data gradient is a 32-row layer of the wide kernel
  }
  const dim3 grid(ceil_div(CO*C*9, 256));
#define SMD_CALL(P, T) hipLaunchKernelGGL((k_conv_pack_w<P>), grid, dim3(256), 0, st, w, (unsigned short*)wp_fwd, (unsigned short*)wp_bwd, CO, C)
  SMD_BY_PIECES(pieces, SMD_CALL);
#undef SMD_CALL
  return hipGetLastError();
}

// Launch shape of the forward / data-gradient form: tile columns, channel tiles per block, K splits (each split a whole number of 16-channel chunks).
static int g_conv_two_tiles = 0;
void set_conv_two_tiles(int v) { g_conv_two_tiles = v; }
static bool conv_two_tiles() { return g_conv_two_tiles != 0; }
struct ConvShape { int TC, TRB, NM, KS, kcs; unsigned gx, gy, gz; dim3 grid; size_t out_elems; };
static ConvShape conv_shape(int B, int CK, int M, int ho, int wo) {
  ConvShape s;
  s.TC = wo >= 48 ? 64 : 32; s.TRB = wo >= 48 ? 4 : 8;
  const long long tiles = (long long)ceil_div(wo, s.TC)*ceil_div(ho, s.TRB)*B;
  s.NM = (M % 64 == 0 && tiles*(M/64) >= 256 && conv_two_tiles()) ? 2 : 1;   // two channel tiles over one patch where that still leaves a block per CU
  const long long base = tiles*(M/(32*s.NM));
  const int KC = CK >> 4;
  int ks = 1;
  if (base < 384) ks = (int)std::min<long long>(std::max(KC/2, 1), (512 + base - 1)/base);   // under 1.5 blocks per CU: split K, at least two chunks per split
  s.kcs = ceil_div(KC, ks); s.KS = ceil_div(KC, s.kcs);
  s.gx = ceil_div(wo, s.TC); s.gy = ceil_div(ho, s.TRB); s.gz = B;
  s.grid = dim3(8*(unsigned)ceil_div((long long)s.gx*s.gy*s.gz*(M/(32*s.NM))*s.KS, 8ll));
  s.out_elems = (size_t)B*M*ho*wo;
  return s;
}
size_t conv_mfma_split_elems(int B, int CK, int M, int ho, int wo) {
  const ConvShape s = conv_shape(B, CK, M, ho, wo);
  return s.KS > 1 ? (size_t)s.KS*s.out_elems : 0;
}

template <int P, bool BWD, typename T>
static void launch_conv_form(const void* in, const void* wp, void* out, float* split_ws, int B, int CK, int M, int hi, int wi, int ho, int wo, hipStream_t st) {
  const ConvShape s = conv_shape(B, CK, M, ho, wo);
  const uint4* wq = (const uint4*)wp;
  const T* i_ = (const T*)in;
  T* dst = s.KS > 1 ? reinterpret_cast<T*>(split_ws) : (T*)out;    // (the kernel writes a split's partial output as fp32 whatever T)
  if (s.NM == 2) {
    if (s.TC == 64) hipLaunchKernelGGL((k_conv_mfma<64, P, BWD, T, T, 2>), s.grid, dim3(512), 0, st, i_, wq, dst, CK, M, hi, wi, ho, wo, s.KS, s.kcs, s.out_elems, s.gx, s.gy, s.gz);
    else hipLaunchKernelGGL((k_conv_mfma<32, P, BWD, T, T, 2>), s.grid, dim3(512), 0, st, i_, wq, dst, CK, M, hi, wi, ho, wo, s.KS, s.kcs, s.out_elems, s.gx, s.gy, s.gz);
  } else if (s.TC == 64) hipLaunchKernelGGL((k_conv_mfma<64, P, BWD, T, T, 1>), s.grid, dim3(256), 0, st, i_, wq, dst, CK, M, hi, wi, ho, wo, s.KS, s.kcs, s.out_elems, s.gx, s.gy, s.gz);
  else hipLaunchKernelGGL((k_conv_mfma<32, P, BWD, T, T, 1>), s.grid, dim3(256), 0, st, i_, wq, dst, CK, M, hi, wi, ho, wo, s.KS, s.kcs, s.out_elems, s.gx, s.gy, s.gz);
  if (s.KS > 1) {
    const size_t n4 = s.out_elems/4;                              // (B M ho wo is a multiple of 4: M is a multiple of 32)
    hipLaunchKernelGGL((k_conv_split_sum<T>), dim3((unsigned)((n4 + 255)/256)), dim3(256), 0, st, split_ws, (T*)out, n4, s.KS);
  }
}

// y (B, CO, h, w) = conv3x3(xp (B, C, h + 2, w + 2)): C % 16 == 0, CO % 32 == 0
size_t conv_mfma_fwd_split_elems(int B, int C, int CO, int h, int w) { return CO % 32 ? 0 : conv_mfma_split_elems(B, C, CO, h, w); }
size_t conv_mfma_bwd_split_elems(int B, int C, int CO, int h, int w) { return C % 32 ? 0 : conv_mfma_split_elems(B, CO, C, h + 2, w + 2); }
hipError_t launch_conv_mfma_fwd(const void* xp, const void* wp_fwd, void* y, float* split_ws, int B, int C, int CO, int h, int w, int pieces, hipStream_t st) {
  if (CO == 16) {
#define SMD_CALL(P, T) launch_conv16<P, T>(xp, wp_fwd, y, B, C, false, h + 2, w + 2, h, w, st)
    SMD_BY_PIECES(pieces, SMD_CALL);
#undef SMD_CALL
    return hipGetLastError();
  }
#define SMD_CALL(P, T) launch_conv_form<P, false, T>(xp, wp_fwd, y, split_ws, B, C, CO, h + 2, w + 2, h, w, st)
  SMD_BY_PIECES(pieces, SMD_CALL);
#undef SMD_CALL
  return hipGetLastError();
}
// g_xp (B, C, h + 2, w + 2) from g_y (B, CO, h, w): CO % 16 == 0, C % 32 == 0
hipError_t launch_conv_mfma_bwd_data(const void* gy, const void* wp_bwd, void* g_xp, float* split_ws, int B, int C, int CO, int h, int w, int pieces, hipStream_t st) {
  if (CO == 16 && C == 16) {
#define SMD_CALL(P, T) launch_conv16<P, T>(gy, wp_bwd, g_xp, B, 16, true, h, w, h + 2, w + 2, st)
    SMD_BY_PIECES(pieces, SMD_CALL);
#undef SMD_CALL
    return hipGetLastError();
  }
#define SMD_CALL(P, T) launch_conv_form<P, true, T>(gy, wp_bwd, g_xp, split_ws, B, CO, C, h, w, h + 2, w + 2, st)
  SMD_BY_PIECES(pieces, SMD_CALL);
#undef SMD_CALL
  return hipGetLastError();
}
// g_w (CO, C, 3, 3) fp32: CO % 32 == 0, any C >= 1 (channel tiles past C are computed on clamped reads and not stored); or CO == 16 with C == 16 | 32
template <int P, typename T>
static void launch_wgrad(const void* xp_, const void* gy_, float* partial, int B, int C, int CO, int h, int w, hipStream_t st) {
  const T* xp = (const T*)xp_; const T* gy = (const T*)gy_;
  dim3 grid; int rows;
  if (CO == 16) {
    wgrad16_shape(B, C, h, w, grid, rows);
    if constexpr (std::is_same<T, float>::value) {                  // fp32 tensors: the LDS-DMA form
      if (C == 16) hipLaunchKernelGGL((k_conv16_wgrad_dma<1, P>), grid, dim3(128), 0, st, xp, gy, partial, h, w, rows);
      else hipLaunchKernelGGL((k_conv16_wgrad_dma<2, P>), grid, dim3(256), 0, st, xp, gy, partial, h, w, rows);
    } else {
      if (C == 16) hipLaunchKernelGGL((k_conv16_wgrad_mfma<1, P, T>), grid, dim3(128), 0, st, xp, gy, partial, h, w, rows);
      else hipLaunchKernelGGL((k_conv16_wgrad_mfma<2, P, T>), grid, dim3(256), 0, st, xp, gy, partial, h, w, rows);
    }
    return;
  }
  wgrad_shape(B, C, CO, h, w, grid, rows);
  if constexpr (std::is_same<T, float>::value) hipLaunchKernelGGL((k_conv_wgrad_dma<P>), grid, dim3(256), 0, st, xp, gy, partial, C, CO, h, w, rows);
  else hipLaunchKernelGGL((k_conv_wgrad_mfma<P, T>), grid, dim3(256), 0, st, xp, gy, partial, C, CO, h, w, rows);
}
hipError_t launch_conv_mfma_bwd_wgt(const void* xp, const void* gy, float* g_w, float* partial, int B, int C, int CO, int h, int w, int pieces, hipStream_t st) {
#define SMD_CALL(P, T) launch_wgrad<P, T>(xp, gy, partial, B, C, CO, h, w, st)
  SMD_BY_PIECES(pieces, SMD_CALL);
#undef SMD_CALL
  dim3 grid; int rows;
  if (CO == 16) wgrad16_shape(B, C, h, w, grid, rows); else wgrad_shape(B, C, CO, h, w, grid, rows);
  const unsigned T = grid.x*grid.y*(unsigned)B, G = wgrad_slices(T);
  const int n = CO*C*9;
  double* slice = reinterpret_cast<double*>(partial + (size_t)T*n);
  if (G == 1) hipLaunchKernelGGL(k_conv_wgrad_finalize<true>, dim3(ceil_div(n, 64), 1), dim3(256), 0, st, partial, T, 1u, CO, C, slice, g_w);
  else {
    hipLaunchKernelGGL(k_conv_wgrad_finalize<false>, dim3(ceil_div(n, 64), G), dim3(256), 0, st, partial, T, G, CO, C, slice, g_w);
    hipLaunchKernelGGL(k_conv_wgrad_finalize2, dim3(ceil_div(n, 256)), dim3(256), 0, st, slice, G, CO, C, g_w);
  }
  return hipGetLastError();
}

}  // namespace smd

#ifdef SMD_CONV_TRACE
extern "C" int smd_debug_conv_trace(unsigned long long* host_out, int blocks) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(smd::g_conv_trace), (size_t)blocks*40*sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif
