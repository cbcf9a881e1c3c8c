// smd_recon_fwd.hip — round-2 forward of the fused view-synthesis photometric loss for gfx950.
//
// The round-1 kernel was VALU-issue bound (about 425 vector instructions per 64-pixel row at ~3.5 cycles each, see
// profiles/r02_valu_rate2.txt for the issue rates this rewrite is priced against).  Same wave-strip streaming structure
// (one wave = 64 consecutive columns, 62 interior + 1 halo lane per side, walking down `rh` rows; horizontal 3x3 taps through
// DPP wave shifts, reflection by data in the halo lanes), but the instruction count per row is cut by:
//   * sliding vertical sums: per (support, channel) the state is {x(j-1), P = r(j-2) + r(j-1) for r in x, x^2, x*y}; a new
//     row costs V = P + r(j), P' = r(j-1) + r(j).  No 3-row ring, and with the row loop unrolled by two (the two state
//     copies swap roles) not a single register move.  Image-border reflection in y is a wave-uniform multiplier on the
//     new row (top) or one subtraction that recovers row h-2 from P (bottom).
//   * target-side window sums (sum y, 9*sum y^2 - (sum y)^2 + C2) do not depend on scale or support: k_recon_prep computes them
//     once per sample, together with the repack of the supports into 12-byte RGB texels (padded by one texel right / below so
//     that the bilinear tap block needs no clamp) and the identity ("static") error of the automask — one launch instead of
//     round 1's pack + identity passes.  The main kernel reads them back (three row loads per step: target pixel, {S_y, c_0},
//     {c_1, c_2, static error}; L2 hits for three of four scales).
//   * geometry: grid normalisation folded into the homography, per-lane column part hoisted out of the row loop, med3
//     clamps, one float->int conversion per tap block.
//   * addressing: every array is a buffer resource; the per-lane column offset is a loop-invariant VGPR and the row / plane
//     offset a scalar (soffset), so the coalesced loads and stores cost no VALU instruction.
// Supports are processed in pairs inside one launch (n <= 4 in a single pass, no err/sel read-modify-write); the gathers
// of the next pair / next row are in flight under the current pair's SSIM math.
// K0 is inside: the DISP instantiations take the decoder's low-resolution disparities, up-sample the row
// they are about to warp, convert it to depth and write `depth_up` once.
// What bounds the kernel: about 340-355 vector instructions and 17-18 vector memory instructions per row step for
// two supports; the CU's texture unit spends ~16.5 cycles per wave memory instruction whatever its width, which makes the two
// pipes equally loaded.  The launch shape (strip height, tapered tail) is chosen by smd_api.hip from wave traces.
// Reference semantics: src/tools/geometry.py:285-391, src/losses/photometric.py:23-88, src/losses/reconstruction.py:43-126.
#include "smd_common.h"
#include "smd_kernels.h"
#include "smd_smooth_dev.h"

#ifndef SMD_FWD_PRIO
#define SMD_FWD_PRIO 0   // experiment: s_setprio by remaining rows in the shared-ring forward (see step())
#endif
#ifndef SMD_ABLATE
#define SMD_ABLATE 0   // diagnosis builds only (scripts/dev/ablate.sh): bit 0 no tap gathers, bit 1 no row loads (incl. K0), bit 2 no stores, bit 3 no ta/tb loads, bit 4 no target-row load
#endif

namespace smd {

// ATen area_pixel_compute_source_index (align_corners=False) + index / lambda split, as the K0 kernel (smd_depth.hip)
__device__ __forceinline__ void src_index_f(int dst, float scale, int n_in, int& i0, int& i1, float& l1) {
  const float src = fmaxf(fmaf(scale, (float)dst + 0.5f, -0.5f), 0.f);
  i0 = min((int)src, n_in - 1);
  i1 = min(i0 + 1, n_in - 1);
  l1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
}

// ---------------------------------------------------------------------------------------------
// k_recon_prep: per (strip, sample).  Reads the planar target and support frames once and fills the caller-kept `packed`
// buffer (layout: smd_kernels.h) that the main kernel and the backward read through ONE buffer resource:
//   texels (n,b,h+1,w+1,3)   RGB texels of the supports, zero padding column w and row h
//   ypix   (b,h,w,3)         the target as RGB texels
//   ta, tb (b,h,w,4) each    {S_y[3], c_0}, {c_1, c_2, identity error, 0}: S_y = 3x3 reflect-padded window sum of the target
//                            channel, c = 9*S_yy - S_y^2 + 81*C2 (x81 variance term of photometric.py:44-47); identity error =
//                            min / mean over supports of the photometric error of the UN-warped support (reconstruction.py:70-71).
// ---------------------------------------------------------------------------------------------
template <int N, bool SSIM>
struct PrepCtx {
  const ReconPrepArgs& a;
  int h, w, u, r0, r1;
  unsigned lane4;          // byte offset of this lane's (reflected) column inside a row of floats
  bool interior;
  unsigned hw4, w4;        // bytes per plane / per row
  unsigned so_sup[N], so_tex[N], so_y, so_ta, so_tb;   // wave-uniform byte offsets: planar support k, its texel image, this sample's ypix / ta / tb
  rsrc_t rs_tgt, rs_sup, rs_pk;
  float Px[N][3], Pxx[N][3], Pxy[N][3], Py[3], Pyy[3];
  float ny[3], nx[N][3];   // loads in flight for the next row

  __device__ __forceinline__ void load_row(int j) {
    const unsigned ro = (unsigned)j*w4;
#pragma unroll
    for (int c = 0; c < 3; ++c) ny[c] = bld(rs_tgt, lane4, ro + (unsigned)c*hw4);
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) nx[k][c] = bld(rs_sup, lane4, so_sup[k] + ro + (unsigned)c*hw4);
  }
  __device__ __forceinline__ void pack_row(int j, const float (&X)[N][3], const float (&Y)[3]) {   // repack of row j (interior lanes)
    if (a.first_pass) { f3 t; t.x = Y[0]; t.y = Y[1]; t.z = Y[2]; bst3(rs_pk, lane4*3u, so_y + (unsigned)j*w4*3u, t); }
    const unsigned ro = (unsigned)j*((unsigned)w + 1u)*12u;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      f3 t; t.x = X[k][0]; t.y = X[k][1]; t.z = X[k][2];
      bst3(rs_pk, lane4*3u, so_tex[k] + ro, t);
      if (u == w - 1) { const f3 z = {0.f, 0.f, 0.f}; bst3(rs_pk, lane4*3u + 12u, so_tex[k] + ro, z); }
    }
  }

  // One row step: row j becomes the NEW row (Xn, Yn), the centre row is j-1 (Xo, Yo), P holds r(j-2) + r(j-1).
  // VIRT: j == h, the row below the image is row h-2 (ReflectionPad2d(1)) = P - r(h-1).
  template <bool EMIT, bool VIRT>
  __device__ __forceinline__ void step(int j, float (&Xo)[N][3], float (&Xn)[N][3], float (&Yo)[3], float (&Yn)[3]) {
    const int v = j - 1;
    if (!VIRT) {
#pragma unroll
      for (int c = 0; c < 3; ++c) Yn[c] = ny[c];
#pragma unroll
      for (int k = 0; k < N; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) Xn[k][c] = nx[k][c];
      if (j >= r0 && j < r1 && interior) pack_row(j, Xn, Yn);   // each row is interior to exactly one strip
      if (j + 1 <= min(r1, h - 1)) load_row(j + 1);
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) Yn[c] = Py[c] - Yo[c];
#pragma unroll
      for (int k = 0; k < N; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) Xn[k][c] = Px[k][c] - Xo[k][c];
    }
    const float m = (v == 0) ? 2.f : 1.f;  // row -1 is row 1: the new row counts twice for the first image row
    constexpr float c1 = 81.f*kC1, c2 = 81.f*kC2;
    float sy[3] = {0.f, 0.f, 0.f}, cy1[3], cy2[3] = {0.f, 0.f, 0.f};
    if (SSIM) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float yy = Yn[c]*Yn[c];
        const float Vy = fmaf(m, Yn[c], Py[c]), Vyy = fmaf(m, yy, Pyy[c]);
        Py[c] = Yo[c] + Yn[c]; Pyy[c] = fmaf(Yo[c], Yo[c], yy);
        if (EMIT) {
          float s1, s2;
          hsum2(Vy, Vyy, s1, s2);
          sy[c] = s1; cy2[c] = fmaf(9.f, s2, c2) - s1*s1; cy1[c] = fmaf(s1, s1, c1);
        }
      }
    }
    float best = 0.f, acc = 0.f;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      float es = 0.f, el = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float xn = Xn[k][c], xo = Xo[k][c];
        if (EMIT) el += fabsf(xo - Yo[c]);
        if (SSIM) {
          const float xx = xn*xn, xy = xn*Yn[c];
          const float Vx = fmaf(m, xn, Px[k][c]), Vxx = fmaf(m, xx, Pxx[k][c]), Vxy = fmaf(m, xy, Pxy[k][c]);
          Px[k][c] = xo + xn; Pxx[k][c] = fmaf(xo, xo, xx); Pxy[k][c] = fmaf(xo, Yo[c], xy);
          if (EMIT) {
            float sx, sxx, sxy;
            hsum3(Vx, Vxx, Vxy, sx, sxx, sxy);
            es += ssim_err81(sx, sxx, sxy, sy[c], cy1[c], cy2[c]);
          }
        }
      }
      if (EMIT) {
        const float e = SSIM ? fmaf(kWSsim/3.f, es, ((1.f - kWSsim)/3.f)*el) : el*(1.f/3.f);
        if (k == 0) { best = e; acc = e; } else { acc += e; best = fminf(best, e); }
      }
    }
    if (EMIT && interior) {
      const unsigned to = (unsigned)v*w4*4u;
      const bool use_min = a.flags & SMD_USE_MIN;
      float e = use_min ? best : acc;
      if (!a.first_pass) { const float prev = bld(rs_pk, lane4*4u + 8u, so_tb + to); e = use_min ? fminf(e, prev) : e + prev; }
      if (a.last_pass && !use_min) e = e/(float)a.n;
      if (a.first_pass) {
        f4 t0, t1; t0.x = sy[0]; t0.y = sy[1]; t0.z = sy[2]; t0.w = cy2[0]; t1.x = cy2[1]; t1.y = cy2[2]; t1.z = e; t1.w = 0.f;
        bst4(rs_pk, lane4*4u, so_ta + to, t0); bst4(rs_pk, lane4*4u, so_tb + to, t1);
      } else bst(rs_pk, lane4*4u + 8u, so_tb + to, e);
    }
  }

  __device__ __forceinline__ void init(float (&X)[N][3], float (&Y)[3]) {   // first row of the strip: P = r(jstart) alone
#pragma unroll
    for (int c = 0; c < 3; ++c) { Y[c] = ny[c]; Py[c] = Y[c]; Pyy[c] = Y[c]*Y[c]; }
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) { X[k][c] = nx[k][c]; Px[k][c] = X[k][c]; Pxx[k][c] = X[k][c]*X[k][c]; Pxy[k][c] = X[k][c]*Y[c]; }
  }
};

template <int N, bool SSIM>
__global__ __launch_bounds__(64*kWavesPerBlock) void k_recon_prep(const ReconPrepArgs a) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nstrips = a.nsx*a.nsy;
  const int nbx = ceil_div(nstrips, kWavesPerBlock);
  // K0 fused: the vertical half of the bilinear up-sampling is the same for every pixel of an image row, so it is tabulated
  // once per call ({byte offset of the two low-resolution rows, blend weight} per (scale, row); rows h, h+1 for the prefetch)
  // and the main kernel fetches its strip's entries once.  (Before the early return of waves without a strip: all 256 threads
  // of block 0 must take part, or entries beyond 64 x the number of live waves stay unwritten for tiny images.)
  if (a.arrive != nullptr && blockIdx.x == 0)   // arrival counters of the in-launch reductions of the main kernel / the backward
    for (int e = threadIdx.x; e < a.b + 1; e += 64*kWavesPerBlock) a.arrive[e] = 0u;
  if (a.rowtab != nullptr && blockIdx.x == 0) {
    for (int e = threadIdx.x; e < a.sc_S*(a.h + 4); e += 64*kWavesPerBlock) {
      const int sc_i = e/(a.h + 4), row = e - sc_i*(a.h + 4);
      int y0, y1; float ly;
      src_index_f(row, (float)a.sc_hs[sc_i]/(float)a.h, a.sc_hs[sc_i], y0, y1, ly);
      a.rowtab[e] = uint4{(unsigned)y0*(unsigned)a.sc_ws[sc_i]*4u, (unsigned)y1*(unsigned)a.sc_ws[sc_i]*4u, __builtin_bit_cast(unsigned, ly), 0u};
    }
  }
  const int bi = blockIdx.x/nbx, strip = (blockIdx.x - bi*nbx)*kWavesPerBlock + wid;
  if (strip >= nstrips) return;
  const int sxi = strip % a.nsx, syi = strip/a.nsx;
  PrepCtx<N, SSIM> cx{a};
  cx.h = a.h; cx.w = a.w;
  cx.r0 = syi*a.rh; cx.r1 = min(cx.r0 + a.rh, a.h);
  cx.u = sxi*kFwdCols - 1 + lane;
  const int uc = (cx.u < 0) ? min(-cx.u, a.w - 1) : ((cx.u >= a.w) ? max(2*(a.w - 1) - cx.u, 0) : cx.u);
  cx.lane4 = (unsigned)uc*4u;
  cx.interior = (lane >= 1) && (lane <= kFwdCols) && (cx.u < a.w);
  const size_t hw = (size_t)a.h*a.w;
  cx.hw4 = (unsigned)hw*4u; cx.w4 = (unsigned)a.w*4u;
  cx.rs_tgt = make_rsrc(a.tgt + (size_t)bi*3*hw, 3*hw*4);
  cx.rs_sup = make_rsrc(a.supp, (size_t)a.n*a.b*3*hw*4);
  cx.rs_pk = make_rsrc(a.packed, packed_image_floats(a.b, a.n, a.h, a.w)*4);
  const unsigned texel_bytes = (unsigned)(a.h + 1)*(unsigned)(a.w + 1)*12u;
  cx.so_y = (unsigned)(packed_texel_floats(a.b, a.n, a.h, a.w)*4) + (unsigned)bi*cx.hw4*3u;
  cx.so_ta = (unsigned)((packed_texel_floats(a.b, a.n, a.h, a.w) + packed_ypix_floats(a.b, a.h, a.w))*4) + (unsigned)bi*cx.hw4*4u;
  cx.so_tb = cx.so_ta + (unsigned)(packed_tpix_floats(a.b, a.h, a.w)*4);
#pragma unroll
  for (int k = 0; k < N; ++k) {
    cx.so_sup[k] = (unsigned)((a.i0 + k)*a.b + bi)*3u*cx.hw4;
    cx.so_tex[k] = (unsigned)((a.i0 + k)*a.b + bi)*texel_bytes;
  }

  float XA[N][3], XB[N][3], YA[3], YB[3];
  const int jstart = max(cx.r0 - 1, 0), jlast = min(cx.r1, a.h - 1);
  cx.load_row(jstart);
  int j = jstart + 1;
  if (jstart >= cx.r0 && cx.interior) cx.pack_row(jstart, cx.nx, cx.ny);   // jstart == r0 == 0
  if (cx.r0 > 0) {
    cx.init(XB, YB);
    if (jstart + 1 <= jlast) cx.load_row(jstart + 1);
    cx.template step<false, false>(j, XB, XA, YB, YA);   // centre r0-1 belongs to the strip above
    ++j;
  } else {
    cx.init(XA, YA);
    if (jstart + 1 <= jlast) cx.load_row(jstart + 1);
  }
  bool cur_is_a = true;
  for (; j <= jlast; j += 2) {
    cx.template step<true, false>(j, XA, XB, YA, YB);
    if (j + 1 > jlast) { cur_is_a = false; ++j; break; }
    cx.template step<true, false>(j + 1, XB, XA, YB, YA);
  }
  if (cx.r1 == a.h) {   // the strip owns the last image row: one more step whose new row is the reflected row h-2
    if (!cur_is_a) {
#pragma unroll
      for (int c = 0; c < 3; ++c) YA[c] = YB[c];
#pragma unroll
      for (int k = 0; k < N; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) XA[k][c] = XB[k][c];
    }
    cx.template step<true, true>(a.h, XA, XB, YA, YB);
    // zero padding row h of the texel array
    if (cx.interior) {
      const unsigned ro = (unsigned)a.h*((unsigned)a.w + 1u)*12u;
      const f3 z = {0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < N; ++k) {
        bst3(cx.rs_pk, cx.lane4*3u, cx.so_tex[k] + ro, z);
        if (cx.u == a.w - 1) bst3(cx.rs_pk, cx.lane4*3u + 12u, cx.so_tex[k] + ro, z);
      }
    }
  }
}

hipError_t launch_recon_prep(const ReconPrepArgs& a, hipStream_t st) {
  dim3 grid((unsigned)ceil_div(a.nsx*a.nsy, kWavesPerBlock)*(unsigned)a.b), block(64*kWavesPerBlock);
  const bool ssim = !(a.flags & SMD_LOSS_L1);
#define SMD_PREP(N_) do { if (ssim) hipLaunchKernelGGL((k_recon_prep<N_, true>), grid, block, 0, st, a); \
                          else hipLaunchKernelGGL((k_recon_prep<N_, false>), grid, block, 0, st, a); } while (0)
  switch (a.ni) { case 1: SMD_PREP(1); break; case 2: SMD_PREP(2); break; case 3: SMD_PREP(3); break; default: SMD_PREP(4); break; }
#undef SMD_PREP
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// k_recon_main: per (strip, sample, scale).
// Template parameters: N supports of this launch (1..4); SSIM (false = loss_name 'l1'); SINGLE (one launch covers all supports:
// no carried min / sum is read or written); AUX (the rarely used extras: caller-supplied tie-break noise, `supp_imgs_warp`
// output — kept out of the hot instantiation because their wave-uniform addresses would sit in SGPRs for the whole loop).
// ---------------------------------------------------------------------------------------------
template <int N>
struct MainPend {          // gathers in flight for one pair of supports
  f3 t[(N > 1) ? 2 : 1][4];
  float fx[(N > 1) ? 2 : 1], fy[(N > 1) ? 2 : 1];
};

// LA: look-ahead of the tap gathers in row steps.  1: the taps of row j+1 are requested while row j is scored.  2 (N <= 2 only): the
// taps of row j+2 — two sets of pending registers, each tied to the row buffer (XA / XB) its row will be blended into, so the
// unrolled loop still never moves a register; 28 more VGPRs (3 waves per SIMD instead of 4) for twice the distance between a
// gather and its first use.
// SH (round 3): the block's four waves are the FOUR SCALES of one strip.  What a row step reads on the target side — the target
// pixel of the new row, {S_y, c} and the identity error of the centre row: 40 bytes per pixel, the same for every scale — is
// fetched ONCE per block into an LDS ring by LDS-DMA (`buffer_load_dwordx3/x4 ... lds`: no registers, lane l lands at
// base + 16 l) instead of by every wave through the texture path: wave q brings in rows 4e + q of the next group of four rows
// ("epoch") while the block works on the current one, and the four waves meet at one s_barrier per epoch.  3 of 16 vector-memory
// instructions per row step become 0.75, and the L2 sees the target side once instead of four times.
//   ring: 8 slots (two epochs) x {ypix, ta, tb} x 64 lanes x 16 bytes; slot i & 7 holds {ypix[i], ta[i-1], tb[i-1]}: what step i reads.
constexpr int kRingSlotFloats = 3*64*4;
constexpr int kRingFloats = 8*kRingSlotFloats;

template <int N, bool SSIM, bool SINGLE, bool AUX, bool DISP, int LA, bool SH = false>
struct MainCtx {
  static_assert(LA == 1 || (LA == 2 && N <= 2), "two rows of look-ahead only with a single pair of supports");
  static_assert(!SH || (SSIM && SINGLE && !AUX && DISP && LA == 1), "the shared target ring exists for the hot instantiation only");
  static constexpr int NG = (N + 1)/2;
  const ReconMainArgs& a;
  int bi, s, h, w, r0, r1, jlast;
  // SH state: LDS byte address of the ring (scalar), this lane's float pointer into it, this wave's index in the block
  unsigned ring_lds;
  const float* ring_lane;
  int ring_q, ring_last;   // rows beyond ring_last are never read by this strip

  // One LDS-DMA piece: 64 lanes x `16 or 12` bytes from packed[voff + soff] to LDS[dst + 16*lane].  M0 carries the LDS address and is
  // compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md, LDS-DMA recipe).  hipcc does not count
  // this load in its s_waitcnt bookkeeping: epoch_sync() waits for it explicitly.
  __device__ __forceinline__ void dma16(unsigned voff, unsigned soff, unsigned dst) const {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs_pk), "s"(soff), "s"(dst) : "memory");
  }
  __device__ __forceinline__ void dma12(unsigned voff, unsigned soff, unsigned dst) const {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx3 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs_pk), "s"(soff), "s"(dst) : "memory");
  }
  // This wave's row of epoch `e` (rows 4e .. 4e+3): slot i <- {ypix[i], ta[i-1], tb[i-1]}, rows clamped into the image.
  __device__ __forceinline__ void dma_epoch(int e) const {
    const int i = __builtin_amdgcn_readfirstlane(4*e + ring_q);
    if (i > ring_last) return;
    const unsigned ry = (unsigned)min(i, h - 1), rt = (unsigned)min(max(i - 1, 0), h - 1);
    const unsigned dst = ring_lds + (unsigned)(i & 7)*(unsigned)(kRingSlotFloats*4);
    dma12(lane4*3u, so_y + ry*w4*3u, dst);
    dma16(lane4*4u, so_ta + rt*w4*4u, dst + 1024u);
    dma16(lane4*4u, so_tb + rt*w4*4u, dst + 2048u);
  }
  // Start of an epoch: the rows the block brought in during the previous one are complete and visible to all four waves
  // (placed where the wave has just consumed its tap gathers: every load older than them — the DMA pieces — has landed).
  __device__ __forceinline__ void epoch_sync() const {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  unsigned lane4;          // byte offset of this lane's (reflected) column inside a row of floats
  bool interior, use_min, automask, has_noise, want_w0, has_err;
  unsigned hw4, w4, rowbytes;
  float xmax, ymax, wpf;
  Cam2 cam[N];
  // per-lane column part of the homography rows.  Up to two supports: registers.  Three and four (168-VGPR budget, 3 waves per
  // SIMD): parked in this lane's LDS column and read back every row step — the allocator otherwise spills nine loop-invariant
  // values to scratch and reloads 5.5 of them per row step through the texture path (round 3; -DSMD_FWD_CAM_REGS restores it).
#ifdef SMD_FWD_CAM_REGS
  static constexpr bool kCamLds = false;
#else
  static constexpr bool kCamLds = N > 2;
#endif
  float hx0[kCamLds ? 1 : N], hy0[kCamLds ? 1 : N], hz0[kCamLds ? 1 : N];
  const float* camcol;     // kCamLds: [support][3][64 lanes] of this wave, + lane
  unsigned so_tex[N], so_y, so_ta, so_tb;   // wave-uniform byte offsets into `packed`: texel image of support k, this sample's ypix / ta / tb
  const float* nz_sb;               // noise of this (scale, sample) (AUX)
  rsrc_t rs_pk, rs_depth, rs_err, rs_sel;
  float lsum;
  // liveness of this strip for the backward (smd_kernels.h): lanes (= columns) in which some row's FINAL selection is support k; wave-uniform
  static constexpr int NT = SINGLE ? (N < kLiveSupports ? N : kLiveSupports) : kLiveSupports;
  unsigned long long livem[NT];
  float Px[N][3], Pxx[N][3], Pxy[N][3];
  MainPend<N> P[LA];
  f3 py;                   // target row in flight
  float Dcur, Dnext;       // depth of rows j and j+1
  float vfn;               // (float)(j + LA): the row whose taps the next issue requests
  // DISP (K0 fused): the depth of a row is computed here from the network's low-resolution sigmoid
  // disparity — bilinear up-sampling (`ops.interpolate_like`, src/tools/ops.py:311-314) + `to_scaled` / `to_inv`
  // (src/tools/geometry.py:62-90) — and written out once for the backward, instead of being read from a K0 launch's output.
  rsrc_t rs_disp, rs_dout;
  unsigned dx0, dx1;       // byte offsets of the two low-resolution columns this lane blends
  float dlx, a_scale, a_off;
  // This scale's {offset of row y0, offset of row y1, ly, -} per image row, written by the prep kernel.  The entries of the
  // strip's rows are fetched ONCE, one per lane (tab_*: lane k holds row tab_base + k), and a row's entry is then picked with
  // v_readlane: read per row from memory they were two vector loads per step — wave-uniform data, but each costing the texture
  // unit as much as a 64-lane gather (and the scalar cache is no alternative: it is not invalidated between the prep launch
  // that rewrites the table and this one).
  const uint4* __restrict__ rowtab;
  unsigned tab_x, tab_y, tab_z;
  int tab_base;
  __device__ __forceinline__ void tab_entry(int row, unsigned& o0, unsigned& o1, float& ly) const {
    const int k = row - tab_base;
    o0 = (unsigned)__builtin_amdgcn_readlane((int)tab_x, k); o1 = (unsigned)__builtin_amdgcn_readlane((int)tab_y, k);
    ly = __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)tab_z, k));
  }
  // Horizontally blended disparity of the two low-resolution rows the current image row lies between, the taps of a new lower
  // row in flight, and which rows those are.  Consecutive image rows share their low-resolution rows (a level f times smaller
  // advances once every f image rows), so a row step needs the two taps of a new lower row when the pair advances and nothing
  // otherwise.  (The texture unit charges every wave load ~16 cycles whatever its width, how many lanes are live or in range —
  // scripts/dev/ta_probe.hip — so this saves cache traffic and two of four load instructions, not four.)
  float h0, h1, p2, p3;
  unsigned cur_o0, cur_o1;   // byte offsets of the rows h0 / h1 will belong to once the pending update is applied
  bool pend;                 // finish_depth must first shift h1 -> h0 and blend the taps in flight into h1

  // (explicit fma: with -ffp-contract=fast the compiler otherwise picks, per inlined copy — prologue, first row, row loop — which
  // of the two products it fuses, and the same row then differs by an ulp depending on where in a strip it falls)
  __device__ __forceinline__ float hblend(float a, float b) const { return fmaf(dlx, b, (1.f - dlx)*a); }
  // FIRST: the first row of the strip — both rows, not pipelined.  Afterwards the pair either stays or advances by one row (the
  // launcher uses this instantiation only for pyramid levels that are not taller than the image).
  template <bool FIRST>
  __device__ __forceinline__ void load_dtaps(int row) {
    unsigned o0, o1; float ly_;
    tab_entry(row, o0, o1, ly_);
    if (FIRST) {
      const float a0 = bld(rs_disp, dx0, o0), a1 = bld(rs_disp, dx1, o0), b0 = bld(rs_disp, dx0, o1), b1 = bld(rs_disp, dx1, o1);
      h0 = hblend(a0, a1); h1 = hblend(b0, b1); pend = false;
    } else {
      // branch-free (a conditional load splits the loop into regions and costs ~20 registers): when the pair stays, the two
      // loads are sent out of range — the bounds check drops them before they reach the cache — and their zeros are ignored
      pend = (o0 != cur_o0) || (o1 != cur_o1);
      const unsigned so = pend ? o1 : 0xf0000000u;
      p2 = bld(rs_disp, dx0, so); p3 = bld(rs_disp, dx1, so);
    }
    cur_o0 = o0; cur_o1 = o1;
  }
  __device__ __forceinline__ float finish_depth(int row) {
    { const float hn = hblend(p2, p3); h0 = pend ? h1 : h0; h1 = pend ? hn : h1; }
    unsigned o0_, o1_; float ly;
    tab_entry(row, o0_, o1_, ly);
    const float val = fmaf(ly, h1, (1.f - ly)*h0);
    const float d = fmaf(a_scale, val, a_off);
    const float dep = (d > 0.f) ? __builtin_amdgcn_rcpf(fmaxf(d, kEps32)) : 0.f;
    // each row is interior to one strip.  No control flow around the store (a divergent `if` ends the row step's basic block twice): a lane or a
    // row that must not write is sent out of the buffer's range, where the bounds check drops it
    // (a lane / row that must not write gets an offset the bounds check drops.  0x80000000, not 0xffffffff: the hardware adds the vector and the
    // scalar offset, and whether that sum wraps in 32 bits is not something the probes cover — this sentinel cannot wrap into range: every plane
    // addressed here is far below 2 GiB, and even 0x80000000 + 0xf0000000 lands at 0x70000000, beyond any of them)
    bst(rs_dout, interior ? lane4 : 0x80000000u, (row >= r0 && row < r1) ? (unsigned)row*w4 : 0xf0000000u, dep);
    return dep;
  }

  // ---- coordinates + the four tap loads of one pair of supports -----------------------------------------
  template <int PS>
  __device__ __forceinline__ void issue(int g, float D, float vf) {
    MainPend<N>& P = this->P[PS];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int k = 2*g + kk;
      if (k < N) {
        const Cam2& cm = cam[k];
        const float bx = kCamLds ? camcol[(k*3 + 0)*64] : hx0[kCamLds ? 0 : k];
        const float by = kCamLds ? camcol[(k*3 + 1)*64] : hy0[kCamLds ? 0 : k];
        const float bz = kCamLds ? camcol[(k*3 + 2)*64] : hz0[kCamLds ? 0 : k];
        const float hx = fmaf(cm.H1, vf, bx), hy = fmaf(cm.H4, vf, by), hz = fmaf(cm.H7, vf, bz);
        const float nx = fmaf(D, hx, cm.a0), ny = fmaf(D, hy, cm.a1), yz = fmaf(D, hz, cm.tz);
        const float rz = __builtin_amdgcn_rcpf(fmaxf(yz, kZMin));
        const float sx = fmaf(nx, rz, -0.5f), sy = fmaf(ny, rz, -0.5f);
        const float cx = __builtin_amdgcn_fmed3f(sx, 0.f, xmax), cy = __builtin_amdgcn_fmed3f(sy, 0.f, ymax);
        const float x0 = floorf(cx), y0 = floorf(cy);
        P.fx[kk] = cx - x0; P.fy[kk] = cy - y0;
        const unsigned o = __umul24((unsigned)fmaf(y0, wpf, x0), 12u);   // texel index exact in fp32: (h+1)*(w+1) < 2^24 (checked by the C ABI)
#if (SMD_ABLATE & 1)
        const float fo = __builtin_bit_cast(float, (o & 0x7fffffu) | 0x3f000000u);
        P.t[kk][0] = f3{fo, fo*0.5f, fo*0.25f}; P.t[kk][1] = f3{fo*0.3f, fo, fo}; P.t[kk][2] = f3{fo, fo*0.7f, fo}; P.t[kk][3] = f3{fo*0.9f, fo, fo*0.1f};
#else
        P.t[kk][0] = bld3(rs_pk, o, so_tex[k]); P.t[kk][1] = bld3(rs_pk, o + 12u, so_tex[k]);
        P.t[kk][2] = bld3(rs_pk, o, so_tex[k] + rowbytes); P.t[kk][3] = bld3(rs_pk, o + 12u, so_tex[k] + rowbytes);
#endif
      }
    }
  }

  // The loads that follow the last pair of row j: first pair of row j+1, its target row, the depth of row j+2.  Issued
  // unconditionally — also after the last row of the strip, where nothing consumes them: the tap coordinates are clamped, a
  // depth row below the image reads 0 (buffer bounds), and a conditional issue would make every pending register a loop phi
  // with a second copy (36 more VGPRs).
  template <int PS>
  __device__ __forceinline__ void issue_next_row(int j) {
#if (SMD_ABLATE & 2)
    issue<PS>(0, Dnext, vfn);
    py = f3{vfn*0.001f, 0.5f, vfn*0.002f};
    Dcur = Dnext;
    Dnext = 1.f + vfn*0.01f;
#else
    const float Dn = DISP ? finish_depth(j + LA) : Dnext;
    issue<PS>(0, Dn, vfn);
#if (SMD_ABLATE & 16)
    py = f3{vfn*0.001f, 0.5f, vfn*0.002f};
#else
    if (!SH) py = bld3(rs_pk, lane4*3u, so_y + (unsigned)(j + 1)*w4*3u);
#endif
    Dcur = Dn;
    if (DISP) load_dtaps<false>(j + LA + 1); else Dnext = bld(rs_depth, lane4, (unsigned)(j + LA + 1)*w4);
#endif
    vfn += 1.f;
  }

  template <int PS>
  __device__ __forceinline__ void finish(int g, float (&Xn)[N][3], int j) {
    const MainPend<N>& P = this->P[PS];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int k = 2*g + kk;
      if (k < N) {
        const float w11 = P.fx[kk]*P.fy[kk], w01 = P.fx[kk] - w11, w10 = P.fy[kk] - w11, w00 = (1.f - P.fx[kk]) - w10;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          Xn[k][c] = fmaf(w11, P.t[kk][3][c], fmaf(w10, P.t[kk][2][c], fmaf(w01, P.t[kk][1][c], w00*P.t[kk][0][c])));
      }
    }
    if (AUX && want_w0 && j >= r0 && j < r1 && interior) {     // loss_dict['supp_imgs_warp'] (scale 0 only; logging path)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int k = 2*g + kk;
        if (k < N) {
          float* wo = a.warp0 + ((size_t)(a.i0 + k)*a.b + bi)*3*(size_t)(hw4/4u) + (size_t)j*w + (lane4 >> 2);
#pragma unroll
          for (int c = 0; c < 3; ++c) wo[(size_t)c*(hw4/4u)] = Xn[k][c];
        }
      }
    }
  }

  // One row step: row j is the NEW row, centre row v = j-1.  EMIT = false only updates the sliding sums.
  // VIRT (j == h): the new row is the reflected row h-2: the target row is read again, the warped row is P - x(h-1).
  // PS: the pending set tied to Xn (LA = 2: 0 for XA, 1 for XB; LA = 1: always 0)
  template <bool EMIT, bool VIRT, int PS>
  __device__ __forceinline__ void step(int j, float (&Xo)[N][3], float (&Xn)[N][3], float (&Yo)[3], float (&Yn)[3]) {
    const int v = j - 1;
    constexpr float c1 = 81.f*kC1;
    f4 t0 = {0.f, 0.f, 0.f, 0.f};
    f3 t1 = {0.f, 0.f, 0.f};
    float nz = 0.f;
    if (EMIT && !SH) {   // what the centre row shares across scales and supports: {S_y[3], cy2[3], identity error, -}
      const unsigned to = (unsigned)v*w4*4u;
#if (SMD_ABLATE & (2 | 8))
      t0 = f4{4.5f + vfn*0.01f, 4.4f, 4.3f, 0.2f}; t1 = f3{0.25f, 0.3f, 0.05f}; (void)to;
#else
      if (SSIM) t0 = bld4(rs_pk, lane4*4u, so_ta + to);
      t1 = bld3(rs_pk, lane4*4u, so_tb + to);
#endif
      if (AUX && has_noise) nz = nz_sb[(size_t)v*w + (lane4 >> 2)];
    }
    if (!SH && !VIRT) { Yn[0] = py.x; Yn[1] = py.y; Yn[2] = py.z; }
    if (VIRT) { const f3 yy = bld3(rs_pk, lane4*3u, so_y + (unsigned)(h - 2)*w4*3u); Yn[0] = yy.x; Yn[1] = yy.y; Yn[2] = yy.z; }
    const float m = (v == 0) ? 2.f : 1.f;                 // row -1 is row 1: the new row counts twice for the first image row
    float best = 0.f, acc = 0.f;
    int bsel = a.i0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (!VIRT) finish<PS>(g, Xn, j);
      else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) { const int k = 2*g + kk; if (k < N) {
#pragma unroll
          for (int c = 0; c < 3; ++c) Xn[k][c] = Px[k][c] - Xo[k][c]; } }
      }
      if (SH && g == 0) {
        // the taps of this row have just been waited for.  First step of an epoch: meet the other scales, then send for this
        // wave's row of the next epoch; every step: this row's slot of the ring.
        if ((j & 3) == 0) {
          epoch_sync(); dma_epoch((j >> 2) + 1);
#if SMD_FWD_PRIO
          // issue priority by the rows a wave still has to do (the arbiter otherwise serves the OLDEST wave of a SIMD first: the four waves of
          // a SIMD then finish one after the other, and at the end of the launch the last ones run alone, latency-bound; wave traces in
          // profiles/r04_fwd_wave_traces.txt): a wave with more rows ahead of it goes first, so co-resident waves finish closer together
          const int rem = jlast - j;
#if SMD_FWD_PRIO == 2   // the inverse: the fewer rows left, the higher the priority (oldest-first, made stronger)
          if (rem < 4) __builtin_amdgcn_s_setprio(3); else if (rem < 8) __builtin_amdgcn_s_setprio(2); else if (rem < 12) __builtin_amdgcn_s_setprio(1);
#else
          if (rem < 4) __builtin_amdgcn_s_setprio(0); else if (rem < 8) __builtin_amdgcn_s_setprio(1); else if (rem < 12) __builtin_amdgcn_s_setprio(2);
#endif
#endif
        }
        const float* slot = ring_lane + (j & 7)*kRingSlotFloats;
        if (!VIRT) { const f4 yv = *reinterpret_cast<const f4*>(slot); Yn[0] = yv.x; Yn[1] = yv.y; Yn[2] = yv.z; }
        if (EMIT) {
          t0 = *reinterpret_cast<const f4*>(slot + 256);
          const f4 tv = *reinterpret_cast<const f4*>(slot + 512);
          t1 = f3{tv.x, tv.y, tv.z};
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // keep the memory pipe busy: next pair of this row, or the first pair of the next row
      if (!VIRT) {
        if (g + 1 < NG) issue<PS>(g + 1, Dcur, vfn - 1.f);
        else issue_next_row<PS>(j);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int k = 2*g + kk;
        if (k < N) {
          float es = 0.f, el = 0.f;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float xn = Xn[k][c], xo = Xo[k][c];
            if (EMIT) el += fabsf(xo - Yo[c]);
            if (SSIM) {
              const float xx = xn*xn, xy = xn*Yn[c];
              const float Vx = fmaf(m, xn, Px[k][c]), Vxx = fmaf(m, xx, Pxx[k][c]), Vxy = fmaf(m, xy, Pxy[k][c]);
              Px[k][c] = xo + xn; Pxx[k][c] = fmaf(xo, xo, xx); Pxy[k][c] = fmaf(xo, Yo[c], xy);
              if (EMIT) {
                float sx, sxx, sxy;
                hsum3(Vx, Vxx, Vxy, sx, sxx, sxy);
                const float sy = t0[c], cy2 = (c == 0) ? t0.w : ((c == 1) ? t1.x : t1.y);
                es += ssim_err81(sx, sxx, sxy, sy, fmaf(sy, sy, c1), cy2);
              }
            }
          }
          if (EMIT) {
            const float e = SSIM ? fmaf(kWSsim/3.f, es, ((1.f - kWSsim)/3.f)*el) : el*(1.f/3.f);
            if (k == 0) { best = e; acc = e; }
            else { acc += e; if (e < best) { best = e; bsel = a.i0 + k; } }
          }
        }
      }
    }
    if (EMIT) {
      // every lane runs this (the halo lanes' values are finite and go nowhere): only the stores and the loss sum look at `interior`, through
      // an out-of-range lane offset / a select — no exec-mask region in the row step except the rare tie-break one
      const unsigned lane1 = lane4 >> 2, cro = (unsigned)v*w4, cro1 = (unsigned)v*(unsigned)w;
      const unsigned st4 = interior ? lane4 : 0x80000000u, st1 = interior ? lane1 : 0x80000000u;   // (0x80000000 + row offset cannot wrap into range: see the depth store)
      if (!SINGLE && !a.first_pass) {
        const float prev = bld(rs_err, lane4, cro);
        if (use_min) { if (!(best < prev)) { best = prev; bsel = (int)bld8(rs_sel, lane1, cro1); } }
        else acc += prev;
      }
      if (!SINGLE && !a.last_pass) {
        bst(rs_err, st4, cro, use_min ? best : acc);
        bst8(rs_sel, st1, cro1, (unsigned)bsel);
      } else {
        float e = use_min ? best : acc*a.inv_n;
        if (!use_min) bsel = 0;
        float est = automask ? t1.z : __builtin_inff();
        // the tie-break noise is eps*N(0,1): it can only matter when the two errors are within a few eps of each other
        if (AUX && has_noise) est = fmaf(kEps32, nz, est);
        else if (fabsf(est - e) < 1e-5f)
          est = fmaf(kEps32, gauss_noise(a.seed_lo, a.seed_hi, (uint32_t)(((unsigned)s*(unsigned)a.b + (unsigned)bi)*(hw4 >> 2) + cro1 + lane1)), est);
        if (est < e) { e = est; bsel = SMD_SEL_MASKED; }
#if (SMD_ABLATE & 4)
        if (e == 123.456f) bst(rs_err, st4, cro, e + (float)bsel);
#else
        if (has_err) bst(rs_err, st4, cro, e);      // the error map is an optional output (logging / tests): one store less per row
        bst8(rs_sel, st1, cro1, (unsigned)bsel);
#endif
#pragma unroll
        for (int k = 0; k < NT; ++k) livem[k] |= __ballot(bsel == k);   // one v_cmp + one scalar or per support (halo lanes included: they hold real neighbours)
        lsum += interior ? e : 0.f;
      }
    }
  }

  template <int PS>
  __device__ __forceinline__ void init(float (&X)[N][3], float (&Y)[3], int j) {   // P = r(jstart) alone
    Y[0] = py.x; Y[1] = py.y; Y[2] = py.z;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      finish<PS>(g, X, j);
      if (g + 1 < NG) issue<PS>(g + 1, Dcur, vfn - 1.f);
      else issue_next_row<PS>(j);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) { const int k = 2*g + kk; if (k < N) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { Px[k][c] = X[k][c]; Pxx[k][c] = X[k][c]*X[k][c]; Pxy[k][c] = X[k][c]*Y[c]; } } }
    }
  }
  // Everything in flight before the first row is blended: the taps of row jstart into set PSF (the set of the buffer `init` fills),
  // with LA = 2 also those of row jstart + 1 into the other set, the target row jstart, the depth pipeline LA rows ahead.
  template <int PSF>
  __device__ __forceinline__ void prologue(int jstart) {
    if (SH) {   // the epoch of the first row step (r0 = 0: the loop starts at row 1 and only meets an epoch boundary at row 4 — both epochs)
      dma_epoch(r0 >> 2);
      if (r0 == 0) dma_epoch(1);
    }
    vfn = (float)jstart;
    if (DISP) { load_dtaps<true>(jstart); Dnext = finish_depth(jstart); }
    else Dnext = bld(rs_depth, lane4, (unsigned)jstart*w4);
    Dcur = Dnext;
    issue<PSF>(0, Dnext, vfn);
    py = bld3(rs_pk, lane4*3u, so_y + (unsigned)jstart*w4*3u);
    vfn += 1.f;
    if (LA == 2) {
      float D1;
      if (DISP) { load_dtaps<false>(jstart + 1); D1 = finish_depth(jstart + 1); }
      else D1 = bld(rs_depth, lane4, (unsigned)(jstart + 1)*w4);
      issue<(LA == 2) ? 1 - PSF : 0>(0, D1, vfn);
      vfn += 1.f;
    }
    if (DISP) load_dtaps<false>(jstart + LA); else Dnext = bld(rs_depth, lane4, (unsigned)(jstart + LA)*w4);
    if (SH) epoch_sync();   // (waits for what `init` needs next anyway)
  }
};

template <int N, bool SSIM, bool SINGLE, bool AUX, bool DISP, int LA, bool SH>
__device__ __forceinline__ float recon_main_body(const ReconMainArgs& a) {   // -> this lane's share of the loss sum (0 for a wave without a strip)
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int strip, bi_, s_;
  // segment of the (possibly tapered) partition this block belongs to
  const unsigned nblk1 = recon_grid_blocks(a.nsx*a.nsy, a.b1, a.S);
  const bool tail = blockIdx.x >= nblk1;
  const int nstr = a.nsx*(tail ? a.nsy2 : a.nsy), seg_b = tail ? a.b - a.b1 : a.b1, seg_rh = tail ? a.rh2 : a.rh;
  if (SH) {   // block = the four scales of one strip; the XCD-aware decode places four adjacent strips on consecutive blocks of an XCD
    int xb, sub;
    decode_tile(tail ? blockIdx.x - nblk1 : blockIdx.x, ceil_div(nstr, kWavesPerBlock), seg_b, kWavesPerBlock, xb, bi_, sub);
    strip = xb*kWavesPerBlock + sub; s_ = wid;
  } else decode_wave(tail ? blockIdx.x - nblk1 : blockIdx.x, wid, nstr, seg_b, a.S, strip, bi_, s_);
  if (strip >= nstr) return 0.f;   // (SH: the whole block)
  if (tail) bi_ += a.b1;
  const int sxi = strip % a.nsx, syi = strip/a.nsx;

  using Ctx = MainCtx<N, SSIM, SINGLE, AUX, DISP, LA, SH>;
  __shared__ __attribute__((aligned(16))) float ring_mem[SH ? kRingFloats : 4];
  __shared__ float cam_lds[Ctx::kCamLds ? kWavesPerBlock*N*3*64 : 1];   // own-lane columns: written and read by the same lane, no synchronisation
  Ctx cx{a};
  cx.camcol = cam_lds + (Ctx::kCamLds ? wid*N*3*64 + lane : 0);
  cx.bi = bi_; cx.s = s_; cx.h = a.h; cx.w = a.w;
  cx.r0 = syi*seg_rh; cx.r1 = min(cx.r0 + seg_rh, a.h);
  cx.jlast = min(cx.r1, a.h - 1);
  const int u = sxi*kFwdCols - 1 + lane;
  // The pixel column this lane synthesises: its own, or — for the halo lane just outside the image — the reflected one.
  const int uc = (u < 0) ? min(-u, a.w - 1) : ((u >= a.w) ? max(2*(a.w - 1) - u, 0) : u);
  cx.lane4 = (unsigned)uc*4u;
  cx.interior = (lane >= 1) && (lane <= kFwdCols) && (u < a.w);
  cx.use_min = a.flags & SMD_USE_MIN;
  cx.automask = a.flags & SMD_USE_AUTOMASK;
  cx.has_noise = cx.automask && a.noise != nullptr;
  cx.want_w0 = a.warp0 != nullptr && s_ == 0;
  const size_t hw = (size_t)a.h*a.w;
  cx.hw4 = (unsigned)hw*4u; cx.w4 = (unsigned)a.w*4u;
  cx.rowbytes = ((unsigned)a.w + 1u)*12u;
  cx.xmax = (float)(a.w - 1); cx.ymax = (float)(a.h - 1); cx.wpf = (float)(a.w + 1);
  const float uf = (float)uc;
  const unsigned texel_bytes = (unsigned)(a.h + 1)*(unsigned)(a.w + 1)*12u;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const int i = a.i0 + k;
    float bx, by, bz;
    make_cam2(cx.cam[k], bx, by, bz, a.T + ((size_t)i*a.b + bi_)*16, a.K + (size_t)bi_*16, a.Kinv + (size_t)bi_*16, a.wscale, a.hscale, uf);
    if (Ctx::kCamLds) { cam_lds[(wid*N + k)*3*64 + lane] = bx; cam_lds[((wid*N + k)*3 + 1)*64 + lane] = by; cam_lds[((wid*N + k)*3 + 2)*64 + lane] = bz; }
    else { cx.hx0[Ctx::kCamLds ? 0 : k] = bx; cx.hy0[Ctx::kCamLds ? 0 : k] = by; cx.hz0[Ctx::kCamLds ? 0 : k] = bz; }
    cx.so_tex[k] = (unsigned)(i*a.b + bi_)*texel_bytes;
  }
  const size_t sb = ((size_t)s_*a.b + bi_)*hw;
  cx.so_y = (unsigned)(packed_texel_floats(a.b, a.n, a.h, a.w)*4) + (unsigned)bi_*cx.hw4*3u;
  cx.so_ta = (unsigned)((packed_texel_floats(a.b, a.n, a.h, a.w) + packed_ypix_floats(a.b, a.h, a.w))*4) + (unsigned)bi_*cx.hw4*4u;
  cx.so_tb = cx.so_ta + (unsigned)(packed_tpix_floats(a.b, a.h, a.w)*4);
  cx.rs_pk = make_rsrc(a.packed, packed_image_floats(a.b, a.n, a.h, a.w)*4);
  cx.rs_depth = make_rsrc(DISP ? nullptr : a.depth + sb, DISP ? 0 : hw*4);
  if (DISP) {
    const int hs = a.sc.hs[s_], ws = a.sc.ws[s_];
    cx.rs_disp = make_rsrc(a.sc.p[s_] + (size_t)bi_*hs*ws, (size_t)hs*ws*4);
    cx.rs_dout = make_rsrc(a.depth_out + sb, hw*4);
    int x0, x1;
    src_index_f(uc, (float)ws/(float)a.w, ws, x0, x1, cx.dlx);
    cx.dx0 = (unsigned)x0*4u; cx.dx1 = (unsigned)x1*4u;
    cx.a_scale = a.a_scale; cx.a_off = a.a_off;
    cx.rowtab = a.rowtab + (size_t)s_*(a.h + 4);
    cx.tab_base = max(cx.r0 - 1, 0);                       // rows tab_base .. r1 + LA + 1 are looked up: at most rh + LA + 3 <= 64 entries
    { const uint4 e = cx.rowtab[min(cx.tab_base + lane, a.h + 3)]; cx.tab_x = e.x; cx.tab_y = e.y; cx.tab_z = e.z; }
  }
  cx.nz_sb = (AUX && a.noise) ? a.noise + sb : nullptr;
  cx.has_err = a.err != nullptr;
  cx.rs_err = make_rsrc(cx.has_err ? a.err + sb : nullptr, cx.has_err ? hw*4 : 0);
  cx.rs_sel = make_rsrc(a.sel + sb, hw);
  cx.lsum = 0.f;
#pragma unroll
  for (int k = 0; k < Ctx::NT; ++k) cx.livem[k] = 0ull;
#if SMD_FWD_PRIO == 1
  if (SH) __builtin_amdgcn_s_setprio(3);
#endif
  if (SH) {
    cx.ring_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)ring_mem);
    cx.ring_lane = ring_mem + lane*4;
    cx.ring_q = wid;
    cx.ring_last = (cx.r1 == a.h) ? a.h : cx.jlast;
  }

  // The pending set of a row is tied to the buffer the row is blended into: XA <-> set 0, XB <-> set (LA == 2 ? 1 : 0).
  constexpr int SA = 0, SB = (LA == 2) ? 1 : 0;
  const int jstart = max(cx.r0 - 1, 0);
  float XA[N][3], XB[N][3], YA[3], YB[3];
  int j = jstart + 1;
  if (cx.r0 > 0) {
    cx.template prologue<SB>(jstart);
    cx.template init<SB>(XB, YB, jstart);
    cx.template step<false, false, SA>(j, XB, XA, YB, YA);
    ++j;
  } else {
    cx.template prologue<SA>(jstart);
    cx.template init<SA>(XA, YA, jstart);
  }
  bool cur_is_a = true;
  for (; j <= cx.jlast; j += 2) {
    cx.template step<true, false, SB>(j, XA, XB, YA, YB);
    if (j + 1 > cx.jlast) { cur_is_a = false; break; }
    cx.template step<true, false, SA>(j + 1, XB, XA, YB, YA);
  }
  if (cx.r1 == a.h) {   // the strip owns the last image row: one more step whose new row is the reflected row h-2
    if (!cur_is_a) {
#pragma unroll
      for (int c = 0; c < 3; ++c) YA[c] = YB[c];
#pragma unroll
      for (int k = 0; k < N; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) XA[k][c] = XB[k][c];
    }
    cx.template step<true, true, SB>(a.h, XA, XB, YA, YB);
  }
  if (a.live != nullptr && (SINGLE || a.last_pass)) {   // this strip's entry of the liveness table: lane k stores support k's column mask (zeros beyond NT)
    unsigned long long mine = 0ull;
#pragma unroll
    for (int k = 0; k < Ctx::NT; ++k) mine = (lane == k) ? cx.livem[k] : mine;
    unsigned long long* tab = reinterpret_cast<unsigned long long*>(a.live + live_header_floats(a.b));
    if (lane < kLiveSupports) tab[(((size_t)s_*a.b + bi_)*live_max_strips(a.h, a.w) + (size_t)strip)*kLiveSupports + lane] = mine;
  }

  return cx.lsum;
}

// In-launch loss reduction (round 3; the former k_sum_partials launch).  Deterministic: a block's partial is the sum of its
// waves' sums in wave order, the loss the fp64 sum of the block partials in a fixed order — whichever wave happens to do it.
// No block barrier at the end of the kernel (every wave of a block would idle through two memory round trips): a wave parks
// its sum in LDS and bumps an LDS counter; only the wave that arrives LAST in its block goes on — it publishes the block's
// partial write-through (agent scope), drains, counts the block's arrival at agent scope, and if the block is the last of
// the launch it acquires and reduces every partial, alone, 64 lanes wide (cdna_hip_programming.md, Guideline 16).
struct MainTail { double wsum[kWavesPerBlock]; unsigned arrived; };

__device__ __forceinline__ void recon_main_reduce(const ReconMainArgs& a, MainTail& tl, float lane_sum) {
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float tot = wave_sum(lane_sum);
  unsigned old = 0;
  if (lane == 0) {
    tl.wsum[wid] = (double)tot;
    old = __hip_atomic_fetch_add(&tl.arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);   // orders the LDS store before, the LDS loads after
  }
  old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
  if (old != (unsigned)kWavesPerBlock - 1u) return;        // not the last wave of the block
  unsigned last = 0;
  if (lane == 0) {
    double bsum = 0.0;
#pragma unroll
    for (int k = 0; k < kWavesPerBlock; ++k) bsum += tl.wsum[k];
    __hip_atomic_store((unsigned long long*)a.partial + blockIdx.x, __builtin_bit_cast(unsigned long long, bsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = (__hip_atomic_fetch_add(a.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)a.main_blocks - 1u) ? 1u : 0u;
  }
  if (!__builtin_amdgcn_readfirstlane((int)last)) return;
  SMD_TAIL_ACQUIRE();
  // The sweep is pure latency (this wave runs alone at the very end of the launch): 16-byte agent-scope loads, sixteen of them
  // per lane issued before the first is used — one round trip per 2048 partials.  Beyond the last partial the buffer reads 0.
  const unsigned bytes = (unsigned)a.main_blocks*8u;   // (the grid may carry guest blocks behind the main ones: they have no partial)
  const rsrc_t rs = make_rsrc(a.partial, bytes);
  typedef double d2 __attribute__((ext_vector_type(2)));
  double acc = 0.0;
  for (unsigned base = 0; base < bytes; base += 16u*1024u) {
    d2 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rs, base + (unsigned)q*1024u + (unsigned)lane*16u, 0, 16));   // aux 16 = sc1: agent scope
#pragma unroll
    for (int q = 0; q < 16; q += 4) acc += ((v[q].x + v[q].y) + (v[q + 1].x + v[q + 1].y)) + ((v[q + 2].x + v[q + 2].y) + (v[q + 3].x + v[q + 3].y));   // fixed order
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) {
    const float l_rec = (float)(acc*a.loss_scale);
    a.loss[0] = l_rec;
    __hip_atomic_store(a.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch on this buffer
    if (a.comb.out3) loss_combine_arrive(a.comb, 1, l_rec);   // fused loss path: the weighted sum with the smoothness term, by whoever is second
  }
}

// register budget: up to two supports are held to 128 VGPRs (4 waves per SIMD: the K0-fused instantiation uses 126, nothing spilled), three and four to
// 168 (3 waves; the four-support instantiation uses 158).  A fifth wave (96 VGPRs) spills 28 values: profiles/r04_fwd_shape_sweep.txt
#ifdef SMD_TRACE_WAVES   // diagnosis builds only (scripts/dev/wave_trace.py): when and where every wave of the last launch ran
__device__ unsigned long long g_wave_trace[1 << 16][3];
#endif
template <int N, bool SSIM, bool SINGLE, bool AUX, bool DISP, int LA = 1, bool SH = false>
__global__ __launch_bounds__(64*kWavesPerBlock, ((N <= 2 && LA == 1) ? 4 : 3)) void k_recon_main(const ReconMainArgs a) {
#ifdef SMD_TRACE_WAVES
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#endif
  __shared__ MainTail tail;
  const bool reduce = (SINGLE || a.last_pass) && a.partial != nullptr;
  if (reduce) {                       // the only block barrier, at the start, where every wave still is
    if (threadIdx.x == 0) tail.arrived = 0u;
    __syncthreads();
  }
  if (DISP && SINGLE && !AUX) {
    // Fused loss path (round 5): the blocks behind the main ones are GUESTS running the smoothness sweep over the same disparity pyramid
    // (smd_smooth_dev.h) — dispatched last, they fill the drain of this launch (its final fifth runs at one or two waves per SIMD) and their
    // reduction chain ends long before the main blocks do.
    if (a.guest_blocks != 0 && (int)blockIdx.x >= a.main_blocks) { smooth_main_block(a.sc, a.b, a.sm, (int)blockIdx.x - a.main_blocks); return; }
  }
  const float lane_sum = recon_main_body<N, SSIM, SINGLE, AUX, DISP, LA, SH>(a);
#ifdef SMD_TRACE_WAVES
  if ((threadIdx.x & 63) == 0) {
    const unsigned widx = blockIdx.x*kWavesPerBlock + (threadIdx.x >> 6);
    if (widx < (1u << 16)) {
      g_wave_trace[widx][0] = t0; g_wave_trace[widx][1] = __builtin_amdgcn_s_memrealtime();
      g_wave_trace[widx][2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    }
  }
#endif
  if (reduce) recon_main_reduce(a, tail, lane_sum);
}
#ifdef SMD_TRACE_WAVES
extern "C" int smd_debug_wave_trace(unsigned long long* host_out, int max_waves) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wave_trace), (size_t)max_waves*3*sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

// A launch records which instantiation it is (smd_last_kernel_variant): the label bench.py prints is the kernel that ran.
template <int N, bool SSIM, bool SINGLE, bool AUX, bool DISP, int LA = 1, bool SH = false>
static void launch_main_t(dim3 grid, dim3 block, hipStream_t st, const ReconMainArgs& a) {
  auto tf = [](bool v) { return v ? "true" : "false"; };
  note_variant(0, "smd::k_recon_main<%d, %s, %s, %s, %s, %d, %s>", N, tf(SSIM), tf(SINGLE), tf(AUX), tf(DISP), LA, tf(SH));
  hipLaunchKernelGGL((k_recon_main<N, SSIM, SINGLE, AUX, DISP, LA, SH>), grid, block, 0, st, a);
}

int recon_main_blocks(const ReconMainArgs& a) {
  return (int)(recon_grid_blocks(a.nsx*a.nsy, a.b1, a.S) + (a.b1 < a.b ? recon_grid_blocks(a.nsx*a.nsy2, a.b - a.b1, a.S) : 0u));
}

hipError_t launch_recon_main(const ReconMainArgs& a_in, hipStream_t st) {
  ReconMainArgs a = a_in;
  a.main_blocks = recon_main_blocks(a);
  const bool ssim = !(a.flags & SMD_LOSS_L1);
  {  // guest blocks exist in the K0-fused single-pass instantiations only
    const bool hot = ssim && a.first_pass && a.last_pass && a.warp0 == nullptr && a.noise == nullptr && a.depth_out != nullptr;
    if (a.guest_blocks != 0 && !hot) return hipErrorInvalidValue;
  }
  dim3 grid((unsigned)(a.main_blocks + a.guest_blocks)), block(64*kWavesPerBlock);
  const bool single = a.first_pass && a.last_pass;
  const bool aux = a.warp0 != nullptr || a.noise != nullptr;
  const bool disp = a.depth_out != nullptr;     // K0 fused: only on the first pass over the supports, SSIM instantiations
  // hot: every support in one launch, no extras; otherwise the general instantiation (carried min / sum, noise tensor, warp output)
#define SMD_MAIN(N_) do { \
    if (ssim && single && !aux) { \
      if (disp && N_ <= 2 && a.lookahead == 2) launch_main_t<(N_ <= 2 ? N_ : 2), true, true, false, true, 2>(grid, block, st, a); \
      else if (disp && a.share) launch_main_t<N_, true, true, false, true, 1, true>(grid, block, st, a); \
      else if (disp) launch_main_t<N_, true, true, false, true>(grid, block, st, a); \
      else launch_main_t<N_, true, true, false, false>(grid, block, st, a); \
    } else if (ssim) { \
      if (disp) launch_main_t<N_, true, false, true, true>(grid, block, st, a); \
      else launch_main_t<N_, true, false, true, false>(grid, block, st, a); \
    } else launch_main_t<N_, false, false, true, false>(grid, block, st, a); } while (0)
  switch (a.ni) { case 1: SMD_MAIN(1); break; case 2: SMD_MAIN(2); break; case 3: SMD_MAIN(3); break; default: SMD_MAIN(4); break; }
#undef SMD_MAIN
  return hipGetLastError();
}

}  // namespace smd
