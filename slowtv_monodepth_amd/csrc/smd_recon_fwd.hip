// smd_recon_fwd.hip — fused forward of the view-synthesis photometric loss for gfx950.
//
// One launch covers every (strip, sample, scale).  A wave owns 64 consecutive columns (62 interior + 1 halo
// lane per side) of a `rh`-row strip and streams down the rows:
//   per row    : depth + target row (coalesced), per support: 6 FMAs of projective geometry, one rcp, four aligned
//                16-byte gathers from the RGBX-repacked support frame, bilinear blend             (K1c-K1g of SURVEY §2.2)
//   pipeline   : the loads of row j+1 (and the depth of row j+2) are issued before the SSIM math of row j-1, so the
//                dependent chain depth -> coordinates -> gather never stalls the wave on two memory latencies per row
//   horizontal : 3-tap sums of {x, x^2, xy} (and {y, y^2}) through DPP wave shifts folded into v_add_f32_dpp; the
//                reflection padding costs nothing: halo lanes outside the image synthesise the reflected column
//   vertical   : a 3-row ring of the raw pixel values in registers;
//                vertical taps are summed per lane, then the horizontal taps: the 3x3 SSIM window never touches
//                LDS or HBM                                                                      (K2a-K2c)
//   per pixel  : SSIM + L1, min/mean over supports in registers, automask against the identity error,
//                error + selection written once, loss reduced per wave                           (K2d-K2f)
// The identity ("static") error does not depend on the scale, so it is produced once per sample by the same
// template with WARP = false instead of S times as in the reference (reconstruction.py:71).
//
// Template parameters: NI supports held in registers per pass (n > 2 runs several passes that carry the running
// min / sum through `err`/`sel`), WARP (false = identity error), SSIM (false = loss_name 'l1'), SINGLE (one pass:
// no carried state is read or written).
#include "smd_common.h"
#include "smd_kernels.h"

namespace smd {

template <int NI>
struct RowState {          // raw pixel values of one image row (one column per lane)
  float yc[3];             // target
  float xc[NI][3];         // warped (or, for the identity error, un-warped) support
};

template <int NI, bool WARP>
struct Pending {           // loads in flight for the NEXT row (software pipeline: issued one row ahead of their use)
  float y[3];
  f4 t[NI][WARP ? 4 : 1];  // WARP: the 2x2 bilinear taps (NW, NE, SW, SE) as RGBX texels; !WARP: the un-warped support texel
  float fx[NI], fy[NI];
};

// Horizontal 3-tap sum through DPP wave shifts (the compiler folds each shift into a v_add_f32_dpp).  No weights are
// needed for the reflection padding: the halo lane left of column 0 (right of column w-1) synthesises column 1
// (w-2) itself, i.e. it holds the reflected value.
// Written as asm blocks so that each shift stays fused into its add (v_add_f32_dpp) and the pairs stay adjacent: left to
// itself the compiler emits all 48 shifts of a row first (as v_mov_b32_dpp) and keeps their results live, which costs
// a wave of occupancy.  Two or three independent sums share one block: one s_nop covers the VALU-write -> DPP-read
// hazard (2 wait states) that the compiler cannot see inside asm, and the interleaving hides the add latency.
#define SMD_DPP_SHR " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define SMD_DPP_SHL " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__device__ __forceinline__ void hsum2(float a, float b, float& ra, float& rb) {
#ifdef SMD_NO_DPP
  ra = (a + lane_left(a)) + lane_right(a); rb = (b + lane_left(b)) + lane_right(b);
#else
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %2, %2" SMD_DPP_SHR
               "v_add_f32_dpp %1, %3, %3" SMD_DPP_SHR
               "v_add_f32_dpp %0, %2, %0" SMD_DPP_SHL
               "v_add_f32_dpp %1, %3, %1" SMD_DPP_SHL
               : "=&v"(ra), "=&v"(rb) : "v"(a), "v"(b));
#endif
}
__device__ __forceinline__ void hsum3(float a, float b, float c, float& ra, float& rb, float& rc) {
#ifdef SMD_NO_DPP
  ra = (a + lane_left(a)) + lane_right(a); rb = (b + lane_left(b)) + lane_right(b); rc = (c + lane_left(c)) + lane_right(c);
#else
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %3, %3" SMD_DPP_SHR
               "v_add_f32_dpp %1, %4, %4" SMD_DPP_SHR
               "v_add_f32_dpp %2, %5, %5" SMD_DPP_SHR
               "v_add_f32_dpp %0, %3, %0" SMD_DPP_SHL
               "v_add_f32_dpp %1, %4, %1" SMD_DPP_SHL
               "v_add_f32_dpp %2, %5, %2" SMD_DPP_SHL
               : "=&v"(ra), "=&v"(rb), "=&v"(rc) : "v"(a), "v"(b), "v"(c));
#endif
}

template <int NI, bool WARP, bool SSIM, bool SINGLE>
struct FwdCtx {
  const ReconFwdArgs& a;
  int lane, bi, s, h, w, u, uc, r0, r1;
  bool interior, use_min, automask;
  float wl, wr, uf;
  unsigned hw;
  Cam cam[NI];
  const float* splane[NI];
  const float* tgt_b;
  const float* depth_sb;
  unsigned out_base;
  float lsum;

  // ---- stage 1: issue the loads of row j (depth value D already in a register) -------------------
  __device__ __forceinline__ void issue_row(Pending<NI, WARP>& P, int j, float D) {
    const unsigned ro = (unsigned)j*(unsigned)w + (unsigned)uc;
#pragma unroll
    for (int c = 0; c < 3; ++c) P.y[c] = ld1(tgt_b, c*hw + ro);
    const float vf = (float)j;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      if (WARP) {
        const Cam& cm = cam[k];
        float hx = fmaf(cm.H[0], uf, fmaf(cm.H[1], vf, cm.H[2]));
        float hy_ = fmaf(cm.H[3], uf, fmaf(cm.H[4], vf, cm.H[5]));
        float hz = fmaf(cm.H[6], uf, fmaf(cm.H[7], vf, cm.H[8]));
        float nx = fmaf(D, hx, cm.a0), ny = fmaf(D, hy_, cm.a1), yz = fmaf(D, hz, cm.tz);
        float rz = __builtin_amdgcn_rcpf(fmaxf(yz, kZMin));
        float sx = fmaf(nx*rz, a.wscale, -0.5f), sy = fmaf(ny*rz, a.hscale, -0.5f);
        Taps tp = make_taps(sx, sy, h, w);
        P.fx[k] = tp.fx; P.fy[k] = tp.fy;
        const unsigned o = (unsigned)tp.off;
        P.t[k][0] = ld4(splane[k], o); P.t[k][1] = ld4(splane[k], o + 1u);
        P.t[k][2] = ld4(splane[k], o + (unsigned)w); P.t[k][3] = ld4(splane[k], o + (unsigned)w + 1u);
      } else {
        P.t[k][0] = ld4(splane[k], ro);
      }
    }
  }

  // ---- stage 2: consume the loads -> raw pixel values of row j ------------------------------------
  __device__ __forceinline__ void finish_row(RowState<NI>& R, const Pending<NI, WARP>& P, int j) {
#pragma unroll
    for (int c = 0; c < 3; ++c) R.yc[c] = P.y[c];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      if (WARP) {
        const float w11 = P.fx[k]*P.fy[k], w01 = P.fx[k] - w11, w10 = P.fy[k] - w11, w00 = (1.f - P.fx[k]) - w10;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          R.xc[k][c] = fmaf(w11, P.t[k][3][c], fmaf(w10, P.t[k][2][c], fmaf(w01, P.t[k][1][c], w00*P.t[k][0][c])));
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) R.xc[k][c] = P.t[k][0][c];
      }
      if (WARP && a.warp0 != nullptr && s == 0 && interior && j >= r0 && j < r1) {
        float* wo = a.warp0 + ((size_t)(a.i0 + k)*a.b + bi)*3*hw + (size_t)j*w + u;
#pragma unroll
        for (int c = 0; c < 3; ++c) wo[(size_t)c*hw] = R.xc[k][c];
      }
    }
  }

  // ---- emit row v from the ring (A = row v-1, B = row v, C = row v+1) --------------------------
  // Vertical 3-tap sums first (per lane, from the raw ring), then the horizontal taps through DPP: the ring holds
  // 3 + 3*NI values per row instead of the 6 + 9*NI horizontal sums.  Reflection in y is done by the caller
  // (A := C for the first image row, C := A for the last), so every window is a plain sum.
  __device__ __forceinline__ float vsum(float qa, float qb, float qc) { return (qa + qb) + qc; }
  __device__ __forceinline__ float vdot(float pa, float qa, float pb, float qb, float pc, float qc) { return fmaf(pc, qc, fmaf(pa, qa, pb*qb)); }

  __device__ __forceinline__ void emit_row(const RowState<NI>& A, const RowState<NI>& B, const RowState<NI>& C, int v) {
    // window sums are kept un-normalised (x9): ssim = N/D with both N and D scaled by 81*81
    constexpr float c1 = 81.f*kC1, c2 = 81.f*kC2;
    float sy[3], cy1[3], cy2[3];
    if (SSIM) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float s1, s2;
        hsum2(vsum(A.yc[c], B.yc[c], C.yc[c]), vdot(A.yc[c], A.yc[c], B.yc[c], B.yc[c], C.yc[c], C.yc[c]), s1, s2);
        sy[c] = s1; cy1[c] = fmaf(s1, s1, c1); cy2[c] = fmaf(9.f, s2, c2) - s1*s1;
      }
    }
    float best = 0.f, acc = 0.f;
    int bsel = a.i0;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      float es = 0.f, el = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        el += fabsf(B.xc[k][c] - B.yc[c]);
        if (SSIM) {
          float sx, sxx, sxy;
          hsum3(vsum(A.xc[k][c], B.xc[k][c], C.xc[k][c]),
                vdot(A.xc[k][c], A.xc[k][c], B.xc[k][c], B.xc[k][c], C.xc[k][c], C.xc[k][c]),
                vdot(A.xc[k][c], A.yc[c], B.xc[k][c], B.yc[c], C.xc[k][c], C.yc[c]), sx, sxx, sxy);
          float t = sx*sy[c];
          float num = fmaf(2.f, t, c1)*fmaf(2.f, fmaf(9.f, sxy, -t), c2);
          float sx2 = sx*sx;
          float den = (sx2 + cy1[c])*(fmaf(9.f, sxx, -sx2) + cy2[c]);
          float val = fmaf(-0.5f, num*__builtin_amdgcn_rcpf(den), 0.5f);
          es += fminf(fmaxf(val, 0.f), 1.f);
        }
      }
      float e = SSIM ? fmaf(kWSsim/3.f, es, ((1.f - kWSsim)/3.f)*el) : el*(1.f/3.f);
      if (k == 0) { best = e; acc = e; }
      else {
        acc += e;
        if (e < best) { best = e; bsel = a.i0 + k; }
      }
    }
    if (interior) {
      const unsigned idx = out_base + (unsigned)v*(unsigned)w + (unsigned)u;
      if (!SINGLE && !a.first_pass) {
        float prev = a.err[idx];
        if (use_min) { if (!(best < prev)) { best = prev; bsel = a.sel ? a.sel[idx] : 0; } }
        else acc += prev;
      }
      if (!SINGLE && !a.last_pass) {
        a.err[idx] = use_min ? best : acc;
        if (a.sel) a.sel[idx] = (uint8_t)bsel;
      } else {
        float e = use_min ? best : acc/(float)a.n;
        if (!use_min) bsel = 0;
        if (automask) {
          float est = a.e_static[(unsigned)bi*hw + (unsigned)v*(unsigned)w + (unsigned)u];
          // the tie-break noise is eps*N(0,1): it can only matter when the two errors are within a few eps of each other
          if (a.noise) est = fmaf(kEps32, a.noise[idx], est);
          else if (fabsf(est - e) < 1e-5f) est = fmaf(kEps32, gauss_noise(a.seed_lo, a.seed_hi, (uint32_t)idx), est);
          if (est < e) { e = est; bsel = SMD_SEL_MASKED; }
        }
        a.err[idx] = e;
        if (a.sel) a.sel[idx] = (uint8_t)bsel;
        lsum += e;
      }
    }
  }

  // The row loop.  Ring: A = row j-2, B = row j-1, C = row j.  Software pipeline per step j:
  //   finish(C <- loads of row j, issued during step j-1) | issue(loads of row j+1; needs depth(j+1), loaded during
  //   step j-1) | load depth(j+2) | emit(row j-1)  — so a row's gathers are in flight under the previous row's SSIM math.
  // (Unrolling by 3 to avoid shifting the ring — 54 v_mov per row — was measured slower twice: the three role assignments keep
  //  more values live; 128 VGPRs / 4 waves: 137 us, capped at 96 VGPRs it spills: 483 us, against 87 us for the shifting loop.)
  __device__ __forceinline__ void run() {
    RowState<NI> A = {}, B = {}, C = {};
    Pending<NI, WARP> P = {};
    const int jstart = max(r0 - 1, 0);
    const int jlast = min(r1, h - 1);  // last row that is read (row r1 is halo when r1 < h)
    float Dn = 0.f;
    if (WARP) Dn = ld1(depth_sb, (unsigned)jstart*(unsigned)w + (unsigned)uc);
    issue_row(P, jstart, Dn);
    if (WARP && jstart + 1 <= jlast) Dn = ld1(depth_sb, (unsigned)(jstart + 1)*(unsigned)w + (unsigned)uc);
    for (int j = jstart; j <= r1; ++j) {  // j == r1 == h is the virtual row below the image: it only emits row h-1
      if (j <= jlast) finish_row(C, P, j);
      __builtin_amdgcn_sched_barrier(0);
      if (j + 1 <= jlast) {
        issue_row(P, j + 1, Dn);
        if (WARP && j + 2 <= jlast) Dn = ld1(depth_sb, (unsigned)(j + 2)*(unsigned)w + (unsigned)uc);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int v = j - 1;
      if (v >= r0 && v < r1) {
        if (v == 0) A = C;        // ReflectionPad2d(1): row -1 is row 1
        if (j == h) C = A;        //                     row h is row h-2
        emit_row(A, B, C, v);
      }
      __builtin_amdgcn_sched_barrier(0);
      A = B; B = C;
    }
  }
};

template <int NI, bool WARP, bool SSIM, bool SINGLE>
__device__ __forceinline__ void recon_fwd_body(const ReconFwdArgs& a) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int strip, bi_, s_;
  decode_wave(blockIdx.x, wid, a.nsx*a.nsy, a.b, a.S, strip, bi_, s_);
  if (strip >= a.nsx*a.nsy) return;
  const int sxi = strip % a.nsx, syi = strip/a.nsx;

  FwdCtx<NI, WARP, SSIM, SINGLE> cx{a};
  cx.lane = lane; cx.bi = bi_; cx.s = s_; cx.h = a.h; cx.w = a.w;
  const int c0 = sxi*kFwdCols;
  cx.r0 = syi*a.rh; cx.r1 = min(cx.r0 + a.rh, a.h);
  cx.u = c0 - 1 + lane;
  // The pixel column this lane synthesises: its own, or — for the halo lane just outside the image — the reflected one.
  cx.uc = (cx.u < 0) ? min(-cx.u, a.w - 1) : ((cx.u >= a.w) ? max(2*(a.w - 1) - cx.u, 0) : cx.u);
  cx.interior = (lane >= 1) && (lane <= kFwdCols) && (cx.u < a.w);
  cx.wl = cx.wr = 1.f;
  cx.uf = (float)cx.uc;
  cx.use_min = a.flags & SMD_USE_MIN;
  cx.automask = a.flags & SMD_USE_AUTOMASK;
  cx.hw = (unsigned)a.h*(unsigned)a.w;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int i = a.i0 + k;
    if (WARP) make_cam(cx.cam[k], a.T + ((size_t)i*a.b + cx.bi)*16, a.K + (size_t)cx.bi*16, a.Kinv + (size_t)cx.bi*16);
    cx.splane[k] = a.supp_pk + ((size_t)i*a.b + cx.bi)*4*cx.hw;
  }
  cx.tgt_b = a.tgt + (size_t)cx.bi*3*cx.hw;
  cx.depth_sb = WARP ? a.depth + ((size_t)cx.s*a.b + cx.bi)*cx.hw : nullptr;
  cx.out_base = ((unsigned)cx.s*(unsigned)a.b + (unsigned)cx.bi)*cx.hw;
  cx.lsum = 0.f;

  cx.run();

  if ((SINGLE || a.last_pass) && a.partial != nullptr) {
    float tot = wave_sum(cx.lsum);
    if (lane == 0) a.partial[((size_t)cx.s*a.b + cx.bi)*(a.nsx*a.nsy) + strip] = tot;
  }
}

template <int NI, bool WARP, bool SSIM, bool SINGLE>
__global__ __launch_bounds__(64*kWavesPerBlock) void k_recon_fwd(const ReconFwdArgs a) { recon_fwd_body<NI, WARP, SSIM, SINGLE>(a); }

// Same body for the headline configuration with the register budget capped at 128 (4 waves per SIMD).
__global__ __launch_bounds__(64*kWavesPerBlock, 5) void k_recon_fwd_w4(const ReconFwdArgs a) { recon_fwd_body<2, true, true, true>(a); }

hipError_t launch_recon_fwd(const ReconFwdArgs& a, int ni, bool warp, hipStream_t st) {
  dim3 grid(recon_grid_blocks(a.nsx*a.nsy, a.b, a.S)), block(64*kWavesPerBlock);
  const bool ssim = !(a.flags & SMD_LOSS_L1);
  const bool single = a.first_pass && a.last_pass;
#define SMD_LAUNCH(NI_, WARP_, SSIM_, SINGLE_) hipLaunchKernelGGL((k_recon_fwd<NI_, WARP_, SSIM_, SINGLE_>), grid, block, 0, st, a)
#define SMD_PICK(NI_, WARP_)                                                    \
  do {                                                                          \
    if (ssim) { if (single) SMD_LAUNCH(NI_, WARP_, true, true); else SMD_LAUNCH(NI_, WARP_, true, false); } \
    else { SMD_LAUNCH(NI_, WARP_, false, false); }                              \
  } while (0)
  if (warp && ni == 2 && ssim && single && a.variant == 1) hipLaunchKernelGGL(k_recon_fwd_w4, grid, block, 0, st, a);
  else if (warp) { if (ni == 1) SMD_PICK(1, true); else SMD_PICK(2, true); }
  else { if (ni == 1) SMD_PICK(1, false); else SMD_PICK(2, false); }
#undef SMD_PICK
#undef SMD_LAUNCH
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Planar (n,b,3,h,w) support frames -> RGBX texels (n,b,h,w,4).  One aligned 16-byte load per bilinear tap instead of
// three unaligned 8-byte ones: the texture-data path of a CU returns aligned wide accesses at full rate, while the
// planar gather ran it at a quarter of that (profiles/, DESIGN.md "Kernels").  Used by forward and backward.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_supports(const float* __restrict__ supp, f4* __restrict__ out, unsigned hw, unsigned total) {
  for (unsigned p = blockIdx.x*256u + threadIdx.x; p < total; p += gridDim.x*256u) {
    const unsigned img = p/hw, px = p - img*hw;
    const float* src = supp + (size_t)img*3*hw + px;
    f4 t; t.x = src[0]; t.y = src[hw]; t.z = src[2*(size_t)hw]; t.w = 0.f;
    out[p] = t;
  }
}

hipError_t launch_pack_supports(const float* supp, float* supp_pk, int nb, int h, int w, hipStream_t st) {
  const unsigned hw = (unsigned)h*(unsigned)w, total = (unsigned)nb*hw;
  hipLaunchKernelGGL(k_pack_supports, dim3(min(ceil_div((int)total, 256), 8192)), dim3(256), 0, st, supp, (f4*)supp_pk, hw, total);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Deterministic second stage of every scalar reduction: one block, fp64 accumulation.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sum_partials(const float* __restrict__ partial, int count, double scale, float* out) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < count; i += 256) acc += (double)partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int sft = 128; sft > 0; sft >>= 1) {
    if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0]*scale);
}

hipError_t launch_sum_partials(const float* partial, int count, double scale, float* out, hipStream_t st) {
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, st, partial, count, scale, out);
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
hipError_t launch_stream_copy(const void* src, void* dst, size_t nbytes, int mode, hipStream_t st) {
  hipLaunchKernelGGL(k_stream_copy, dim3(256*8), dim3(256), 0, st, (const f4*)src, (f4*)dst, nbytes/16, mode);
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
