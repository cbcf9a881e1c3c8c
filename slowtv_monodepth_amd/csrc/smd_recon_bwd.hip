// smd_recon_bwd.hip — hand-written adjoint of the fused view-synthesis photometric loss (gfx950), round-2 structure.
//
// Same wave-strip streaming as the forward (smd_recon_fwd.hip): 60 interior columns + 2 halo lanes per side, rows
// r0-2 .. r1+1, one support per pass of the row loop (the state of one support is ~90 registers; two at once would halve the
// occupancy of a kernel whose gathers need the waves to hide their latency), three stages per row step j:
//   stage A (row j)   : re-synthesise the warped pixel x and its bilinear partials dx/dsx, dx/dsy from the four RGB taps
//                       whose loads were issued one step earlier; issue the loads of row j+1
//   stage B (row j-1) : window sums by the forward's sliding scheme (P = r(j-2) + r(j-1); reflection by data in the halo
//                       lanes, by a multiplier / a recovered row at the image's top / bottom), the target-side sums read
//                       back from the packed buffer the forward filled; SSIM partials d e/d(Sx, Sxx, Sxy) times the upstream
//                       gradient routed by `sel` (min-reprojection / automask); box-summed horizontally with the ADJOINT
//                       reflection weights (avg_pool2d + reflection_pad2d backward) and accumulated vertically
//   stage C (row j-2) : dL/dx -> dL/d(sx, sy) (zero where the border clamp is active) -> projective chain rule ->
//                       dL/d depth (read-modify-written across supports) and nine per-lane sums that become dL/d(H, a);
//                       a tiny epilogue kernel turns them into dL/dT, dL/dK and dL/dK^-1.
// Nothing of the forward is stored except the packed texels / target sums (which the forward needs itself) and `sel`.
#include "smd_common.h"
#include "smd_kernels.h"

#ifndef SMD_ABLATE_BWD
#define SMD_ABLATE_BWD 0   // diagnosis builds only (scripts/dev/ablate_bwd.sh): bit 0 no tap gathers, bit 1 no row loads, bit 2 no g_depth traffic, bit 3 no LDS
#endif

namespace smd {

// Three independent reflection-weighted horizontal 3-tap sums in one asm block: r = q + wl*left(q) + wr*right(q) as
// v_mul_dpp + v_fmac_dpp (the shift rides on the multiply's first operand) + a plain add; pure outputs, so the results land
// directly in the caller's registers.  One hazard nop covers all of them.
__device__ __forceinline__ void hsum_w3(float a, float b, float c, float wl, float wr, float& ra, float& rb, float& rc) {
#ifdef SMD_NO_DPP
  ra = hsum3(a, wl, wr); rb = hsum3(b, wl, wr); rc = hsum3(c, wl, wr);
#else
  float ta, tb, tc;
  asm volatile("s_nop 1\n\t"
               "v_mul_f32_dpp %0, %3, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "v_mul_f32_dpp %1, %4, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "v_mul_f32_dpp %2, %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "v_fmac_f32_dpp %0, %3, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "v_fmac_f32_dpp %1, %4, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "v_fmac_f32_dpp %2, %5, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
               : "=&v"(ta), "=&v"(tb), "=&v"(tc)
               : "v"(a), "v"(b), "v"(c), "v"(wl), "v"(wr));
  ra = ta + a; rb = tb + b; rc = tc + c;
#endif
}

// Select between two wave-uniform floats on the scalar unit (a float ?: would be a v_cndmask per use).
__device__ __forceinline__ float usel(bool c, float x, float y) {
  return __builtin_bit_cast(float, c ? __builtin_bit_cast(unsigned, x) : __builtin_bit_cast(unsigned, y));
}

// SKIP: 0 = every row does the full adjoint; 2 = rows where no pixel of the wave selected the current support skip the SSIM
// partials, and rows whose three coefficient rows and L1 term are all dead skip the chain rule.  Coherent selection /
// automask regions (any partly trained network) make 2 the fastest; on noise-like masks the branches cost a few percent.
//
// Register discipline: everything with a lifetime of more than one row step lives in a slot indexed by (row mod 3) — raw
// rows X/Y, the h-summed coefficient rows HC, `sel` — and the row loop is unrolled by three with the phase as a template
// parameter, so nothing is ever moved between "current / previous / one before" registers (the rolled version spent 73 of
// its 421 vector instructions per row on v_mov).  The bilinear partials of a row wait two steps in LDS (own column only, no
// synchronisation), the depth of the row stage C handles is simply loaded a second time (an L1/L2 hit).
constexpr int kHist = 7;   // values per lane and row slot of the LDS history: dx/dsx[3], dx/dsy[3], depth

template <bool SSIM, int SKIP, bool ACC>
struct BwdCtx {
  const ReconBwdArgs& a;
  // wave-uniform
  int h, w, r0, r1, pb0, pb1, sup;
  bool use_min, last, has_gin;
  unsigned w4, rowbytes, so_tex, so_y, so_ta, so_tb;
  float xmax, ymax, wpf;
  Cam2 cm;
  rsrc_t rs_pk, rs_depth, rs_sel, rs_gd, rs_gin;
  // per-lane constants
  unsigned lane4, lane1;
  bool interior;
  float wla, wra, hx0, hy0, hz0;
  // upstream gradient x term weight for the columns of the image (0 for halo lanes outside it), as the values a pixel gets
  // when its `sel` equals / differs from `sel_key` (min-reprojection: the support index; mean: the "masked" code)
  float gs_eq, gs_ne, gl_eq, gl_ne;
  unsigned sel_key;
  float* hist;                 // this lane's column of the wave's LDS history: 3 row slots x {gx, gy} x 3 channels
  float* gacc;                 // ACC: this lane's column of the wave's dL/d depth accumulator, one slot per strip row (LDS)
  // state
  float X[3][3], Y[3][3];      // [row mod 3][channel]: re-synthesised warped pixel / target pixel
  float Px[3], Pxx[3], Pxy[3]; // sliding vertical sums: rows j-1 + j-2 once row j is in
  float HC[3][3][3];           // [row mod 3][channel][{A, B, C}]: h-summed partials d/d(Sx, Sxx (x2), Sxy) of a centre row
  unsigned SEL[3];
  unsigned live_hist;          // bit k: the k-th most recent centre row produced coefficients (wave-uniform)
  f3 t0, t1, t2, t3, py;       // loads in flight for the next row
  float pfx, pfy;
  float Dn;                    // depth of row j+1 (for the next issue); it then waits in LDS for stage C three steps later
#if (SMD_ABLATE_BWD & 8)
  float GH[3][6];
#endif
  float ps[9];                 // per-lane sums of {dnx, dnx*v, dny, dny*v, dz, dz*v, gnx, gny, gz}

  // Coordinates + gathers + target pixel of row jr.  Instruction for instruction the forward's `issue`: same taps, same weights.
  __device__ __forceinline__ void issue(int jr, float D) {
    const float vf = (float)jr;
    const float hx = fmaf(cm.H1, vf, hx0), hy = fmaf(cm.H4, vf, hy0), hz = fmaf(cm.H7, vf, hz0);
    const float nx = fmaf(D, hx, cm.a0), ny = fmaf(D, hy, cm.a1), yz = fmaf(D, hz, cm.tz);
    const float rz = __builtin_amdgcn_rcpf(fmaxf(yz, kZMin));
    const float sx = fmaf(nx, rz, -0.5f), sy = fmaf(ny, rz, -0.5f);
    const float cx = __builtin_amdgcn_fmed3f(sx, 0.f, xmax), cy = __builtin_amdgcn_fmed3f(sy, 0.f, ymax);
    const float x0 = floorf(cx), y0 = floorf(cy);
    pfx = cx - x0; pfy = cy - y0;
    const unsigned o = __umul24((unsigned)fmaf(y0, wpf, x0), 12u);
#if (SMD_ABLATE_BWD & 1)
    const float fo = __builtin_bit_cast(float, (o & 0x7fffffu) | 0x3f000000u);
    t0 = f3{fo, fo*0.5f, fo*0.25f}; t1 = f3{fo*0.3f, fo, fo}; t2 = f3{fo, fo*0.7f, fo}; t3 = f3{fo*0.9f, fo, fo*0.1f};
#else
    t0 = bld3(rs_pk, o, so_tex); t1 = bld3(rs_pk, o + 12u, so_tex);
    t2 = bld3(rs_pk, o, so_tex + rowbytes); t3 = bld3(rs_pk, o + 12u, so_tex + rowbytes);
#endif
#if (SMD_ABLATE_BWD & 2)
    py = f3{pfx, pfy, 0.5f};
#else
    py = bld3(rs_pk, lane4*3u, so_y + (unsigned)jr*w4*3u);
#endif
  }

  __device__ __forceinline__ int reflect_row(int r) const { return (r > h - 1) ? max(2*(h - 1) - r, 0) : r; }

  __device__ __forceinline__ void begin(int jstart) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      SEL[r] = SMD_SEL_MASKED;
#pragma unroll
      for (int c = 0; c < 3; ++c) { X[r][c] = 0.f; Y[r][c] = 0.f; HC[r][c][0] = 0.f; HC[r][c][1] = 0.f; HC[r][c][2] = 0.f; }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { Px[c] = 0.f; Pxx[c] = 0.f; Pxy[c] = 0.f; }
#pragma unroll
    for (int k = 0; k < 9; ++k) ps[k] = 0.f;
    live_hist = 0;
    const float Dfirst = bld(rs_depth, lane4, (unsigned)jstart*w4);
    issue(jstart, Dfirst);
    hist[(0*kHist + 6)*64] = Dfirst;               // row jstart lives in slot 0 (read by stage C only when the strip starts at row 0)
    Dn = bld(rs_depth, lane4, (unsigned)reflect_row(jstart + 1)*w4);
  }

  // One row step: stage A on row j (slot PH), stage B on centre row p = j-1, stage C on row q = j-2.
  template <int PH>
  __device__ __forceinline__ void step(int j_) {
    const int j = __builtin_amdgcn_readfirstlane(j_);   // pin the row counter to an SGPR (row offsets are scalar operands of the buffer accesses)
    constexpr int SN = PH, SP = (PH + 2) % 3, SQ = (PH + 1) % 3;
    constexpr float c1 = 81.f*kC1;     // window sums stay un-normalised (x9), see smd_recon_fwd.hip
    const int p = j - 1, q = j - 2;
    const bool doB = SSIM && p >= pb0 && p <= pb1;
    const bool doC = q >= r0 && q < r1;
    f4 ta; f3 tb;                                   // read only where doB holds
#if (SMD_ABLATE_BWD & 2)
    SEL[SP] = (lane1 + (unsigned)p) & 1u;
    if (doB) { ta = f4{pfx + 1.f, pfy + 1.f, 1.5f, 0.3f}; tb = f3{0.2f, 0.4f, 0.1f}; }
#else
    SEL[SP] = bld8(rs_sel, lane1, (unsigned)p*(unsigned)w);   // rows outside the image read 0 and are never used
    if (doB) {
      ta = bld4(rs_pk, lane4*4u, so_ta + (unsigned)p*w4*4u);
      const f2 tb2 = bld2(rs_pk, lane4*4u, so_tb + (unsigned)p*w4*4u);   // {c1, c2}; the third entry is the forward's static error
      tb = f3{tb2.x, tb2.y, 0.f};
    }
#endif
    // ================= stage A: row j — the warped pixel and its bilinear partials from the taps issued one step earlier
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float dn = t1[c] - t0[c], ds = t3[c] - t2[c];
      const float top = fmaf(pfx, dn, t0[c]), bot = fmaf(pfx, ds, t2[c]);
      const float ddy = bot - top;
      X[SN][c] = fmaf(pfy, ddy, top);
      Y[SN][c] = py[c];
#if (SMD_ABLATE_BWD & 8)
      GH[SN][c] = fmaf(pfy, ds - dn, dn); GH[SN][3 + c] = ddy;
#else
      hist[(SN*kHist + c)*64] = fmaf(pfy, ds - dn, dn);   // dx/dsx; the border-clamp mask is applied in stage C
      hist[(SN*kHist + 3 + c)*64] = ddy;                  // dx/dsy
#endif
    }
    // Next row's loads, unconditionally (also after the last row, where nothing consumes them): a conditional issue would turn
    // every register of the in-flight loads into a loop phi with a second copy.  Below the image the next row is the
    // reflected one (ReflectionPad2d(1): row h is row h-2), re-synthesised like any other row: no special case in vector code.
    issue(reflect_row(j + 1), Dn);
    const float Dkeep = Dn;                          // depth of row j+1: parked in LDS at the end of the step (slot SQ is read first)
#if (SMD_ABLATE_BWD & 2)
    Dn = 1.f + pfx;
#else
    Dn = bld(rs_depth, lane4, (unsigned)reflect_row(j + 2)*w4);
#endif

    // ================= stage B: centre row p — SSIM partials, h-summed with the adjoint reflection weights
    if (SSIM) {
      const float m = usel(p == 0, 2.f, 1.f);       // row -1 is row 1
      float Vx[3], Vxx[3], Vxy[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float xn = X[SN][c], xo = X[SP][c];
        const float xx = xn*xn, xy = xn*Y[SN][c];
        Vx[c] = fmaf(m, xn, Px[c]); Vxx[c] = fmaf(m, xx, Pxx[c]); Vxy[c] = fmaf(m, xy, Pxy[c]);
        Px[c] = xo + xn; Pxx[c] = fmaf(xo, xo, xx); Pxy[c] = fmaf(xo, Y[SP][c], xy);
      }
      bool row_live = false;
      float g2 = 0.f;
      if (doB) {
        g2 = (SEL[SP] == sel_key) ? gs_eq : gs_ne;
        // Rows in which no pixel of this wave selected the current support carry no gradient through their windows:
        // skip the SSIM partials (wave-uniform branch; coherent regions of the min-reprojection / automask are common).
        row_live = SKIP >= 1 ? (__builtin_amdgcn_ballot_w64(g2 != 0.f) != 0) : true;
      }
      live_hist = (live_hist << 1) | (row_live ? 1u : 0u);
      if (row_live) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float sx, sxx, sxy;
          hsum3(Vx[c], Vxx[c], Vxy[c], sx, sxx, sxy);
          const float sy = ta[c], cy2 = (c == 0) ? ta.w : ((c == 1) ? tb.x : tb.y), cy1 = fmaf(sy, sy, c1);
          // e = (1 - val)/2, val = a1*a2/(b1*b2) on the x9 sums (numerator and denominator both scaled by 81*81)
          const float t = sx*sy;
          const float a1 = fmaf(2.f, t, c1), a2 = fmaf(18.f, sxy, fmaf(-2.f, t, 81.f*kC2));
          const float sx2 = sx*sx;
          const float b1 = sx2 + cy1, b2 = fmaf(9.f, sxx, cy2 - sx2);
          const float rden = __builtin_amdgcn_rcpf(b1*b2);
          const float val = a1*a2*rden;
          // d e/d val = -1/2 inside the clamp(0, 1) of e, i.e. for |val| <= 1 (inclusive, like clamp's backward); the partials
          // below all carry a factor 2, so the product of the two is -g
          const float prd = (fabsf(val) <= 1.f) ? -g2*rden : 0.f;
          //   da1/dSx = 2 Sy, da2/dSx = -2 Sy, db1/dSx = 2 Sx, db2/dSx = -2 Sx, da2/dSxy = 18, db2/dSxx = 9
          const float dSx = prd*fmaf(sy, a2 - a1, -sx*(val*(b2 - b1)));
          const float p9 = 9.f*prd;
          const float dSxx2 = -(p9*val)*b1;            // 2 dL/dSxx: stage C multiplies by x, not by 2x
          const float dSxy = p9*a1;
          hsum_w3(dSx, dSxx2, dSxy, wla, wra, HC[SP][c][0], HC[SP][c][1], HC[SP][c][2]);
        }
      } else if (SKIP >= 1) {
        // (Without row skipping nothing needs clearing: every coefficient row that a stage C reads with a non-zero weight
        // was computed, the others are stale-but-finite values times a zero weight.)
        asm volatile("" ::: "memory");               // keep the clears inside the rarely taken branch
#pragma unroll
        for (int c = 0; c < 3; ++c) { HC[SP][c][0] = 0.f; HC[SP][c][1] = 0.f; HC[SP][c][2] = 0.f; }
      }
    }

    // ================= stage C: row q — dL/dx -> dL/d(sx, sy) -> depth, pose sums
    if (doC) {
      // weights of coefficient rows q-1 / q+1 in the gradient of row q (reflect_weights_adj, on the scalar unit)
      float lo_q = usel(q == 0, 0.f, usel(q == 1, 2.f, 1.f)), hi_q = usel(q == h - 1, 0.f, usel(q == h - 2, 2.f, 1.f));
      if (h == 2) { lo_q = usel(q == 1, 2.f, 0.f); hi_q = usel(q == 0, 2.f, 0.f); }
      const float gl = (SEL[SQ] == sel_key) ? gl_eq : gl_ne;
      // coefficient rows q-1, q, q+1 all skipped and no L1 term anywhere in the wave: the gradient of this row is zero
      const bool dead = SKIP >= 2 ? ((live_hist & 7u) == 0u && __builtin_amdgcn_ballot_w64(gl != 0.f) == 0) : false;
      const unsigned qro = (unsigned)q*w4;
      const float D2 = hist[(SQ*kHist + 6)*64];
      float gD = 0.f;
      if (!dead) {
        float gpx = 0.f, gpy = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#if (SMD_ABLATE_BWD & 8)
          const float gxq = GH[SQ][c], gyq = GH[SQ][3 + c];
#else
          const float gxq = hist[(SQ*kHist + c)*64], gyq = hist[(SQ*kHist + 3 + c)*64];
#endif
          const float xq = X[SQ][c], yq = Y[SQ][c];
          const float d = xq - yq;
          float gxc = (d != 0.f) ? __builtin_copysignf(gl, d) : 0.f;     // gl*sign(d)
          if (SSIM) {
            const float SA = fmaf(hi_q, HC[SP][c][0], fmaf(lo_q, HC[SN][c][0], HC[SQ][c][0]));
            const float SB = fmaf(hi_q, HC[SP][c][1], fmaf(lo_q, HC[SN][c][1], HC[SQ][c][1]));
            const float SC = fmaf(hi_q, HC[SP][c][2], fmaf(lo_q, HC[SN][c][2], HC[SQ][c][2]));
            gxc += fmaf(xq, SB, fmaf(yq, SC, SA));   // d/dx_q of the x9 sums: 1, 2 x_q (the 2 is in SB), y_q
          }
          gpx = fmaf(gxc, gxq, gpx);
          gpy = fmaf(gxc, gyq, gpy);
        }
        // projective chain rule at (q, u): the cheap geometry is recomputed from the depth
        const float vf = (float)q;
        const float hx = fmaf(cm.H1, vf, hx0), hy = fmaf(cm.H4, vf, hy0), hz = fmaf(cm.H7, vf, hz0);
        const float nx = fmaf(D2, hx, cm.a0), ny = fmaf(D2, hy, cm.a1), yz = fmaf(D2, hz, cm.tz);
        const float rz = __builtin_amdgcn_rcpf(fmaxf(yz, kZMin));
        const float sx = fmaf(nx, rz, -0.5f), sy = fmaf(ny, rz, -0.5f);
        // d(clamped coordinate)/d(unclamped) is 0 on and outside the border (grid_sample's border padding); halo lanes
        // duplicate a neighbour wave's column and contribute nothing
        const float gnx = (interior && sx > 0.f && sx < xmax) ? gpx*rz : 0.f;
        const float gny = (interior && sy > 0.f && sy < ymax) ? gpy*rz : 0.f;
        const float gz = (yz >= kZMin) ? -fmaf(gnx, nx, gny*ny)*rz : 0.f;
        gD = fmaf(gnx, hx, fmaf(gny, hy, gz*hz));
        const float dnx = gnx*D2, dny = gny*D2, dz = gz*D2;
        ps[0] += dnx; ps[1] = fmaf(dnx, vf, ps[1]); ps[2] += dny; ps[3] = fmaf(dny, vf, ps[3]); ps[4] += dz; ps[5] = fmaf(dz, vf, ps[5]);
        ps[6] += gnx; ps[7] += gny; ps[8] += gz;
      }
      // dL/d depth: accumulated across the support passes through g_depth; the last pass adds what reaches depth from other
      // consumers.
#if (SMD_ABLATE_BWD & 4)
      if (gD == 12345.678f) bst(rs_gd, lane4, qro, gD);
      else
#endif
      if (ACC) {
        // The sum over the support passes stays in LDS (strips of <= kAccRows rows): no read-modify-write of g_depth, one global
        // store per pixel by the last pass.  Each lane touches only its own column, so no synchronisation is needed.
        float* slot = gacc + (q - r0)*64;
        if (last && has_gin) gD += bld(rs_gin, lane4, qro);
        if (a.k0_scale != 0.f) gD *= (D2 < 1.f/kEps32) ? -D2*D2*a.k0_scale : 0.f;
        if (sup != 0) gD += *slot;
        if (last) { if (interior) bst(rs_gd, lane4, qro, gD); }
        else *slot = gD;
      } else if (interior && (!dead || sup == 0 || (last && has_gin))) {
        if (last && has_gin) gD += bld(rs_gin, lane4, qro);
        // K0 fused: d depth / d(up-sampled, scaled disparity) applied here, where the depth is at hand (linear, so per pass)
        if (a.k0_scale != 0.f) gD *= (D2 < 1.f/kEps32) ? -D2*D2*a.k0_scale : 0.f;
        if (sup != 0) gD += bld(rs_gd, lane4, qro);
        bst(rs_gd, lane4, qro, gD);
      }
    }
    hist[(SQ*kHist + 6)*64] = Dkeep;               // row j+1's slot: stage C reads it at step j+3 (no second load of the depth)
  }

  __device__ __forceinline__ void run(int jstart) {
    begin(jstart);
    const int jend = r1 + 1;
    for (int j = jstart;;) {
      step<0>(j); if (++j > jend) break;
      step<1>(j); if (++j > jend) break;
      step<2>(j); if (++j > jend) break;
    }
  }
};

// Four waves per SIMD (<= 128 VGPRs): without the cap the allocator settles at 133 and the kernel loses a wave of occupancy,
// 138 -> 127 us at cfg 2 (the gathers' latency is what the extra wave hides).
constexpr int kAccRows = 16;   // tallest strip whose dL/d depth rows are accumulated across the support passes in LDS

template <bool SSIM, int SKIP, bool ACC>
__global__ __launch_bounds__(64*kWavesPerBlock, 4) void k_recon_bwd(const ReconBwdArgs a) {
  // per wave: 3 row slots x ({gx, gy} x 3 channels + depth) x 64 lanes [+ ACC: kAccRows x 64 lanes of dL/d depth]; ONE array (a second
  // __shared__ object makes the compiler serialise LDS and vector-memory waits)
  __shared__ float hist_lds[kWavesPerBlock*(3*kHist + (ACC ? kAccRows : 0))*64];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // segment of the (possibly tapered) partition this block belongs to (smd_kernels.h: ReconMainArgs::b1)
  const unsigned nblk1 = recon_grid_blocks(a.nsx*a.nsy, a.b1, a.S);
  const bool tail = blockIdx.x >= nblk1;
  const int nstrips = a.nsx*(tail ? a.nsy2 : a.nsy), seg_rh = tail ? a.rh2 : a.rh;
  int strip, bi, s;
  decode_wave(tail ? blockIdx.x - nblk1 : blockIdx.x, wid, nstrips, tail ? a.b - a.b1 : a.b1, a.S, strip, bi, s);
  if (strip >= nstrips) return;
  if (tail) bi += a.b1;
  const int sxi = strip % a.nsx, syi = strip/a.nsx;
  const int h = a.h, w = a.w;

  BwdCtx<SSIM, SKIP, ACC> cx{a};
  cx.hist = hist_lds + wid*((3*kHist + (ACC ? kAccRows : 0))*64) + lane;
  cx.gacc = cx.hist + 3*kHist*64;
  cx.h = h; cx.w = w;
  cx.r0 = syi*seg_rh; cx.r1 = min(cx.r0 + seg_rh, h);
  const int u = sxi*kBwdCols - 2 + lane;
  const bool col_ok = (u >= 0) && (u < w);
  // data column: the lane's own, or the reflected one for the halo lanes outside the image (reflection by data)
  const int uc = (u < 0) ? min(-u, w - 1) : ((u >= w) ? max(2*(w - 1) - u, 0) : u);
  cx.interior = (lane >= 2) && (lane < 2 + kBwdCols) && (u < w);
  cx.lane4 = (unsigned)uc*4u; cx.lane1 = (unsigned)uc;
  reflect_weights_adj(min(max(u, 0), w - 1), w, cx.wla, cx.wra);   // how much column u receives from u-1 / u+1
  if (!col_ok) { cx.wla = 0.f; cx.wra = 0.f; }
  const float uf = (float)uc;

  cx.use_min = a.flags & SMD_USE_MIN;
  const size_t hw = (size_t)h*w;
  const unsigned hw4 = (unsigned)hw*4u;
  cx.w4 = (unsigned)w*4u;
  float gscale = a.g_loss[0]/((float)a.S*(float)a.b*(float)h*(float)w);
  if (!cx.use_min) gscale /= (float)a.n;
  const float gm_ssim = col_ok ? gscale*(SSIM ? kWSsim/3.f : 0.f) : 0.f;
  const float gm_l1 = col_ok ? gscale*(SSIM ? (1.f - kWSsim)/3.f : 1.f/3.f) : 0.f;
  cx.gs_eq = cx.use_min ? gm_ssim : 0.f; cx.gs_ne = cx.use_min ? 0.f : gm_ssim;
  cx.gl_eq = cx.use_min ? gm_l1 : 0.f; cx.gl_ne = cx.use_min ? 0.f : gm_l1;
  cx.xmax = (float)(w - 1); cx.ymax = (float)(h - 1); cx.wpf = (float)(w + 1);

  const size_t sb = ((size_t)s*a.b + bi)*hw;
  cx.rs_pk = make_rsrc(a.packed, packed_total_floats(a.b, a.n, h, w)*4);
  cx.rs_depth = make_rsrc(a.depth + sb, hw*4);
  cx.rs_sel = make_rsrc(a.sel + sb, hw);
  cx.rs_gd = make_rsrc(a.g_depth + sb, hw*4);
  cx.has_gin = a.g_in != nullptr;
  cx.rs_gin = make_rsrc(cx.has_gin ? a.g_in + sb : nullptr, cx.has_gin ? hw*4 : 0);
  const unsigned texel_bytes = (unsigned)(h + 1)*(unsigned)(w + 1)*12u;
  cx.rowbytes = ((unsigned)w + 1u)*12u;
  cx.so_y = (unsigned)(packed_texel_floats(a.b, a.n, h, w)*4) + (unsigned)bi*hw4*3u;
  cx.so_ta = (unsigned)((packed_texel_floats(a.b, a.n, h, w) + packed_ypix_floats(a.b, h, w))*4) + (unsigned)bi*hw4*4u;
  cx.so_tb = cx.so_ta + (unsigned)(packed_tpix_floats(a.b, h, w)*4);

  const int jstart = max(cx.r0 - 2, 0);
  cx.pb0 = max(cx.r0 - 1, 0); cx.pb1 = min(cx.r1, h - 1); // centre rows whose coefficients are needed

  for (int i = 0; i < a.n; ++i) {
    make_cam2(cx.cm, cx.hx0, cx.hy0, cx.hz0, a.T + ((size_t)i*a.b + bi)*16, a.K + (size_t)bi*16, a.Kinv + (size_t)bi*16,
              a.wscale, a.hscale, uf);
    cx.so_tex = (unsigned)(i*a.b + bi)*texel_bytes;
    cx.sup = i;
    cx.sel_key = cx.use_min ? (unsigned)i : (unsigned)SMD_SEL_MASKED;
    cx.last = (i == a.n - 1);
    cx.run(jstart);

    // per-wave pose partials: d/d(H[0..8], a0, a1, tz) of the UN-scaled homography (rows 0/1 of the folded one carry the grid
    // scale); the column factor of H[.,0] is constant per lane
    float* pp = a.pose_partial + (((size_t)i*a.b + bi)*(size_t)a.pose_stride + (size_t)s*nstrips + strip)*kPoseSums;
    const float* ps = cx.ps;
    const float ws = a.wscale, hs = a.hscale;
    const float psum[kPoseSums] = {ps[0]*uf*ws, ps[1]*ws, ps[0]*ws, ps[2]*uf*hs, ps[3]*hs, ps[2]*hs, ps[4]*uf, ps[5], ps[4],
                                   ps[6]*ws, ps[7]*hs, ps[8]};
#pragma unroll
    for (int k = 0; k < kPoseSums; ++k) {
      const float tot = wave_sum(psum[k]);
      if (lane == 0) pp[k] = tot;
    }
  }
}

hipError_t launch_recon_bwd(const ReconBwdArgs& a, hipStream_t st) {
  dim3 grid(recon_grid_blocks(a.nsx*a.nsy, a.b1, a.S) + (a.b1 < a.b ? recon_grid_blocks(a.nsx*a.nsy2, a.b - a.b1, a.S) : 0u)), block(64*kWavesPerBlock);
  const bool ssim = !(a.flags & SMD_LOSS_L1);
  // more than one support and strips short enough: the per-pixel sum over the support passes is kept in LDS
  const bool acc = a.n >= 2 && a.rh <= kAccRows && (a.b1 == a.b || a.rh2 <= kAccRows);
#define SMD_BWD(SSIM_, SKIP_) do { \
    if (acc) hipLaunchKernelGGL((k_recon_bwd<SSIM_, SKIP_, true>), grid, block, 0, st, a); \
    else hipLaunchKernelGGL((k_recon_bwd<SSIM_, SKIP_, false>), grid, block, 0, st, a); } while (0)
  if (ssim) {
    if (a.skip_level >= 1) SMD_BWD(true, 2); else SMD_BWD(true, 0);
  } else SMD_BWD(false, 0);
#undef SMD_BWD
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Epilogue: sum the per-wave partials of dL/d(H, a0, a1, tz) and push them through
//   H[0:2] = K2 * M,  H[2] = M[2],  M = R * Ki3,  (a0, a1) = K2 * t,  tz = t[2]
// to dL/dT (n,b,4,4), dL/dK (b,4,4), dL/dKinv (b,4,4).  One block of 16 waves per sample: every wave reduces some of the
// n*12 sums (fp64, fixed order -> deterministic, four loads in flight), then one thread per support does the 3x3 algebra and
// thread 0 adds the supports' contributions to dL/dK, dL/dKinv in index order.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_pose_finalize(const float* __restrict__ pose_partial, int entries1, int entries2, int b1, int stride,
                                                        const float* __restrict__ T, const float* __restrict__ K,
                                                        const float* __restrict__ Kinv, float* g_T, float* g_K, float* g_Kinv,
                                                        int b, int n) {
  __shared__ double tot[SMD_MAX_SUPPORTS][kPoseSums];
  __shared__ double gKs[SMD_MAX_SUPPORTS][6], gKis[SMD_MAX_SUPPORTS][9];
  const int bi = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int entries = bi < b1 ? entries1 : entries2;   // tapered partition: the last samples were cut into more strips
  for (int pr = wv; pr < n*kPoseSums; pr += 16) {
    const int i = pr/kPoseSums, k = pr - i*kPoseSums;
    const float* pp = pose_partial + ((size_t)i*b + bi)*(size_t)stride*kPoseSums + k;
    double acc = 0.0;
    int e = lane;
    for (; e + 192 < entries; e += 256) {
      const float v0 = pp[(size_t)e*kPoseSums], v1 = pp[(size_t)(e + 64)*kPoseSums], v2 = pp[(size_t)(e + 128)*kPoseSums], v3 = pp[(size_t)(e + 192)*kPoseSums];
      acc += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
    }
    for (; e < entries; e += 64) acc += (double)pp[(size_t)e*kPoseSums];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) tot[i][k] = acc;
  }
  __syncthreads();
  if ((int)threadIdx.x < n) {
    const int i = threadIdx.x;
    const float* Tm = T + ((size_t)i*b + bi)*16;
    const float* Km = K + (size_t)bi*16;
    const float* Ki = Kinv + (size_t)bi*16;
    double R[9], t[3], K2[6], Ki3[9], M[9];
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) { R[r*3 + c] = Tm[r*4 + c]; Ki3[r*3 + c] = Ki[r*4 + c]; } t[r] = Tm[r*4 + 3]; }
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) K2[r*3 + c] = Km[r*4 + c];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M[r*3 + c] = R[r*3]*Ki3[c] + R[r*3 + 1]*Ki3[3 + c] + R[r*3 + 2]*Ki3[6 + c];
    const double* gH = tot[i];        // 3x3
    const double ga[2] = {tot[i][9], tot[i][10]};
    const double gtz = tot[i][11];
    double gM[9];
    for (int m = 0; m < 3; ++m) for (int c = 0; c < 3; ++c) gM[m*3 + c] = K2[m]*gH[c] + K2[3 + m]*gH[3 + c];
    for (int c = 0; c < 3; ++c) gM[6 + c] += gH[6 + c];
    for (int r = 0; r < 2; ++r) for (int m = 0; m < 3; ++m)
      gKs[i][r*3 + m] = gH[r*3]*M[m*3] + gH[r*3 + 1]*M[m*3 + 1] + gH[r*3 + 2]*M[m*3 + 2] + ga[r]*t[m];
    double gt[3];
    for (int m = 0; m < 3; ++m) gt[m] = K2[m]*ga[0] + K2[3 + m]*ga[1];
    gt[2] += gtz;
    float* gTo = g_T + ((size_t)i*b + bi)*16;
    for (int r = 0; r < 3; ++r) {
      for (int m = 0; m < 3; ++m)
        gTo[r*4 + m] = (float)(gM[r*3]*Ki3[m*3] + gM[r*3 + 1]*Ki3[m*3 + 1] + gM[r*3 + 2]*Ki3[m*3 + 2]);
      gTo[r*4 + 3] = (float)gt[r];
    }
    for (int c = 0; c < 4; ++c) gTo[12 + c] = 0.f;
    for (int m = 0; m < 3; ++m) for (int c = 0; c < 3; ++c)
      gKis[i][m*3 + c] = R[m]*gM[c] + R[3 + m]*gM[3 + c] + R[6 + m]*gM[6 + c];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (g_K) {
      float* o = g_K + (size_t)bi*16;
      for (int k = 0; k < 16; ++k) o[k] = 0.f;
      for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) {
        double acc = 0.0;
        for (int i = 0; i < n; ++i) acc += gKs[i][r*3 + c];
        o[r*4 + c] = (float)acc;
      }
    }
    if (g_Kinv) {
      float* o = g_Kinv + (size_t)bi*16;
      for (int k = 0; k < 16; ++k) o[k] = 0.f;
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        double acc = 0.0;
        for (int i = 0; i < n; ++i) acc += gKis[i][r*3 + c];
        o[r*4 + c] = (float)acc;
      }
    }
  }
}

hipError_t launch_pose_finalize(const float* pose_partial, int entries1, int entries2, int b1, int stride, const float* T, const float* K,
                                const float* Kinv, float* g_T, float* g_K, float* g_Kinv, int b, int n, hipStream_t st) {
  hipLaunchKernelGGL(k_pose_finalize, dim3(b), dim3(1024), 0, st, pose_partial, entries1, entries2, b1, stride, T, K, Kinv, g_T, g_K, g_Kinv, b, n);
  return hipGetLastError();
}

}  // namespace smd
