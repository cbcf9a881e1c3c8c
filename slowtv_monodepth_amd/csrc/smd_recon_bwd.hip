// smd_recon_bwd.hip — hand-written adjoint of the fused view-synthesis photometric loss (gfx950).
//
// Same wave-strip streaming as the forward (smd_recon_fwd.hip): 60 interior columns + 2 halo lanes per side, rows
// r0-2 .. r1+1, one support per wave (the state of one support is ~90 registers; two at once were built in round 4 —
// k_recon_bwd_pair below — and are slower: profiles/r04_pair_backward.txt).  Since round 3 the supports of a strip are handled by
// DIFFERENT waves of one block, concurrently (n = 2: a block is 2 strips x 2 supports): a work unit is half as long (a launch is only
// ~3 generations of waves, so its tail is a fraction of a unit), the waves of a strip pull the same target-side rows (`sel`, target
// pixel, window terms, depth) through one L1, and the per-pixel sum of dL/d depth over the supports is formed once, from LDS, by the
// strip's last wave.  Three stages per row step j:
//   stage A (row j)   : re-synthesise the warped pixel x and its bilinear partials dx/dsx, dx/dsy from the four RGB taps
//                       whose loads were issued one step earlier; issue the loads of row j+1
//   stage B (row j-1) : window sums of the centre row from its three raw rows (reflection by data: in the halo lanes horizontally,
//                       by re-synthesised rows -1 = 1 and h = h-2 vertically), the target-side sums read back from the packed buffer
//                       the forward filled; SSIM partials d e/d(Sx, Sxx, Sxy) times the upstream gradient routed by `sel`
//                       (min-reprojection / automask); box-summed horizontally with the ADJOINT reflection weights
//                       (avg_pool2d + reflection_pad2d backward) and accumulated vertically
//   stage C (row j-2) : dL/dx -> dL/d(sx, sy) (zero where the border clamp is active) -> projective chain rule ->
//                       dL/d depth (this support's share, parked in LDS) and nine per-lane sums that become dL/d(H, a);
//                       the K0-adjoint launch that follows (or the block that finishes a sample last) turns the sums into
//                       dL/dT, dL/dK and dL/dK^-1.
// Nothing of the forward is stored except the packed texels / target sums (which the forward needs itself) and `sel`.
#include "smd_common.h"
#include "smd_kernels.h"
#include "smd_pose_fin.h"

// The row-loop variants (plain / gated, and whatever the experiment switches select) promise the SAME gradients bit for bit: every
// fused multiply-add in this file is written as one (fmaf); left to -ffp-contract=fast the compiler decides per instantiation which
// a*b + c it fuses, and two instantiations of the same source then differ in the last bit.
#pragma clang fp contract(off)

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

// SKIP: 0 = every row does the full adjoint; >= 1 (round 4) = the row loop is GATED by what the selection maps route to this wave's
// support.  Before the row loop a wave reads the `sel` bytes of its centre rows once (in batches whose loads overlap) and keeps
//   selbits  (per lane)      bit i: the pixel (row rb + i, this lane's column) routes gradient to the current support
//   L        (wave-uniform)  bit i: centre row rb + i has at least one such pixel in the wave's 64 columns
//   N = L | L << 1 | L >> 1  bit i: row rb + i enters a live 3x3 window: it has to be re-synthesised, and its gradient is not zero
// A row step then runs stage A only for a row of N (and the loads of the next row are only issued if that one is in N), stage B only
// for a centre row of L, stage C only for a row of N; everything else costs a few scalar instructions.  With every pixel masked the
// kernel is its prologue and the zero fill (round 3's skipping still re-synthesised every row: 92 us at cfg 2).  Same gradients bit for
// bit as SKIP = 0: a skipped row contributes exact zeros there.  The window sums are formed from the three raw rows of a centre
// (no running state to keep consistent across skipped rows).
//
// Register discipline: everything with a lifetime of more than one row step lives in a slot indexed by (row mod 3) — raw
// rows X/Y, the h-summed coefficient rows HC — and the row loop is unrolled by three with the phase as a template
// parameter, so nothing is ever moved between "current / previous / one before" registers (the rolled version spent 73 of
// its 421 vector instructions per row on v_mov).  The bilinear partials of a row and its depth wait two steps in LDS (own column
// only, no synchronisation).
constexpr int kHist = 7;   // values per lane and row slot of the LDS history: dx/dsx[3], dx/dsy[3], depth
constexpr int kSelBatch = 10;   // `sel` rows requested at once by the prologue (8 + 2 rows of a tapered strip in one batch, 16 + 2 in two)

// Experiment switches of the row loop (scripts/dev/bwd_lib_variants.sh builds one library per combination; the defaults are what won):
#ifndef SMD_BWD_PEEL
#define SMD_BWD_PEEL 1       // 1: the pipeline's fill (two A-only steps, two A+B steps) is peeled off and the steady-state body has no range checks
#endif
#ifndef SMD_BWD_ROWSEL
#define SMD_BWD_ROWSEL 1     // the plain loop (SKIP = 0) reads `sel` one row per step, as part of the row's loads, instead of scanning the strip up front
#endif
#ifndef SMD_BWD_STATEFUL
#define SMD_BWD_STATEFUL 0   // the plain loop keeps sliding vertical sums P = r(j-2) + r(j-1) (the forward's scheme) instead of re-adding three raw rows
#endif

// XTRA: the instantiation also serves the two rare cases that cost every row step two scalar branches at the end of its basic block — a
// gradient that reaches the depth from another consumer (g_in) and a wave that takes several supports in turn (n > 4, or SMD_BWD_WPS < n).
template <bool SSIM, int SKIP, bool ACC, bool XTRA>
struct BwdCtx {
  static constexpr bool kScan = (SKIP >= 1) || !SMD_BWD_ROWSEL;        // `sel` of the whole strip scanned before the row loop
  static constexpr bool kSliding = (SKIP == 0) && SMD_BWD_STATEFUL;    // (a gated loop skips rows: no running state)
  static constexpr bool kPeel = SMD_BWD_PEEL;
  const ReconBwdArgs& a;
  // wave-uniform
  int h, w, r0, r1, pb0, pb1;
  int rb;                            // image row of bit 0 of selbits / L / N (r0 - 3: every row a step looks at has a bit index >= 0)
  unsigned L, N;
  bool use_min, add_gin, acc_prev;   // add_gin: this pass adds the gradient that reaches the depth from other consumers; acc_prev: a
                                     // previous pass of this wave (n > 4) left its share in the LDS rows
  unsigned w4, rowbytes, so_tex, so_y, so_ta, so_tb;
  float xmax, ymax, wpf;
  float g_ssim, g_l1;                // upstream gradient x term weight of a pixel that routes gradient to this support
  Cam2 cm;
  rsrc_t rs_pk, rs_depth, rs_sel, rs_gd, rs_gin;
  // per-lane constants
  unsigned lane4, lane1;
  bool interior, col_ok;
  float wla, wra, hx0, hy0, hz0;
  unsigned sel_key;
  unsigned selbits;            // kScan: bit i = the pixel (row rb + i, this lane's column) routes gradient to the current support
  unsigned SELR[3];            // !kScan: `sel` of the rows in flight, [row mod 3]
  float* hist;                 // this lane's column of the wave's LDS history: 3 row slots x {gx, gy} x 3 channels
  float* gacc;                 // ACC: this lane's column of the wave's dL/d depth rows (this wave's supports), one slot per strip row (LDS)
  // state
  float X[3][3], Y[3][3];      // [row mod 3][channel]: re-synthesised warped pixel / target pixel
  float Px[3], Pxx[3], Pxy[3]; // kSliding: vertical sums of rows j-1 + j-2 once row j is in
  float HC[3][3][3];           // [row mod 3][channel][{A, B, C}]: h-summed partials d/d(Sx, Sxx (x2), Sxy) of a centre row
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

  // ReflectionPad2d(1) by data, above and below the image: row -1 is row 1, row h is row h-2 (rows further out only feed the dummy
  // centre rows -1 / h of the peeled form, whose coefficients are exact zeros: any valid row will do)
  // (branch-free on the scalar unit — s_abs / s_sub / s_max: written as nested conditionals the compiler emitted two scalar BRANCHES per call,
  // four per row step, each one splitting the step's basic block)
  __device__ __forceinline__ int reflect_row(int r) const { const int t = abs(r); return max((h - 1) - abs((h - 1) - t), 0); }
  __device__ __forceinline__ bool bit(unsigned m, int row) const { return (m >> (unsigned)(row - rb)) & 1u; }
  __device__ __forceinline__ bool routes(unsigned v) const { return (v == sel_key) == use_min; }   // min-reprojection: sel == support; mean: sel != "masked"

  // kScan: `sel` of the centre rows pb0 .. pb1, once per support pass: which pixels / rows route gradient to the current support.  All of
  // a strip's rows are requested before the first is looked at (two batches of ten: 8 + 2 rows of a tapered strip, 16 + 2 of a full
  // one), and so are the depths of the first two rows of the pipeline: one round trip for the whole prologue.
  __device__ __forceinline__ void sel_rows(int base, unsigned (&v)[kSelBatch]) const {
#pragma unroll
    for (int k = 0; k < kSelBatch; ++k)
#if (SMD_ABLATE_BWD & 2)
      v[k] = (lane1 + (unsigned)(base + k)) & 1u;
#else
      v[k] = bld8(rs_sel, lane1, (unsigned)min(max(base + k, 0), h - 1)*(unsigned)w);
#endif
  }
  __device__ __forceinline__ void sel_bits(int base, const unsigned (&v)[kSelBatch]) {
#pragma unroll
    for (int k = 0; k < kSelBatch; ++k) {
      const int p = base + k;
      if (p <= pb1 && p >= 0 && p < h) {     // (dummy centre rows route nothing)
        const bool on = col_ok && routes(v[k]);
        const unsigned sh = (unsigned)(p - rb);
        selbits |= (on ? 1u : 0u) << sh;
        if (SKIP >= 1) L |= (__builtin_amdgcn_ballot_w64(on) != 0 ? 1u : 0u) << sh;
      }
    }
  }

  __device__ __forceinline__ void begin(int jstart) {
    selbits = 0u; L = 0xffffffffu; N = 0xffffffffu;
    unsigned va[kSelBatch], vb[kSelBatch];
    const bool two = pb1 - pb0 >= kSelBatch;
    if (kScan) {
      L = 0u;
      sel_rows(pb0, va);
      if (two) sel_rows(pb0 + kSelBatch, vb);
    }
    const float Dfirst = bld(rs_depth, lane4, (unsigned)reflect_row(jstart)*w4);          // (speculative with SKIP: two loads per wave)
    Dn = bld(rs_depth, lane4, (unsigned)reflect_row(jstart + 1)*w4);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      SELR[r] = SMD_SEL_MASKED;
#pragma unroll
      for (int c = 0; c < 3; ++c) { X[r][c] = 0.f; Y[r][c] = 0.f; HC[r][c][0] = 0.f; HC[r][c][1] = 0.f; HC[r][c][2] = 0.f; }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { Px[c] = 0.f; Pxx[c] = 0.f; Pxy[c] = 0.f; }
#pragma unroll
    for (int k = 0; k < 9; ++k) ps[k] = 0.f;
    t0 = f3{0.f, 0.f, 0.f}; t1 = t0; t2 = t0; t3 = t0; py = t0; pfx = 0.f; pfy = 0.f;
    // The first row's gathers go out BEFORE the scan is evaluated (its ~100 scalar / vector instructions then run under their latency;
    // a gated wave whose first row turns out dead wastes five loads)
    issue(reflect_row(jstart), Dfirst);
    hist[(0*kHist + 6)*64] = Dfirst;               // row jstart lives in slot 0
    if (kScan) {
      sel_bits(pb0, va);
      if (two) sel_bits(pb0 + kSelBatch, vb);
      for (int base = pb0 + 2*kSelBatch; base <= pb1; base += kSelBatch) { sel_rows(base, va); sel_bits(base, va); }   // strips taller than 18 rows (single support only)
      if (SKIP >= 1) N = L | (L << 1) | (L >> 1); else L = 0xffffffffu;
    }
  }

  // One row step: stage A on row j (slot PH), stage B on centre row p = j-1, stage C on row q = j-2.
  // DOB / DOC: 0 / 1 = the stage is compiled out / in (peeled form: the first two steps of a strip only synthesise, the next two also
  // have a centre row, from the fifth on every step carries all three stages and the body has no range checks); 2 = decided per step
  // from the rows' ranges (the compact form: one body per phase).
  // GATED: the step consults the row masks (false: every stage of the step runs — the plain loop, and the gated loop's fast path through
  // stretches where everything is live).
  template <int PH, int DOB, int DOC, bool GATED>
  __device__ __forceinline__ void step(int j_) {
    const int j = __builtin_amdgcn_readfirstlane(j_);   // pin the row counter to an SGPR (row offsets are scalar operands of the buffer accesses)
    constexpr int SN = PH, SP = (PH + 2) % 3, SQ = (PH + 1) % 3;
    constexpr float c1 = 81.f*kC1;     // window sums stay un-normalised (x9), see smd_recon_fwd.hip
    const int p = j - 1, q = j - 2;
    const bool inB = SSIM && DOB != 0 && (DOB == 1 || (p >= pb0 && p <= pb1));
    const bool inC = DOC != 0 && (DOC == 1 || (q >= r0 && q < r1));
    const bool doB = inB && (!GATED || bit(L, p));
    const bool doC = inC && (!GATED || bit(N, q));
    const bool doA = !GATED || bit(N, j);
    const bool doI = !GATED || bit(N, j + 1);      // row j+1 is needed: request its taps now, park its depth at the end of the step
    const bool doD = !GATED || bit(N, j + 2);      // row j+2 is needed: request its depth now

    f4 ta; f3 tb;                                   // read only where doB holds
#if !(SMD_ABLATE_BWD & 2)
    if (!kScan) SELR[SP] = bld8(rs_sel, lane1, (unsigned)min(max(p, 0), h - 1)*(unsigned)w);
#endif
    if (doB) {
#if (SMD_ABLATE_BWD & 2)
      ta = f4{pfx + 1.f, pfy + 1.f, 1.5f, 0.3f}; tb = f3{0.2f, 0.4f, 0.1f};
#else
      const unsigned pc = (unsigned)min(max(p, 0), h - 1);               // (dummy centre rows read a real row: their weight is an exact zero, their values must be finite)
      ta = bld4(rs_pk, lane4*4u, so_ta + pc*w4*4u);
      const f2 tb2 = bld2(rs_pk, lane4*4u, so_tb + pc*w4*4u);   // {c1, c2}; the third entry is the forward's static error
      tb = f3{tb2.x, tb2.y, 0.f};
#endif
    }
    // ================= stage A: row j — the warped pixel and its bilinear partials from the taps issued one step earlier
    if (doA) {
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
    }
    // Next row's loads.  SKIP = 0: unconditionally (also after the last row, where nothing consumes them).  Rows outside the image are
    // reflected ones (ReflectionPad2d(1): row -1 is row 1, row h is row h-2), re-synthesised like any other row: no special case in vector code.
    if (doI) issue(reflect_row(j + 1), Dn);
    const float Dkeep = Dn;                          // depth of row j+1: parked in LDS at the end of the step (slot SQ is read first)
    if (doD) {
#if (SMD_ABLATE_BWD & 2)
      Dn = 1.f + pfx;
#else
      Dn = bld(rs_depth, lane4, (unsigned)reflect_row(j + 2)*w4);
#endif
    }

    // ================= stage B: centre row p — SSIM partials, h-summed with the adjoint reflection weights
    if (SSIM) {
      // window sums of centre p from its three raw rows (slots SQ = p-1, SP = p, SN = p+1); the image's top and bottom rows get their
      // reflected neighbour by data ("row -1" is a re-synthesised row 1, "row h" a re-synthesised row h-2)
      float Vx[3], Vxx[3], Vxy[3];
      if (kSliding) {   // the forward's sliding scheme: V = P + r(j), then P' = r(j-1) + r(j); every step, whatever its centre row
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float xn = X[SN][c], xo = X[SP][c];
          const float xx = xn*xn, xy = xn*Y[SN][c];
          Vx[c] = Px[c] + xn; Vxx[c] = Pxx[c] + xx; Vxy[c] = Pxy[c] + xy;
          Px[c] = xo + xn; Pxx[c] = fmaf(xo, xo, xx); Pxy[c] = fmaf(xo, Y[SP][c], xy);
        }
      }
      if (doB) {
        float g2;
        if (kScan) g2 = ((selbits >> (unsigned)(p - rb)) & 1u) ? g_ssim : 0.f;
        else g2 = (col_ok && routes(SELR[SP])) ? g_ssim : 0.f;
        if (!kSliding) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float xq = X[SQ][c], xp = X[SP][c], xn = X[SN][c];
            Vx[c] = (xq + xp) + xn; Vxx[c] = fmaf(xn, xn, fmaf(xp, xp, xq*xq)); Vxy[c] = fmaf(xn, Y[SN][c], fmaf(xp, Y[SP][c], xq*Y[SQ][c]));
          }
        }
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
      } else if (GATED && inB && ((N >> (unsigned)(p - 1 - rb)) & 7u) != 0u) {
        // a dead centre row next to a live one: a stage C will read its coefficient row — as zeros
        // (Without gating nothing needs clearing: every coefficient row that a stage C reads with a non-zero weight was computed.)
#pragma unroll
        for (int c = 0; c < 3; ++c) { HC[SP][c][0] = 0.f; HC[SP][c][1] = 0.f; HC[SP][c][2] = 0.f; }
      }
    }

    // ================= stage C: row q — dL/dx -> dL/d(sx, sy) -> depth, pose sums
    if (inC) {
      const unsigned qro = (unsigned)q*w4;
      float gD = 0.f;
      float D2 = 0.f;
      if (doC) {
        // weights of coefficient rows q-1 / q+1 in the gradient of row q (reflect_weights_adj, on the scalar unit)
        float lo_q = usel(q == 0, 0.f, usel(q == 1, 2.f, 1.f)), hi_q = usel(q == h - 1, 0.f, usel(q == h - 2, 2.f, 1.f));
        if (h == 2) { lo_q = usel(q == 1, 2.f, 0.f); hi_q = usel(q == 0, 2.f, 0.f); }
        float gl;
        if (kScan) gl = ((selbits >> (unsigned)(q - rb)) & 1u) ? g_l1 : 0.f;
        else gl = (col_ok && routes(SELR[SQ])) ? g_l1 : 0.f;
        D2 = hist[(SQ*kHist + 6)*64];
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
      // dL/d depth of this support.  K0 fused: d depth / d(up-sampled, scaled disparity) is applied here, where the depth is at
      // hand (linear, so per support); what reaches the depth from other consumers is added once, by the wave of support 0.
      if (XTRA && add_gin) {
        gD += bld(rs_gin, lane4, qro);
        if (GATED && !doC) D2 = bld(rs_depth, lane4, qro);     // a skipped row never had its depth parked
      }
      if (a.k0_scale != 0.f) gD *= (D2 < 1.f/kEps32) ? -D2*D2*a.k0_scale : 0.f;
#if (SMD_ABLATE_BWD & 4)
      if (gD == 12345.678f) bst(rs_gd, lane4, qro, gD);
      else
#endif
      if (ACC) {
        // The supports of a strip are summed from LDS by the strip's last wave (k_recon_bwd): each lane parks its own column, one
        // slot per strip row; no read-modify-write of g_depth, one global store per pixel.
        float* slot = gacc + (q - r0)*64;
        if (XTRA && acc_prev) gD += *slot;
        *slot = gD;
      } else if (interior) bst(rs_gd, lane4, qro, gD);   // a single support: the row is final
    }
    if (doI) hist[(SQ*kHist + 6)*64] = Dkeep;      // row j+1's slot: stage C reads it at step j+3 (no second load of the depth)
  }

  // -DSMD_BWD_FASTPATH (experiment, off): a gated wave takes the branch-free body wherever its masks say that everything the step touches is
  // live — rows j, j+1, j+2 needed, centre row j-1 live, row j-2 needed; one scalar test per step.
  template <int PH, int DOB, int DOC>
  __device__ __forceinline__ void step_any(int j) {
    if (SKIP == 0) { step<PH, DOB, DOC, false>(j); return; }
#ifdef SMD_BWD_FASTPATH   // (off: with both bodies in one loop the register allocator spills ~70 values at 128 VGPRs — as round 3's per-wave choice did)
    if (DOB == 1 && DOC == 1) {                                   // steady state of the peeled pipeline: j - 2 >= r0 > rb
      const unsigned nb = N >> (unsigned)(j - 2 - rb);            // bit 0: row j-2 ... bit 4: row j+2
      if ((nb & 0x1du) == 0x1du && bit(L, j - 1)) { step<PH, DOB, DOC, false>(j); return; }
    }
#endif
    step<PH, DOB, DOC, true>(j);
  }

  __device__ __forceinline__ void run(int jstart) {
    begin(jstart);
    const int jend = r1 + 1;
    int j = jstart;
    if (kPeel) {                           // jstart = r0 - 2: r1 - r0 + 4 >= 5 steps
      step_any<0, 0, 0>(j); ++j;
      step_any<1, 0, 0>(j); ++j;
      step_any<2, 1, 0>(j); ++j;
      step_any<0, 1, 0>(j); ++j;
      for (;;) {
        step_any<1, 1, 1>(j); if (++j > jend) break;
        step_any<2, 1, 1>(j); if (++j > jend) break;
        step_any<0, 1, 1>(j); if (++j > jend) break;
      }
    } else {
      for (;;) {
        step_any<0, 2, 2>(j); if (++j > jend) break;
        step_any<1, 2, 2>(j); if (++j > jend) break;
        step_any<2, 2, 2>(j); if (++j > jend) break;
      }
    }
  }
};

// Four waves per SIMD (<= 128 VGPRs): without the cap the allocator settles at 133 and the kernel loses a wave of occupancy,
// 138 -> 127 us at cfg 2 (the gathers' latency is what the extra wave hides).
constexpr int kAccRows = 16;   // tallest strip of the multi-support instantiations: its dL/d depth rows wait in LDS for the strip's sum

// NS = waves per strip: wave (strip in block, k) handles supports k, k + NS, ... of its strip (NS = 1: every support, one after
// the other — balanced waves whatever the selection masks look like; NS = min(n, 4): half / a quarter as long work units).
// ACC: more than one support — the strip's dL/d depth rows are summed in LDS.
#ifdef SMD_TRACE_WAVES
__device__ unsigned long long g_wave_trace_bwd[1 << 16][3];
#endif
// Does any pixel within one row / column of the strip (rows r0 .. r1-1, columns c0 .. c0+59) select `key`?  Conservative in the rows (a forward strip's
// mask covers all of its rows), exact in the columns.  The footprint overlaps at most 3 x 3 forward strips in every partition the heuristics choose:
// lane l < 9 fetches entry (l / 3, l mod 3) — indices beyond the footprint repeat its last strip — masks it with the lanes of its columns, and one
// ballot answers for the wave: ONE vector load, one round trip, a handful of short-lived VGPRs.  (Done on the scalar unit — nine s_loads and their
// 64-bit masks — it pushed the kernel over its SGPR budget: spills through VGPRs into scratch, +56 MB of HBM writes per launch in round 5's first PMC pass.)
__device__ __forceinline__ bool strip_is_live(const ReconBwdArgs& a, int s, int bi, int key, int r0, int r1, int c0, int lane) {
  const int frh = bi < a.fwd_b1 ? a.fwd_rh : a.fwd_rh2;            // rows per forward strip of this sample (the forward's partition: launch arguments, no dependent load)
  const int fnsx = ceil_div(a.w, kFwdCols);
  const int rlo = max(r0 - 1, 0), rhi = min(r1, a.h - 1), clo = max(c0 - 1, 0), chi = min(c0 + kBwdCols, a.w - 1);
  const int fy0 = rlo/frh, fy1 = rhi/frh, fx0 = clo/kFwdCols, fx1 = chi/kFwdCols;
  if (key >= kLiveSupports || fy1 - fy0 > 2 || fx1 - fx0 > 2) return true;   // (strips of 4 rows against a tall backward strip: knob settings only)
  const unsigned long long* tab = reinterpret_cast<const unsigned long long*>(a.live + live_header_floats(a.b))
                                  + ((size_t)s*a.b + bi)*live_max_strips(a.h, a.w)*kLiveSupports + key;
  const int q = min(lane, 8), dy = q/3, dx = q - dy*3;
  const int fy = min(fy0 + dy, fy1), fx = min(fx0 + dx, fx1);
  const unsigned long long m = tab[((size_t)fy*fnsx + fx)*kLiveSupports];
  // forward lane l of strip fx holds column 62 fx - 1 + l: the lanes of columns max(clo, 62 fx - 1) .. min(chi, 62 fx + 62)
  const int l0 = max(clo - (fx*kFwdCols - 1), 0), l1 = min(chi - (fx*kFwdCols - 1), 63);
  const unsigned long long range = (l1 >= 63 ? ~0ull : ((1ull << (l1 + 1)) - 1ull)) & ~((1ull << l0) - 1ull);
  return __ballot((m & range) != 0ull) != 0ull;
}

template <bool SSIM, int SKIP, int NS, bool ACC, bool XTRA>
__global__ __launch_bounds__(64*kWavesPerBlock, 4) void k_recon_bwd(const ReconBwdArgs a) {
#ifdef SMD_TRACE_WAVES
  const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime();
#endif
  static_assert(ACC || NS == 1, "several waves per strip need the LDS sum");
  constexpr int SPB = (kWavesPerBlock/NS > 0) ? kWavesPerBlock/NS : 1;   // strips per block
  // (a wave's region is at least the epilogue's scratch, which aliases it: single-support strips have no dL/d depth rows)
  constexpr int kWaveFloats = ((3*kHist + (ACC ? kAccRows : 0))*64 >= 2*kFinScratchDoubles) ? (3*kHist + (ACC ? kAccRows : 0))*64 : 2*kFinScratchDoubles;
  static_assert(kWaveFloats*4 >= kFinScratchDoubles*8, "the epilogue's scratch aliases a wave's LDS rows");
  // per wave: 3 row slots x ({gx, gy} x 3 channels + depth) x 64 lanes [+ ACC: kAccRows x 64 lanes of dL/d depth]; ONE array (a second
  // __shared__ object in the row loop makes the compiler serialise LDS and vector-memory waits); the arrival counters of the
  // epilogue chain sit behind it
  constexpr int kPoseArea = SMD_MAX_SUPPORTS*kPoseSums;      // per wave: its pose sums, [support][12], zero for the supports it does not handle
  constexpr int kLdsFloats = SPB*NS*(kWaveFloats + kPoseArea) + 8;
  __shared__ __attribute__((aligned(16))) float hist_lds[kLdsFloats];
  float* const pose_lds = hist_lds + SPB*NS*kWaveFloats;
  unsigned* const cnt = reinterpret_cast<unsigned*>(pose_lds + SPB*NS*kPoseArea);   // [0 .. SPB): waves of a strip done; [4]: waves of the block done
  const int lane = threadIdx.x & 63;
  // Which support a wave of the block takes ROTATES with the block index (round 5).  The hardware places wave j of a workgroup on SIMD j mod 4, so
  // with "wave j = support j" every wave of support k in the launch shared a SIMD with the other waves of support k.  `wid` below is the wave's
  // LOGICAL index (strip, support): LDS regions, the order of the cross-support sum and of the pose sums all follow it, so the results are the same
  // bits as without the rotation.  (blockIdx >> 3: workgroup p runs on XCD p mod 8, so the blocks that share a CU differ in p >> 3.)
  const int wid_hw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sib = wid_hw/NS, kw = (wid_hw - sib*NS + (int)((blockIdx.x >> 3) % (unsigned)NS)) % NS;   // strip within the block, first support of this wave
  const int wid = sib*NS + kw;
  // segment of the (possibly tapered) partition this block belongs to (smd_kernels.h: ReconMainArgs::b1)
  const unsigned nblk1 = recon_grid_blocks(a.nsx*a.nsy, a.b1, a.S, SPB);
  const bool tail = blockIdx.x >= nblk1;
  const int nstrips = a.nsx*(tail ? a.nsy2 : a.nsy), seg_rh = tail ? a.rh2 : a.rh;
  const int nbx = ceil_div(nstrips, SPB);
  int xb, bi, s;
  decode_tile(tail ? blockIdx.x - nblk1 : blockIdx.x, nbx, tail ? a.b - a.b1 : a.b1, a.S, xb, bi, s);
  const int slot = s;                      // the block's slot among the S blocks of its tile: indexes its pose-sum entry
  // Four scales (round 5), fewer than four waves per strip: the block's waves are SPB SCALES of ONE strip — the decode's scale index picks the strip
  // within a tile of SPB adjacent strips and the group of SPB scales — instead of SPB strips of one scale.  A wave that takes its supports in turn
  // re-reads the strip's target-side rows (target pixel, window terms: 44 of the 49 bytes per pixel are the same for every scale) once per support;
  // waves that walk the same rows of the same strip share them through one L1, and gather near the same texels.
  const bool scales_block = SPB > 1 && a.scales_block != 0;
  const int strip = scales_block ? xb*SPB + slot % SPB : xb*SPB + sib;
  if (scales_block) s = (slot/SPB)*SPB + sib;
  if (tail) bi += a.b1;
  const int sxi = strip % a.nsx, syi = strip/a.nsx;
  // Liveness of this wave's first support (round 5), asked BEFORE the block's start-up barrier so that the scalar loads of the table entries are in
  // flight while the block zeroes its LDS: a check placed in front of the row loop cost every wave two dependent round trips (+5 % at cfg 2, where
  // nothing is dead).
  // Plain loop only: the gated loop finds a dead strip with its own scan of `sel` (its prologue and the zero fill are all it then runs), and it sits at
  // its register limit — the probe cost it three spilled dwords per lane for 5 % where nearly everything is masked.
  constexpr bool kProbe = SKIP == 0;
  bool first_live = true;
  if (kProbe && ACC && a.live != nullptr && a.g_in == nullptr && strip < nstrips)
    first_live = strip_is_live(a, s, bi, (a.flags & SMD_USE_MIN) ? kw : 0, syi*seg_rh, min(syi*seg_rh + seg_rh, a.h), sxi*kBwdCols, lane);
  for (int e = threadIdx.x; e < SPB*NS*kPoseArea; e += 64*SPB*NS) pose_lds[e] = 0.f;
  if (threadIdx.x < 8) cnt[threadIdx.x] = 0u;
  __syncthreads();                                           // the only block barrier: at the start, where every wave still is
  constexpr int nw = NS;                                   // waves that work on a strip
  if (strip >= nstrips) return;                // nothing to do (the chain below counts live waves only)
  const int live_waves = scales_block ? SPB*nw : min(SPB, nstrips - xb*SPB)*nw;   // (scales_block: the host guarantees nstrips % SPB == 0, every block has its strip)
  const int h = a.h, w = a.w;
  const int r0 = syi*seg_rh, r1 = min(r0 + seg_rh, h);
  const int u = sxi*kBwdCols - 2 + lane;
  // data column: the lane's own, or the reflected one for the halo lanes outside the image (reflection by data)
  const int uc = (u < 0) ? min(-u, w - 1) : ((u >= w) ? max(2*(w - 1) - u, 0) : u);
  const bool interior = (lane >= 2) && (lane < 2 + kBwdCols) && (u < w);
  const size_t hw = (size_t)h*w;
  const size_t sb = ((size_t)s*a.b + bi)*hw;
  float* const wave_lds = hist_lds + wid*kWaveFloats;

  {
    BwdCtx<SSIM, SKIP, ACC, XTRA> cx{a};
    cx.hist = wave_lds + lane;
    cx.gacc = cx.hist + 3*kHist*64;
    cx.h = h; cx.w = w;
    cx.r0 = r0; cx.r1 = r1; cx.rb = r0 - 3;
    const bool col_ok = (u >= 0) && (u < w);
    cx.interior = interior; cx.col_ok = col_ok;
    cx.lane4 = (unsigned)uc*4u; cx.lane1 = (unsigned)uc;
    reflect_weights_adj(min(max(u, 0), w - 1), w, cx.wla, cx.wra);   // how much column u receives from u-1 / u+1
    if (!col_ok) { cx.wla = 0.f; cx.wra = 0.f; }
    const float uf = (float)uc;

    cx.use_min = a.flags & SMD_USE_MIN;
    const unsigned hw4 = (unsigned)hw*4u;
    cx.w4 = (unsigned)w*4u;
    float gscale = (a.g_loss[0]*a.g_scale)/((float)a.S*(float)a.b*(float)h*(float)w);
    if (!cx.use_min) gscale /= (float)a.n;
    cx.g_ssim = uniform(gscale*(SSIM ? kWSsim/3.f : 0.f));      // (columns outside the image never route: scan_sel)
    cx.g_l1 = uniform(gscale*(SSIM ? (1.f - kWSsim)/3.f : 1.f/3.f));
    cx.xmax = (float)(w - 1); cx.ymax = (float)(h - 1); cx.wpf = (float)(w + 1);

    cx.rs_pk = make_rsrc(a.packed, packed_image_floats(a.b, a.n, h, w)*4);
    cx.rs_depth = make_rsrc(a.depth + sb, hw*4);
    cx.rs_sel = make_rsrc(a.sel + sb, hw);
    cx.rs_gd = make_rsrc(s == a.direct_scale ? a.g_direct + (size_t)bi*hw : a.g_depth + sb, hw*4);
    const bool has_gin = a.g_in != nullptr;
    cx.rs_gin = make_rsrc(has_gin ? a.g_in + sb : nullptr, has_gin ? hw*4 : 0);
    const unsigned texel_bytes = (unsigned)(h + 1)*(unsigned)(w + 1)*12u;
    cx.rowbytes = ((unsigned)w + 1u)*12u;
    cx.so_y = (unsigned)(packed_texel_floats(a.b, a.n, h, w)*4) + (unsigned)bi*hw4*3u;
    cx.so_ta = (unsigned)((packed_texel_floats(a.b, a.n, h, w) + packed_ypix_floats(a.b, h, w))*4) + (unsigned)bi*hw4*4u;
    cx.so_tb = cx.so_ta + (unsigned)(packed_tpix_floats(a.b, h, w)*4);

    // rows outside the image are reflected ones (BwdCtx::reflect_row).  Peeled form: every strip runs rows r0-2 .. r1+1 with centre
    // rows r0-1 .. r1 (-1 and h: dummy rows, exact zeros); compact form: the image's top strip starts at row -1 (= row 1) and no dummy rows
    const bool peel = BwdCtx<SSIM, SKIP, ACC, XTRA>::kPeel;
    const int jstart = peel ? r0 - 2 : max(r0 - 2, -1);
    cx.pb0 = peel ? r0 - 1 : max(r0 - 1, 0); cx.pb1 = peel ? r1 : min(r1, h - 1);   // centre rows whose coefficients are needed

    for (int i = kw; i < a.n; i += NS) {
      cx.add_gin = has_gin && i == 0;
      cx.acc_prev = i != kw;
      // Liveness (round 5): the forward left, per forward strip and support, the columns in which some row selects that support.  If no such
      // column lies within one pixel of this strip's rows r0-1 .. r1 and columns c0-1 .. c0+60, every gradient this wave would compute for support
      // i is an exact zero (the plain row loop multiplies by the routing mask) — park zeros and go on.  Wave-uniform, a few scalar loads.
      if (kProbe && ACC && a.live != nullptr && !cx.add_gin && !(i == kw ? first_live : strip_is_live(a, s, bi, cx.use_min ? i : 0, r0, r1, sxi*kBwdCols, lane))) {
        if (!cx.acc_prev) for (int r = 0; r < r1 - r0; ++r) cx.gacc[r*64] = 0.f;
        continue;     // (its pose sums stay the zeros the block started with)
      }
      make_cam2(cx.cm, cx.hx0, cx.hy0, cx.hz0, a.T + ((size_t)i*a.b + bi)*16, a.K + (size_t)bi*16, a.Kinv + (size_t)bi*16,
                a.wscale, a.hscale, uf);
      cx.so_tex = (unsigned)(i*a.b + bi)*texel_bytes;
      cx.sel_key = cx.use_min ? (unsigned)i : (unsigned)SMD_SEL_MASKED;
      cx.run(jstart);

      // per-wave pose sums: d/d(H[0..8], a0, a1, tz) of the UN-scaled homography (rows 0/1 of the folded one carry the grid
      // scale); the column factor of H[.,0] is constant per lane.  Parked in this wave's row of the block's pose area (LDS).
      const float* ps = cx.ps;
      const float ws = a.wscale, hs = a.hscale;
      const float psum[kPoseSums] = {ps[0]*uf*ws, ps[1]*ws, ps[0]*ws, ps[2]*uf*hs, ps[3]*hs, ps[2]*hs, ps[4]*uf, ps[5], ps[4],
                                     ps[6]*ws, ps[7]*hs, ps[8]};
      float mine = 0.f;
#pragma unroll
      for (int k = 0; k < kPoseSums; ++k) {
        const float tot = wave_sum(psum[k]);
        if (lane == k) mine = tot;
      }
      if (lane < kPoseSums) pose_lds[wid*kPoseArea + i*kPoseSums + lane] = mine;
    }
  }

#ifdef SMD_TRACE_WAVES   // diagnosis builds only (scripts/dev/wave_trace.py bwd): when and where this wave ran its row loop
  if (lane == 0) {
    const unsigned widx = blockIdx.x*kWavesPerBlock + wid;
    if (widx < (1u << 16)) {
      g_wave_trace_bwd[widx][0] = trace_t0; g_wave_trace_bwd[widx][1] = __builtin_amdgcn_s_memrealtime();
      g_wave_trace_bwd[widx][2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    }
  }
#endif
  // ---- epilogue chain, no block barrier (every wave of a block would idle through two memory round trips): a wave that is
  // done bumps LDS counters; the LAST wave of a strip sums the strip's dL/d depth rows, the LAST wave of the block counts the
  // block's arrival at agent scope, and the wave that completes a sample reduces its pose sums (Guideline 16 of
  // cdna_hip_programming.md: write-through payload, drained by every storing wave before it is counted; one acquire by the reader).
  unsigned o_strip = 0, o_block = 0;
  if (lane == 0) {
    if (ACC) o_strip = __hip_atomic_fetch_add(cnt + sib, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);   // orders this wave's LDS rows before, the others' after
    o_block = __hip_atomic_fetch_add(cnt + 4, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  o_strip = (unsigned)__builtin_amdgcn_readfirstlane((int)o_strip); o_block = (unsigned)__builtin_amdgcn_readfirstlane((int)o_block);
  if (ACC && o_strip == (unsigned)nw - 1u) {
    // dL/d depth of the strip = sum over its waves' LDS rows in wave order (deterministic whichever wave adds them up)
    const float* base = hist_lds + (sib*NS)*kWaveFloats + 3*kHist*64 + lane;
    const rsrc_t rs_gd = make_rsrc(s == a.direct_scale ? a.g_direct + (size_t)bi*hw : a.g_depth + sb, hw*4);
#pragma unroll 4
    for (int r = 0; r < r1 - r0; ++r) {
      float g = base[r*64];
#pragma unroll
      for (int k = 1; k < NS; ++k) g += base[k*kWaveFloats + r*64];
      if (interior) bst(rs_gd, (unsigned)uc*4u, (unsigned)(r0 + r)*(unsigned)w*4u, g);
    }
  }
  if (o_block != (unsigned)live_waves - 1u) return;
  // the last wave of the block: the block's pose sums = its waves' sums in wave order (a wave without a strip left its zeros),
  // published write-through as ONE entry per (support, block)
  for (int e = lane; e < a.n*kPoseSums; e += 64) {
    const int i = e/kPoseSums, k = e - i*kPoseSums;
    float v = 0.f;
#pragma unroll
    for (int wv = 0; wv < SPB*NS; ++wv) v += pose_lds[wv*kPoseArea + e];
    float* pp = a.pose_partial + (((size_t)i*a.b + bi)*(size_t)a.pose_stride + (size_t)slot*nbx + xb)*kPoseSums;
    __hip_atomic_store((unsigned*)(pp + k), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (a.arrive == nullptr) return;   // the sample's epilogue rides in the launch that follows (K0 adjoint: smd_depth.hip), no hand-off needed
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned last = 0;
  if (lane == 0) {
    const unsigned expected = (unsigned)nbx*(unsigned)a.S;
    last = (__hip_atomic_fetch_add(a.arrive + bi, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expected - 1u) ? 1u : 0u;
  }
  if (!__builtin_amdgcn_readfirstlane((int)last)) return;
  SMD_TAIL_ACQUIRE();
  // scratch: the head of this wave's own history rows (free: its row loop is over, and a strip's sum reads only the dL/d depth rows behind them)
  pose_finalize_wave(a, bi, a.S*nbx, reinterpret_cast<double*>(wave_lds));
  if (lane == 0) __hip_atomic_store(a.arrive + bi, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call on this buffer
}

#ifdef SMD_EXPERIMENTS
#include "experiments/smd_recon_bwd_pair.inc"   // k_recon_bwd_pair (dropped in round 4; knob bwd_pair)
#endif

// The same epilogue as a launch of its own, for the un-fused ViewSynth backward (smd_unfused.hip), whose partials come from a
// one-thread-per-pixel kernel: one block of four waves per sample.
__global__ __launch_bounds__(256) void k_pose_finalize(const ReconBwdArgs a, int entries) {
  __shared__ double scratch[fin_scratch_doubles(4)];
  pose_finalize<true>(a, (int)blockIdx.x, entries, scratch, (int)(threadIdx.x >> 6), 4);
}

hipError_t launch_pose_finalize(const float* pose_partial, int entries, int stride, const float* T, const float* K, const float* Kinv,
                                float* g_T, float* g_K, float* g_Kinv, int b, int n, hipStream_t st) {
  ReconBwdArgs a = {};
  a.pose_partial = const_cast<float*>(pose_partial); a.pose_stride = stride; a.T = T; a.K = K; a.Kinv = Kinv;
  a.g_T = g_T; a.g_K = g_K; a.g_Kinv = g_Kinv; a.b = b; a.n = n;
  hipLaunchKernelGGL(k_pose_finalize, dim3(b), dim3(256), 0, st, a, entries);
  return hipGetLastError();
}

template <bool SSIM, int SKIP, int NS, bool ACC>
static void launch_bwd_t(dim3 grid, dim3 block, hipStream_t st, const ReconBwdArgs& a) {
  // the common case — every support has its own wave, nothing else feeds the depth — runs the instantiation without the two per-step branches
  const bool xtra = a.g_in != nullptr || a.n > NS;
  note_variant(1, "smd::k_recon_bwd<%s, %d, %d, %s, %s>%s", SSIM ? "true" : "false", SKIP, NS, ACC ? "true" : "false", xtra ? "true" : "false",
               (a.scales_block && kWavesPerBlock/NS > 1) ? " [a block = the scales of one strip]" : "");
  if (xtra) hipLaunchKernelGGL((k_recon_bwd<SSIM, SKIP, NS, ACC, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((k_recon_bwd<SSIM, SKIP, NS, ACC, false>), grid, block, 0, st, a);
}

#ifdef SMD_EXPERIMENTS
template <bool SSIM, int NP>
static void launch_bwd_pair_t(dim3 grid, hipStream_t st, const ReconBwdArgs& a) {
  note_variant(1, "smd::k_recon_bwd_pair<%s, %d>", SSIM ? "true" : "false", NP);
  hipLaunchKernelGGL((k_recon_bwd_pair<SSIM, NP>), grid, dim3(64*kPairWaves), 0, st, a);
}
#endif

hipError_t launch_recon_bwd(const ReconBwdArgs& a, hipStream_t st) {
#ifndef SMD_EXPERIMENTS
  if (a.pair) return hipErrorInvalidValue;
#else
  if (a.pair) {   // two supports per wave: n = 2 or 4, min-reprojection, plain row loop (smd_api.hip decides)
    if (!(a.n == 2 || a.n == 4) || !(a.flags & SMD_USE_MIN) || a.rh > kAccRows || a.rh2 > kAccRows) return hipErrorInvalidValue;
    const int np = a.n/2, spbp = kPairWaves/np;
    dim3 gridp(recon_grid_blocks(a.nsx*a.nsy, a.b1, a.S, spbp) + (a.b1 < a.b ? recon_grid_blocks(a.nsx*a.nsy2, a.b - a.b1, a.S, spbp) : 0u));
    const bool ssimp = !(a.flags & SMD_LOSS_L1);
    if (ssimp) { if (np == 1) launch_bwd_pair_t<true, 1>(gridp, st, a); else launch_bwd_pair_t<true, 2>(gridp, st, a); }
    else { if (np == 1) launch_bwd_pair_t<false, 1>(gridp, st, a); else launch_bwd_pair_t<false, 2>(gridp, st, a); }
    return hipGetLastError();
  }
#endif
  const int ns = a.wps, spb = kWavesPerBlock/ns;
  if (ns < 1 || ns > 4 || ns > a.n) return hipErrorInvalidValue;
  dim3 grid(recon_grid_blocks(a.nsx*a.nsy, a.b1, a.S, spb) + (a.b1 < a.b ? recon_grid_blocks(a.nsx*a.nsy2, a.b - a.b1, a.S, spb) : 0u)), block(64*ns*spb);
  const bool ssim = !(a.flags & SMD_LOSS_L1);
  if (a.n > 1 && (a.rh > kAccRows || (a.b1 < a.b && a.rh2 > kAccRows))) return hipErrorInvalidValue;   // the strip's rows must fit the LDS sum (smd_api.hip clamps)
#define SMD_BWD_NS(SSIM_, SKIP_) do { \
    if (a.n == 1) launch_bwd_t<SSIM_, SKIP_, 1, false>(grid, block, st, a); \
    else switch (ns) { \
    case 1: launch_bwd_t<SSIM_, SKIP_, 1, true>(grid, block, st, a); break; \
    case 2: launch_bwd_t<SSIM_, SKIP_, 2, true>(grid, block, st, a); break; \
    case 3: launch_bwd_t<SSIM_, SKIP_, 3, true>(grid, block, st, a); break; \
    default: launch_bwd_t<SSIM_, SKIP_, 4, true>(grid, block, st, a); break; } } while (0)
  if (ssim) {
    if (a.skip_level >= 1) SMD_BWD_NS(true, 2); else SMD_BWD_NS(true, 0);
  } else SMD_BWD_NS(false, 0);
#undef SMD_BWD_NS
  return hipGetLastError();
}

}  // namespace smd

#ifdef SMD_TRACE_WAVES
extern "C" int smd_debug_wave_trace_bwd(unsigned long long* host_out, int max_waves) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(smd::g_wave_trace_bwd), (size_t)max_waves*3*sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif
