// smd_recon_bwd.hip — hand-written adjoint of the fused view-synthesis photometric loss (gfx950), round-2 structure.
//
// Same wave-strip streaming as the forward (smd_recon_fwd.hip): 60 interior columns + 2 halo lanes per side, rows
// r0-2 .. r1+1, one support per pass of the row loop (the state of one support is ~75 registers; two at once would halve the
// occupancy of a kernel that is bound by memory latency), three stages per row step j:
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

namespace smd {

// Three independent reflection-weighted horizontal 3-tap sums in one asm block: r = q + wl*left(q) + wr*right(q) as two
// DPP-sourced v_fmac per value (the shift rides on the FMA's first operand); one hazard nop covers all of them.
__device__ __forceinline__ void hsum_w3(float a, float b, float c, float wl, float wr, float& ra, float& rb, float& rc) {
#ifdef SMD_NO_DPP
  ra = hsum3(a, wl, wr); rb = hsum3(b, wl, wr); rc = hsum3(c, wl, wr);
#else
  ra = a; rb = b; rc = c;
  asm volatile("s_nop 1\n\t"
               "v_fmac_f32_dpp %0, %3, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "v_fmac_f32_dpp %1, %4, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "v_fmac_f32_dpp %2, %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "v_fmac_f32_dpp %0, %3, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "v_fmac_f32_dpp %1, %4, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "v_fmac_f32_dpp %2, %5, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
               : "+&v"(ra), "+&v"(rb), "+&v"(rc)
               : "v"(a), "v"(b), "v"(c), "v"(wl), "v"(wr));
#endif
}

// SKIP: 0 = every row does the full adjoint; 2 = rows where no pixel of the wave selected the current support skip the SSIM
// partials, and rows whose three coefficient rows and L1 term are all dead skip the chain rule.  Coherent selection /
// automask regions (any partly trained network) make 2 the fastest; on noise-like masks the branches cost a few percent.
template <bool SSIM, int SKIP>
__global__ __launch_bounds__(64*kWavesPerBlock) void k_recon_bwd(const ReconBwdArgs a) {
  __shared__ float hist_lds[kWavesPerBlock*3*6*64];   // per wave: 3 row slots x {gx, gy} x 3 channels x 64 lanes
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nstrips = a.nsx*a.nsy;
  int strip, bi, s;
  decode_wave(blockIdx.x, wid, nstrips, a.b, a.S, strip, bi, s);
  if (strip >= nstrips) return;
  float* const hist = hist_lds + wid*(3*6*64) + lane;
  const int sxi = strip % a.nsx, syi = strip/a.nsx;
  const int h = a.h, w = a.w;
  const int r0 = syi*a.rh, r1 = min(r0 + a.rh, h);

  const int u = sxi*kBwdCols - 2 + lane;
  const bool col_ok = (u >= 0) && (u < w);
  // data column: the lane's own, or the reflected one for the halo lanes outside the image (reflection by data)
  const int uc = (u < 0) ? min(-u, w - 1) : ((u >= w) ? max(2*(w - 1) - u, 0) : u);
  const bool interior = (lane >= 2) && (lane < 2 + kBwdCols) && (u < w);
  const unsigned lane4 = (unsigned)uc*4u, lane1 = (unsigned)uc;
  float wla, wra;                                   // adjoint reflection weights: how much column u receives from u-1 / u+1
  reflect_weights_adj(min(max(u, 0), w - 1), w, wla, wra);
  if (!col_ok) { wla = 0.f; wra = 0.f; }
  const float uf = (float)uc;

  const bool use_min = a.flags & SMD_USE_MIN;
  const size_t hw = (size_t)h*w;
  const unsigned hw4 = (unsigned)hw*4u, w4 = (unsigned)w*4u;
  const float w_ssim = SSIM ? kWSsim/3.f : 0.f, w_l1 = SSIM ? (1.f - kWSsim)/3.f : 1.f/3.f;
  float gscale = a.g_loss[0]/((float)a.S*(float)a.b*(float)h*(float)w);
  if (!use_min) gscale /= (float)a.n;
  constexpr float c1 = 81.f*kC1;   // window sums stay un-normalised (x9), see smd_recon_fwd.hip
  const float xmax = (float)(w - 1), ymax = (float)(h - 1), wpf = (float)(w + 1);

  const size_t sb = ((size_t)s*a.b + bi)*hw;
  const rsrc_t rs_pk = make_rsrc(a.packed, packed_total_floats(a.b, a.n, h, w)*4);
  const rsrc_t rs_depth = make_rsrc(a.depth + sb, hw*4);
  const rsrc_t rs_sel = make_rsrc(a.sel + sb, hw);
  const rsrc_t rs_gd = make_rsrc(a.g_depth + sb, hw*4);
  const rsrc_t rs_gin = make_rsrc(a.g_in ? a.g_in + sb : nullptr, a.g_in ? hw*4 : 0);
  const bool direct0_scale = (a.g_disp0 != nullptr) && (s == 0);
  const rsrc_t rs_gd0 = make_rsrc(direct0_scale ? a.g_disp0 + (size_t)bi*hw : nullptr, direct0_scale ? hw*4 : 0);
  const unsigned texel_bytes = (unsigned)(h + 1)*(unsigned)(w + 1)*12u, rowbytes = ((unsigned)w + 1u)*12u;
  const unsigned so_y = (unsigned)(packed_texel_floats(a.b, a.n, h, w)*4) + (unsigned)bi*hw4*3u;
  const unsigned so_ta = (unsigned)((packed_texel_floats(a.b, a.n, h, w) + packed_ypix_floats(a.b, h, w))*4) + (unsigned)bi*hw4*4u;
  const unsigned so_tb = so_ta + (unsigned)(packed_tpix_floats(a.b, h, w)*4);

  const int jstart = max(r0 - 2, 0);
  const int jlast = min(r1 + 1, h - 1);             // last real row that is loaded
  const int pb0 = max(r0 - 1, 0), pb1 = min(r1, h - 1);   // centre rows whose coefficients are needed

  for (int i = 0; i < a.n; ++i) {
    Cam cm;
    make_cam(cm, a.T + ((size_t)i*a.b + bi)*16, a.K + (size_t)bi*16, a.Kinv + (size_t)bi*16);
    const float hx0 = fmaf(cm.H[0], uf, cm.H[2]), hy0 = fmaf(cm.H[3], uf, cm.H[5]), hz0 = fmaf(cm.H[6], uf, cm.H[8]);
    const unsigned so_tex = (unsigned)(i*a.b + bi)*texel_bytes;

    // state: rows j-1 (o) and j-2 (q) of the raw values, the sliding sums, the bilinear partials of rows j, j-1, j-2
    float xo[3] = {}, xq[3] = {}, yo[3] = {}, yq[3] = {};
    float Px[3] = {}, Pxx[3] = {}, Pxy[3] = {};
    float gx0[3] = {}, gy0[3] = {};                   // dx/dpx, dx/dpy of the current row (clamp mask, grid scale folded in)
    float ac1[3][3] = {}, ac0[3][3] = {};             // vertical accumulators of the h-summed coefficient maps {A, B, C}
    float ps[9] = {};                                 // per-lane sums of {dnx, dnx*v, dny, dny*v, dz, dz*v, gnx, gny, gz}
    float D0 = 0.f, D1 = 0.f, D2 = 0.f;               // depth of rows j, j-1, j-2
    unsigned selp = SMD_SEL_MASKED, selq = SMD_SEL_MASKED;
    unsigned live_hist = 0;                           // bit k: stage B of the k-th most recent row produced coefficients (wave-uniform)
    f3 t0 = {}, t1 = {}, t2 = {}, t3 = {}, py = {};   // loads in flight for the next row
    float pfx = 0.f, pfy = 0.f, pkx = 0.f, pky = 0.f;

    auto issue = [&](int jr, float D) {               // coordinates + gathers + target row of row jr
      const float vf = (float)jr;
      const float hx = fmaf(cm.H[1], vf, hx0), hyy = fmaf(cm.H[4], vf, hy0), hz = fmaf(cm.H[7], vf, hz0);
      const float nx = fmaf(D, hx, cm.a0), ny = fmaf(D, hyy, cm.a1), yz = fmaf(D, hz, cm.tz);
      const float rz = __builtin_amdgcn_rcpf(fmaxf(yz, kZMin));
      const float sx = fmaf(nx*rz, a.wscale, -0.5f), sy = fmaf(ny*rz, a.hscale, -0.5f);
      pkx = (sx > 0.f && sx < xmax) ? a.wscale : 0.f;   // d(clamped coordinate)/d(unclamped), times the grid scale
      pky = (sy > 0.f && sy < ymax) ? a.hscale : 0.f;
      const float cx = __builtin_amdgcn_fmed3f(sx, 0.f, xmax), cy = __builtin_amdgcn_fmed3f(sy, 0.f, ymax);
      const float x0 = floorf(cx), y0 = floorf(cy);
      pfx = cx - x0; pfy = cy - y0;
      const unsigned o = __umul24((unsigned)fmaf(y0, wpf, x0), 12u);
      t0 = bld3(rs_pk, o, so_tex); t1 = bld3(rs_pk, o + 12u, so_tex);
      t2 = bld3(rs_pk, o, so_tex + rowbytes); t3 = bld3(rs_pk, o + 12u, so_tex + rowbytes);
      py = bld3(rs_pk, lane4*3u, so_y + (unsigned)jr*w4*3u);
    };

    int slot = 0;
    float Da = bld(rs_depth, lane4, (unsigned)jstart*w4);       // depth of the row whose loads are in flight
    issue(jstart, Da);
    float Db = bld(rs_depth, lane4, (unsigned)(jstart + 1)*w4); // ... and of the row after it (a row below the image reads 0)

    for (int j = jstart; j <= r1 + 1; ++j) {
      // ---- roll the two-row history (row j-1 -> j-2) before row j overwrites the "current" slots
      D2 = D1; D1 = D0; selq = selp;
      const int p = j - 1, q = j - 2;
      const bool doB = SSIM && p >= pb0 && p <= pb1;
      const bool doC = q >= r0 && q < r1;
      // ================= stage A: row j =================
      float xn[3], yn[3];
      f4 ta = {}; f3 tb = {};
      if (doB || (!SSIM && p >= 0 && p < h)) selp = bld8(rs_sel, lane1, (unsigned)p*(unsigned)w);
      if (doB) {
        ta = bld4(rs_pk, lane4*4u, so_ta + (unsigned)p*w4*4u);
        tb = bld3(rs_pk, lane4*4u, so_tb + (unsigned)p*w4*4u);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float dn = t1[c] - t0[c], ds = t3[c] - t2[c];
        const float top = fmaf(pfx, dn, t0[c]), bot = fmaf(pfx, ds, t2[c]);
        const float ddy = bot - top;
        xn[c] = fmaf(pfy, ddy, top);
        gx0[c] = fmaf(pfy, ds - dn, dn)*pkx;
        gy0[c] = ddy*pky;
        yn[c] = py[c];
      }
      // The partials are needed two row steps later (stage C).  Twelve registers of history per lane would cost the kernel a wave
      // of occupancy, so they wait in LDS: each lane writes and later reads only its own column slot (no synchronisation).
#pragma unroll
      for (int c = 0; c < 3; ++c) { hist[(slot*6 + c)*64] = gx0[c]; hist[(slot*6 + 3 + c)*64] = gy0[c]; }
      D0 = Da;
      // Next row's loads, unconditionally (also after the last row, where nothing consumes them): the tap coordinates are
      // clamped, a depth row below the image reads 0 (buffer bounds), and a conditional issue would turn every register of
      // the in-flight loads into a loop phi with a second copy.
      issue(j + 1, Db);
      Da = Db;
      Db = bld(rs_depth, lane4, (unsigned)(j + 2)*w4);
      if (j > jlast) {
        // j == h: the row below the image is row h-2 (ReflectionPad2d(1)), which is still in the history; beyond that the
        // values are never used
#pragma unroll
        for (int c = 0; c < 3; ++c) { xn[c] = xq[c]; yn[c] = yq[c]; }
      }

      // ================= stage B: centre row p = j-1 — SSIM partials, h-summed with the adjoint weights =================
      float hc[3][3] = {};
      if (SSIM) {
        const float m = (p == 0) ? 2.f : 1.f;         // row -1 is row 1
        float Vx[3], Vxx[3], Vxy[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float xx = xn[c]*xn[c], xy = xn[c]*yn[c];
          Vx[c] = fmaf(m, xn[c], Px[c]); Vxx[c] = fmaf(m, xx, Pxx[c]); Vxy[c] = fmaf(m, xy, Pxy[c]);
          Px[c] = xo[c] + xn[c]; Pxx[c] = fmaf(xo[c], xo[c], xx); Pxy[c] = fmaf(xo[c], yo[c], xy);
        }
        if (!doB) live_hist <<= 1;
        else {
          const bool active = use_min ? (selp == (unsigned)i) : (selp != (unsigned)SMD_SEL_MASKED);
          const float g = (active && col_ok) ? gscale*w_ssim : 0.f;
          // Rows in which no pixel of this wave selected the current support carry no gradient through their windows:
          // skip the SSIM partials (wave-uniform branch; coherent regions of the min-reprojection / automask are common).
          const bool row_live = SKIP >= 1 ? (__builtin_amdgcn_ballot_w64(g != 0.f) != 0) : true;
          live_hist = (live_hist << 1) | (row_live ? 1u : 0u);
          if (row_live) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              float sx, sxx, sxy;
              hsum3(Vx[c], Vxx[c], Vxy[c], sx, sxx, sxy);
              const float sy = ta[c], cy2 = (c == 0) ? ta.w : ((c == 1) ? tb.x : tb.y), cy1 = fmaf(sy, sy, c1);
              // e = (1 - N/Dn)/2 with N = a1*a2, Dn = b1*b2 on the x9 sums (both scaled by 81*81)
              const float t = sx*sy;
              const float a1 = fmaf(2.f, t, c1), a2 = fmaf(2.f, fmaf(9.f, sxy, -t), 81.f*kC2);
              const float sx2 = sx*sx;
              const float b1 = sx2 + cy1, b2 = fmaf(9.f, sxx, -sx2) + cy2;
              const float rden = __builtin_amdgcn_rcpf(b1*b2);
              const float val = a1*a2*rden;
              const float e = fmaf(-0.5f, val, 0.5f);
              const float pass = (e >= 0.f && e <= 1.f) ? -0.5f*g : 0.f;   // d e/d val, gated by the clamp(0,1), times upstream
              // partials w.r.t. the x9 sums Sx, Sxx, Sxy (a1, a2, b1, b2 as functions of them):
              //   da1/dSx = 2 Sy, da2/dSx = -2 Sy, db1/dSx = 2 Sx, db2/dSx = -2 Sx, da2/dSxy = 18, db2/dSxx = 9
              const float prd = pass*rden;
              const float dSx = prd*(2.f*sy*(a2 - a1) - 2.f*sx*val*(b2 - b1));
              const float dSxx = prd*(-9.f*val*b1);
              const float dSxy = prd*(18.f*a1);
              hsum_w3(dSx, dSxx, dSxy, wla, wra, hc[c][0], hc[c][1], hc[c][2]);
            }
          }
        }
      }

      // ================= stage C: row q = j-2 — dL/dx -> dL/d(px,py) -> depth, pose sums =================
      if (doC) {
        float lo_q, hi_q;
        reflect_weights_adj(q, h, lo_q, hi_q);        // hi_q: weight of coefficient row q+1 in the gradient of row q
        const bool active = use_min ? (selq == (unsigned)i) : (selq != (unsigned)SMD_SEL_MASKED);
        const float gl = (active && col_ok) ? gscale*w_l1 : 0.f;
        // coefficient rows q-1, q, q+1 all skipped and no L1 term anywhere in the wave: the gradient of this row is zero
        const bool dead = SKIP >= 2 ? ((live_hist & 7u) == 0u && __builtin_amdgcn_ballot_w64(gl != 0.f) == 0) : false;
        const unsigned qro = (unsigned)q*w4;
        float gD = 0.f;
        if (!dead) {
          float gpx = 0.f, gpy = 0.f;
          const int rslot = (slot + 1) % 3;               // the slot written two steps ago
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float gx2c = hist[(rslot*6 + c)*64], gy2c = hist[(rslot*6 + 3 + c)*64];
            const float d = xq[c] - yq[c];
            float gxc = gl*((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f));
            if (SSIM) {
              const float SA = fmaf(hi_q, hc[c][0], ac1[c][0]), SB = fmaf(hi_q, hc[c][1], ac1[c][1]), SC = fmaf(hi_q, hc[c][2], ac1[c][2]);
              gxc += fmaf(2.f*xq[c], SB, fmaf(yq[c], SC, SA));   // d/dx_q of the x9 sums: 1, 2 x_q, y_q
            }
            gpx = fmaf(gxc, gx2c, gpx);
            gpy = fmaf(gxc, gy2c, gpy);
          }
          // projective chain rule at (q, u): the cheap geometry is recomputed from the depth kept in the ring
          const float vf = (float)q;
          const float hx = fmaf(cm.H[1], vf, hx0), hyy = fmaf(cm.H[4], vf, hy0), hz = fmaf(cm.H[7], vf, hz0);
          const float nx = fmaf(D2, hx, cm.a0), ny = fmaf(D2, hyy, cm.a1), yz = fmaf(D2, hz, cm.tz);
          const float rz = __builtin_amdgcn_rcpf(fmaxf(yz, kZMin));
          float gnx = gpx*rz, gny = gpy*rz;
          float gz = (yz >= kZMin) ? -(gpx*nx + gpy*ny)*rz*rz : 0.f;
          if (!interior) { gnx = 0.f; gny = 0.f; gz = 0.f; }
          gD = fmaf(gnx, hx, fmaf(gny, hyy, gz*hz));
          const float dnx = gnx*D2, dny = gny*D2, dz = gz*D2;
          ps[0] += dnx; ps[1] = fmaf(dnx, vf, ps[1]); ps[2] += dny; ps[3] = fmaf(dny, vf, ps[3]); ps[4] += dz; ps[5] = fmaf(dz, vf, ps[5]);
          ps[6] += gnx; ps[7] += gny; ps[8] += gz;
        }
        // dL/d depth: accumulated across the support passes through g_depth; the last pass adds what reaches depth from other
        // consumers and, for the full-resolution disparity scale of the K0-fused path, applies d depth/d disp on the spot.
        const bool last = (i == a.n - 1);
        const bool direct0 = last && direct0_scale;
        if (interior && (!dead || i == 0 || (last && (a.g_in != nullptr || direct0)))) {
          if (i != 0) gD += bld(rs_gd, lane4, qro);
          if (last && a.g_in != nullptr) gD += bld(rs_gin, lane4, qro);
          if (direct0) bst(rs_gd0, lane4, qro, gD*((D2 < 1.f/kEps32) ? -D2*D2 : 0.f)*a.a_scale);
          else bst(rs_gd, lane4, qro, gD);
        }
      }

      // ================= roll =================
      if (SSIM) {
        float lo_na, hi_na;
        reflect_weights_adj(min(max(p + 1, 0), h - 1), h, lo_na, hi_na);  // weight of coefficient row p in the gradient of row p+1
        if (p + 1 >= h || p < 0) lo_na = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int k = 0; k < 3; ++k) { ac1[c][k] = ac0[c][k] + hc[c][k]; ac0[c][k] = lo_na*hc[c][k]; }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) { xq[c] = xo[c]; xo[c] = xn[c]; yq[c] = yo[c]; yo[c] = yn[c]; }
      slot = (slot == 2) ? 0 : slot + 1;
    }

    // per-wave pose partials: d/d(H[0..8], a0, a1, tz); the column factor of H[.,0] is constant per lane
    float* pp = a.pose_partial + (((size_t)i*a.b + bi)*((size_t)a.S*nstrips) + (size_t)s*nstrips + strip)*kPoseSums;
    const float psum[kPoseSums] = {ps[0]*uf, ps[1], ps[0], ps[2]*uf, ps[3], ps[2], ps[4]*uf, ps[5], ps[4], ps[6], ps[7], ps[8]};
#pragma unroll
    for (int k = 0; k < kPoseSums; ++k) {
      const float tot = wave_sum(psum[k]);
      if (lane == 0) pp[k] = tot;
    }
  }
}

hipError_t launch_recon_bwd(const ReconBwdArgs& a, hipStream_t st) {
  dim3 grid(recon_grid_blocks(a.nsx*a.nsy, a.b, a.S)), block(64*kWavesPerBlock);
  const bool ssim = !(a.flags & SMD_LOSS_L1);
  if (ssim) {
    if (a.skip_level >= 1) hipLaunchKernelGGL((k_recon_bwd<true, 2>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_recon_bwd<true, 0>), grid, block, 0, st, a);
  } else hipLaunchKernelGGL((k_recon_bwd<false, 0>), grid, block, 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Epilogue: sum the per-wave partials of dL/d(H, a0, a1, tz) and push them through
//   H[0:2] = K2 * M,  H[2] = M[2],  M = R * Ki3,  (a0, a1) = K2 * t,  tz = t[2]
// to dL/dT (n,b,4,4), dL/dK (b,4,4), dL/dKinv (b,4,4).  One block per sample; fp64 accumulation.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pose_finalize(const float* __restrict__ pose_partial, int entries,
                                                       const float* __restrict__ T, const float* __restrict__ K,
                                                       const float* __restrict__ Kinv, float* g_T, float* g_K, float* g_Kinv,
                                                       int b, int n) {
  __shared__ double tot[kPoseSums];
  const int bi = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double gK[6] = {0, 0, 0, 0, 0, 0}, gKi[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const float* pp = pose_partial + ((size_t)i*b + bi)*(size_t)entries*kPoseSums;
    // each of the four waves reduces three of the twelve sums (fp64, fixed order -> deterministic)
    for (int k = wv*3; k < wv*3 + 3; ++k) {
      double acc = 0.0;
      for (int e = lane; e < entries; e += 64) acc += (double)pp[(size_t)e*kPoseSums + k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
      if (lane == 0) tot[k] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float* Tm = T + ((size_t)i*b + bi)*16;
      const float* Km = K + (size_t)bi*16;
      const float* Ki = Kinv + (size_t)bi*16;
      double R[9], t[3], K2[6], Ki3[9], M[9];
      for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) { R[r*3 + c] = Tm[r*4 + c]; Ki3[r*3 + c] = Ki[r*4 + c]; } t[r] = Tm[r*4 + 3]; }
      for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) K2[r*3 + c] = Km[r*4 + c];
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M[r*3 + c] = R[r*3]*Ki3[c] + R[r*3 + 1]*Ki3[3 + c] + R[r*3 + 2]*Ki3[6 + c];
      const double* gH = tot;          // 3x3
      const double ga[2] = {tot[9], tot[10]};
      const double gtz = tot[11];
      double gM[9];
      for (int m = 0; m < 3; ++m) for (int c = 0; c < 3; ++c) gM[m*3 + c] = K2[m]*gH[c] + K2[3 + m]*gH[3 + c];
      for (int c = 0; c < 3; ++c) gM[6 + c] += gH[6 + c];
      for (int r = 0; r < 2; ++r) for (int m = 0; m < 3; ++m)
        gK[r*3 + m] += gH[r*3]*M[m*3] + gH[r*3 + 1]*M[m*3 + 1] + gH[r*3 + 2]*M[m*3 + 2] + ga[r]*t[m];
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
        gKi[m*3 + c] += R[m]*gM[c] + R[3 + m]*gM[3 + c] + R[6 + m]*gM[6 + c];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (g_K) {
      float* o = g_K + (size_t)bi*16;
      for (int k = 0; k < 16; ++k) o[k] = 0.f;
      for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) o[r*4 + c] = (float)gK[r*3 + c];
    }
    if (g_Kinv) {
      float* o = g_Kinv + (size_t)bi*16;
      for (int k = 0; k < 16; ++k) o[k] = 0.f;
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o[r*4 + c] = (float)gKi[r*3 + c];
    }
  }
}

hipError_t launch_pose_finalize(const float* pose_partial, int entries, const float* T, const float* K, const float* Kinv,
                                float* g_T, float* g_K, float* g_Kinv, int b, int n, hipStream_t st) {
  hipLaunchKernelGGL(k_pose_finalize, dim3(b), dim3(256), 0, st, pose_partial, entries, T, K, Kinv, g_T, g_K, g_Kinv, b, n);
  return hipGetLastError();
}

}  // namespace smd
