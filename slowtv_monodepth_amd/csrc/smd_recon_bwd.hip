// smd_recon_bwd.hip — hand-written adjoint of the fused view-synthesis photometric loss (gfx950).
//
// Same wave-strip streaming structure as the forward (smd_recon_fwd.hip) with a three-stage software pipeline
// per row step j (60 interior columns, 2 halo lanes per side, rows r0-2 .. r1+1):
//   stage A (row j)   : re-synthesise the warped pixel x and its bilinear partials dx/dsx, dx/dsy from the four RGBX
//                       taps whose loads were issued one row earlier (software pipeline); issue row j+1's loads
//   stage B (row j-1) : window sums from the 3-row ring of raw values (vertical taps per lane, horizontal taps via DPP),
//                       SSIM partials d e/d(Sx, Sxx, Sxy) times the upstream gradient routed by `sel`
//                       (min-reprojection / automask), box-summed with the ADJOINT reflection weights
//                       (avg_pool2d + reflection_pad2d backward) -> complete for row j-2
//   stage C (row j-2) : dL/dx -> dL/d(sx, sy) (zero where the border clamp is active) -> projective chain rule
//                       -> dL/d depth (written once per pixel) and twelve per-lane sums dL/d(H, a) that a tiny
//                       epilogue kernel turns into dL/dT, dL/dK and dL/dK^-1.
// Nothing is re-read from HBM except the inputs themselves; no intermediate tensor of the forward is stored.
#include "smd_common.h"
#include "smd_kernels.h"

namespace smd {

// Reflection-weighted horizontal 3-tap through DPP wave shifts, as ONE asm block: keeps shift and FMA adjacent (left
// alone the compiler batches every shift of a row first and keeps ~50 results live).  s_nop covers the VALU-write ->
// DPP-read hazard the compiler cannot see inside asm.
__device__ __forceinline__ float hsum_w(float q, float wl, float wr) {
#ifdef SMD_NO_DPP
  return hsum3(q, wl, wr);
#else
  float r, t;
  asm volatile("s_nop 1\n\t"
               "v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "v_fma_f32 %0, %3, %1, %2\n\t"
               "v_mov_b32_dpp %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "v_fma_f32 %0, %4, %1, %0"
               : "=&v"(r), "=&v"(t) : "v"(q), "v"(wl), "v"(wr));
  return r;
#endif
}

// Three independent weighted sums in one block: one hazard nop, interleaved shifts and FMAs.
__device__ __forceinline__ void hsum_w3(float a, float b, float c, float wl, float wr, float& ra, float& rb, float& rc) {
#ifdef SMD_NO_DPP
  ra = hsum3(a, wl, wr); rb = hsum3(b, wl, wr); rc = hsum3(c, wl, wr);
#else
  // r = q + wl*left(q) + wr*right(q) as two DPP-sourced v_fmac per value (the shift rides on the FMA's first operand)
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

struct BwdPending {   // loads in flight for the next row
  f3 t[4];            // bilinear taps NW, NE, SW, SE (RGB texels)
  float y[3];
  float fx, fy, kx, ky;
};

// SKIP: 0 = every row does the full adjoint; 1 = rows where no pixel of the wave selected the current support skip the SSIM
// partials; 2 = additionally skip the chain rule of rows whose three coefficient rows and L1 term are all dead.  Coherent
// selection / automask regions (any partly trained network) make 2 the fastest (-20 % at the microbenchmark's poses);
// on noise-like masks (random initialisation) the branches cost ~5 %.
template <int SKIP>
__global__ __launch_bounds__(64*kWavesPerBlock) void k_recon_bwd(const ReconBwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nstrips = a.nsx*a.nsy;
  int strip, bi, s;
  decode_wave(blockIdx.x, wid, nstrips, a.b, a.S, strip, bi, s);
  if (strip >= nstrips) return;
  const int sxi = strip % a.nsx, syi = strip/a.nsx;
  const int h = a.h, w = a.w;
  const int c0 = sxi*kBwdCols;
  const int r0 = syi*a.rh, r1 = min(r0 + a.rh, h);

  const int u = c0 - 2 + lane;
  const bool col_ok = (u >= 0) && (u < w);
  const unsigned uc = (unsigned)min(max(u, 0), w - 1);
  const bool interior = (lane >= 2) && (lane < 2 + kBwdCols) && (u < w);
  float wl, wr, wla, wra;
  reflect_weights((int)uc, w, wl, wr);
  reflect_weights_adj((int)uc, w, wla, wra);
  if (!col_ok) { wl = wr = wla = wra = 0.f; }
  const float uf = (float)u;

  const bool use_min = a.flags & SMD_USE_MIN;
  const bool l1_only = a.flags & SMD_LOSS_L1;
  const unsigned hw = (unsigned)h*(unsigned)w;
  const float w_ssim = l1_only ? 0.f : kWSsim/3.f;
  const float w_l1 = l1_only ? 1.f/3.f : (1.f - kWSsim)/3.f;
  float gscale = a.g_loss[0]/((float)a.S*(float)a.b*(float)h*(float)w);
  if (!use_min) gscale /= (float)a.n;
  constexpr float c1 = 81.f*kC1, c2 = 81.f*kC2;   // window sums stay un-normalised (x9), see smd_recon_fwd.hip

  const float* tgt_b = a.tgt + (size_t)bi*3*hw;
  const float* depth_sb = a.depth + ((size_t)s*a.b + bi)*hw;
  const uint8_t* sel_sb = a.sel + ((size_t)s*a.b + bi)*hw;
  float* gd_sb = a.g_depth + ((size_t)s*a.b + bi)*hw;

  for (int i = 0; i < a.n; ++i) {
    Cam cm;
    make_cam(cm, a.T + ((size_t)i*a.b + bi)*16, a.K + (size_t)bi*16, a.Kinv + (size_t)bi*16);
    const unsigned wp = (unsigned)w + 1u;   // padded texel rows (smd_kernels.h: packed_texel_floats)
    const float* spk = a.supp_pk + ((size_t)i*a.b + bi)*3*(size_t)(h + 1)*wp;

    // rings of raw per-pixel values: index 0 = row j, 1 = row j-1, 2 = row j-2
    float x0[3] = {}, x1[3] = {}, x2[3] = {}, y0[3] = {}, y1[3] = {}, y2[3] = {};
    float gx0[3] = {}, gx1[3] = {}, gx2[3] = {}, gy0[3] = {}, gy1[3] = {}, gy2[3] = {};  // dx/dpx, dx/dpy (clamp mask, grid scale folded in)
    float ac1[3][3] = {}, ac0[3][3] = {};      // vertical accumulators of the h-summed coefficient maps {A, B, C}
    float psum[kPoseSums] = {};
    unsigned live_hist = 0;   // bit k: stage B of the k-th most recent row produced coefficients (wave-uniform)
    BwdPending P = {};

    auto issue = [&](int jr, float D) {   // stage 1 of row jr: coordinates + gathers
      const unsigned ro = (unsigned)jr*(unsigned)w + uc;
      const float vf = (float)jr;
      float hx = fmaf(cm.H[0], uf, fmaf(cm.H[1], vf, cm.H[2]));
      float hyy = fmaf(cm.H[3], uf, fmaf(cm.H[4], vf, cm.H[5]));
      float hz = fmaf(cm.H[6], uf, fmaf(cm.H[7], vf, cm.H[8]));
      float nx = fmaf(D, hx, cm.a0), ny = fmaf(D, hyy, cm.a1), yz = fmaf(D, hz, cm.tz);
      float rz = __builtin_amdgcn_rcpf(fmaxf(yz, kZMin));
      float sx = fmaf(nx*rz, a.wscale, -0.5f), sy = fmaf(ny*rz, a.hscale, -0.5f);
      Taps tp = make_taps(sx, sy, h, w, (int)wp);
      P.fx = tp.fx; P.fy = tp.fy; P.kx = tp.mx*a.wscale; P.ky = tp.my*a.hscale;
      const unsigned o = (unsigned)tp.off;
      P.t[0] = ld3(spk, o); P.t[1] = ld3(spk, o + 1u); P.t[2] = ld3(spk, o + wp); P.t[3] = ld3(spk, o + wp + 1u);
#pragma unroll
      for (int c = 0; c < 3; ++c) P.y[c] = ld1(tgt_b, c*hw + ro);
    };

    const int jstart = max(r0 - 2, 0);
    const int jlast = min(r1 + 1, h - 1);
    float Dn = ld1(depth_sb, (unsigned)jstart*(unsigned)w + uc);
    issue(jstart, Dn);
    if (jstart + 1 <= jlast) Dn = ld1(depth_sb, (unsigned)(jstart + 1)*(unsigned)w + uc);

    for (int j = jstart; j <= r1 + 1; ++j) {
      // ================= stage A: row j — consume its loads, issue the next row's =================
      if (j <= jlast) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float dn = P.t[1][c] - P.t[0][c], ds = P.t[3][c] - P.t[2][c];
          const float top = fmaf(P.fx, dn, P.t[0][c]), bot = fmaf(P.fx, ds, P.t[2][c]);
          const float ddy = bot - top;
          x0[c] = fmaf(P.fy, ddy, top);
          gx0[c] = fmaf(P.fy, ds - dn, dn)*P.kx;
          gy0[c] = ddy*P.ky;
          y0[c] = P.y[c];
        }
      }
      if (j + 1 <= jlast) {
        issue(j + 1, Dn);
        if (j + 2 <= jlast) Dn = ld1(depth_sb, (unsigned)(j + 2)*(unsigned)w + uc);
      }

      // ================= stage B: row p = j-1 — SSIM partials, h-summed with the adjoint weights =================
      const int p = j - 1;
      float hc[3][3] = {};
      if (!(!l1_only && p >= max(r0 - 1, 0) && p <= min(r1, h - 1))) live_hist <<= 1;
      if (!l1_only && p >= max(r0 - 1, 0) && p <= min(r1, h - 1)) {
        float lo_p, hi_p;
        reflect_weights(p, h, lo_p, hi_p);   // vertical reflection weights of rows p-1 (ring 2) and p+1 (ring 0)
        if (j >= h) hi_p = 0.f;              // row p+1 does not exist: ring 0 holds stale (finite) values
        const uint8_t sl = sel_sb[(unsigned)p*(unsigned)w + uc];
        const bool active = use_min ? (sl == (uint8_t)i) : (sl != (uint8_t)SMD_SEL_MASKED);
        const float g = (active && col_ok) ? gscale*w_ssim : 0.f;
        // Rows in which no pixel of this wave selected the current support carry no gradient through their windows:
        // skip the SSIM partials (wave-uniform branch; coherent regions of the min-reprojection / automask are common).
        const bool row_live = SKIP >= 1 ? (__builtin_amdgcn_ballot_w64(g != 0.f) != 0) : true;
        live_hist = (live_hist << 1) | (row_live ? 1u : 0u);
        if (row_live)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float ya = lo_p*y2[c], yc_ = hi_p*y0[c], xa = lo_p*x2[c], xc_ = hi_p*x0[c];
          float sy, syy, sx, sxx, sxy, unused;
          hsum_w3((xa + x1[c]) + xc_, fmaf(xc_, x0[c], fmaf(xa, x2[c], x1[c]*x1[c])), fmaf(xc_, y0[c], fmaf(xa, y2[c], x1[c]*y1[c])),
                  wl, wr, sx, sxx, sxy);
          hsum_w3((ya + y1[c]) + yc_, fmaf(yc_, y0[c], fmaf(ya, y2[c], y1[c]*y1[c])), 0.f, wl, wr, sy, syy, unused);
          // e = (1 - N/Dn)/2 with N = a1*a2, Dn = b1*b2 on the x9 sums (both scaled by 81*81)
          const float t = sx*sy;
          const float a1 = fmaf(2.f, t, c1), a2 = fmaf(2.f, fmaf(9.f, sxy, -t), c2);
          const float sx2 = sx*sx;
          const float b1 = sx2 + fmaf(sy, sy, c1), b2 = fmaf(9.f, sxx, -sx2) + (fmaf(9.f, syy, c2) - sy*sy);
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

      // ================= stage C: row q = j-2 — dL/dx -> dL/d(px,py) -> depth, pose sums =================
      const int q = j - 2;
      if (q >= r0 && q < r1) {
        float lo_q, hi_q;
        reflect_weights_adj(q, h, lo_q, hi_q);
        const unsigned rq = (unsigned)q*(unsigned)w + uc;
        const uint8_t sl = sel_sb[rq];
        const bool active = use_min ? (sl == (uint8_t)i) : (sl != (uint8_t)SMD_SEL_MASKED);
        const float gl = (active && col_ok) ? gscale*w_l1 : 0.f;
        // coefficient rows q-1, q, q+1 all skipped and no L1 term anywhere in the wave: the gradient of this row is zero
        const bool dead = SKIP >= 2 ? ((live_hist & 7u) == 0u && __builtin_amdgcn_ballot_w64(gl != 0.f) == 0) : false;
        if (dead) {
          if (interior && i == 0) gd_sb[rq] = 0.f;
        } else {
        float gpx = 0.f, gpy = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float d = x2[c] - y2[c];
          float gxc = gl*((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f));
          if (!l1_only) {
            const float SA = fmaf(hi_q, hc[c][0], ac1[c][0]), SB = fmaf(hi_q, hc[c][1], ac1[c][1]), SC = fmaf(hi_q, hc[c][2], ac1[c][2]);
            gxc += fmaf(2.f*x2[c], SB, fmaf(y2[c], SC, SA));   // d/dx_q of the x9 sums: 1, 2 x_q, y_q
          }
          gpx = fmaf(gxc, gx2[c], gpx);
          gpy = fmaf(gxc, gy2[c], gpy);
        }
        // projective chain rule at (q, u): recompute the cheap geometry
        const float D = ld1(depth_sb, rq);
        const float vf = (float)q;
        float hx = fmaf(cm.H[0], uf, fmaf(cm.H[1], vf, cm.H[2]));
        float hyy = fmaf(cm.H[3], uf, fmaf(cm.H[4], vf, cm.H[5]));
        float hz = fmaf(cm.H[6], uf, fmaf(cm.H[7], vf, cm.H[8]));
        float nx = fmaf(D, hx, cm.a0), ny = fmaf(D, hyy, cm.a1), yz = fmaf(D, hz, cm.tz);
        float rz = __builtin_amdgcn_rcpf(fmaxf(yz, kZMin));
        float gnx = gpx*rz, gny = gpy*rz;
        float gz = (yz >= kZMin) ? -(gpx*nx + gpy*ny)*rz*rz : 0.f;
        if (!interior) { gnx = 0.f; gny = 0.f; gz = 0.f; }
        float gD = fmaf(gnx, hx, fmaf(gny, hyy, gz*hz));
        if (interior) {
          float* gp = gd_sb + rq;
          if (i == 0) *gp = gD; else *gp += gD;
        }
        float dnx = gnx*D, dny = gny*D, dz = gz*D;
        psum[0] = fmaf(dnx, uf, psum[0]); psum[1] = fmaf(dnx, vf, psum[1]); psum[2] += dnx;
        psum[3] = fmaf(dny, uf, psum[3]); psum[4] = fmaf(dny, vf, psum[4]); psum[5] += dny;
        psum[6] = fmaf(dz, uf, psum[6]);  psum[7] = fmaf(dz, vf, psum[7]);  psum[8] += dz;
        psum[9] += gnx; psum[10] += gny; psum[11] += gz;
        }
      }

      // ================= roll =================
      if (!l1_only) {
        float lo_na, hi_na;
        reflect_weights_adj(min(max(p + 1, 0), h - 1), h, lo_na, hi_na);  // weight of coefficient row p in out(p+1)
        if (p + 1 >= h || p < 0) lo_na = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int k = 0; k < 3; ++k) { ac1[c][k] = ac0[c][k] + hc[c][k]; ac0[c][k] = lo_na*hc[c][k]; }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        x2[c] = x1[c]; x1[c] = x0[c]; y2[c] = y1[c]; y1[c] = y0[c];
        gx2[c] = gx1[c]; gx1[c] = gx0[c]; gy2[c] = gy1[c]; gy1[c] = gy0[c];
      }
    }

    // per-wave pose partials
    float* pp = a.pose_partial + (((size_t)i*a.b + bi)*((size_t)a.S*nstrips) + (size_t)s*nstrips + strip)*kPoseSums;
#pragma unroll
    for (int k = 0; k < kPoseSums; ++k) {
      float tot = wave_sum(psum[k]);
      if (lane == 0) pp[k] = tot;
    }
  }
}

hipError_t launch_recon_bwd(const ReconBwdArgs& a, hipStream_t st) {
  dim3 grid(recon_grid_blocks(a.nsx*a.nsy, a.b, a.S)), block(64*kWavesPerBlock);
  if (a.skip_level >= 2) hipLaunchKernelGGL(k_recon_bwd<2>, grid, block, 0, st, a);
  else if (a.skip_level == 1) hipLaunchKernelGGL(k_recon_bwd<1>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(k_recon_bwd<0>, grid, block, 0, st, a);
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
