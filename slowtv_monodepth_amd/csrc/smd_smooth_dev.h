// smd_smooth_dev.h — device-side bodies of the smoothness sweep and its streaming adjoint, shared between the kernels of smd_smooth.hip and the
// fused loss path, where they run as guest blocks of the reconstruction launches (smd_recon_fwd.hip, smd_depth.hip).
#pragma once
#include "smd_common.h"
#include "smd_kernels.h"

namespace smd {

// Offset (in float2 units) of scale s in the per-pixel edge-weight buffer {exp(-|dI/dx|), exp(-|dI/dy|)}.
__host__ __device__ inline size_t edge_offset(const ScaleSet& sc, int b, int s) {
  size_t off = 0;
  for (int k = 0; k < s; ++k) off += (size_t)b*sc.hs[k]*sc.ws[k];
  return off;
}

// Offset (bytes) of the arrival counters behind the edge weights: [S*b] pairs + 1 (smd_disp_smooth_edge_weight_bytes).
__host__ __device__ inline size_t edge_arrive_offset(const ScaleSet& sc, int b) { return (edge_offset(sc, b, sc.S)*sizeof(float2) + 255) & ~(size_t)255; }

// total = w_rec*l_rec + w_sm*l_sm of the fused loss path (src/core/trainer.py:462-464 with the frozen `weights`), formed by whichever of the
// two final reducers — the reconstruction's last block, the smoothness sweep's last pair — arrives second.  Called by ONE lane.
// out3 = {total, l_rec, l_sm}; which: 1 = reconstruction, 2 = smoothness.  fp32, the eager expression's roundings: fl(fl(w_rec l_rec) + fl(w_sm l_sm)).
inline __device__ void loss_combine_arrive(const LossCombine& c, int which, float value) {
  __hip_atomic_store((unsigned*)c.out3 + which, __builtin_bit_cast(unsigned, value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (__hip_atomic_fetch_add(c.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) return;
  const float other = __builtin_bit_cast(float, __hip_atomic_load((const unsigned*)c.out3 + (3 - which), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  const float l_rec = which == 1 ? value : other, l_sm = which == 1 ? other : value;
  c.out3[0] = __fadd_rn(__fmul_rn(c.w_rec, l_rec), __fmul_rn(c.w_sm, l_sm));
  __hip_atomic_store(c.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch on this buffer
}

// The sweep over the disparities: per wave, partial sums of the UN-normalised edge energy E' = sum w |d_p - d_q| and of d itself.
// Because dhat = d / m with one m > 0 per image, E = E' / m: the mean is not needed inside the pixel loop, so the reference's
// mean pass and its stencil pass collapse into one sweep.  A wave owns 63 columns + one halo lane and kSmoothRows rows: one
// disparity load and one 8-byte weight load per row, all in flight together; `edge_w` null = no edge weighting (weights 1).
//
// `arrive` != null: the second stage runs inside this launch (round 3; the former k_smooth_finalize launch) as a chain of wave-level
// hand-offs without a block barrier (cdna_hip_programming.md, Guideline 16): a wave publishes its partial write-through and
// drains; the LAST wave of a block counts the block's arrival for its (scale, sample) pair; the wave that completes a pair
// reduces it to (mean, E) -> stats and publishes the pair's share of the loss; the wave that completes the last pair adds
// the shares up in a fixed order.  fp64, deterministic whichever waves end up doing it.
//   arrive[0 .. S*b): arrival counters of the pairs, arrive[S*b]: pairs done (zeroed by k_smooth_edges, reset by their last arrivers)
//   contrib: [S*b] doubles behind the partials in the workspace
//
// A device function since round 5: it runs as the kernel k_smooth_main (smd_smooth.hip) or as GUEST blocks at the end of k_recon_main's grid
// (the fused loss path, smd_recon_fwd.hip) — there its 7 us of memory-bound work and its reduction chain hide in the drain of the big
// launch.  `blk`: index among the sweep's blocks; 256 threads.
inline __device__ void smooth_main_block(const ScaleSet& sc, int b, const SmoothFwdJob& jb, int blk) {
  const float* __restrict__ edge_w = jb.edge_w;
  float* partial = jb.partial; const int max_units = jb.max_units;
  float* stats = jb.stats; float* loss = jb.loss; unsigned* arrive = jb.arrive; double* contrib = jb.contrib;
  // exactly the blocks that have work: for s = S-1 .. 0 (coarse scales first), for each sample, ceil(units_s / 4) blocks
  int s = sc.S - 1;
  for (; s > 0; --s) { const int nb = ceil_div(smooth_units_main(sc.hs[s], sc.ws[s]), 4)*b; if (blk < nb) break; blk -= nb; }
  const int hs = sc.hs[s], ws = sc.ws[s], n = hs*ws;
  const int units = smooth_units_main(hs, ws), bpi = ceil_div(units, 4);
  const int bi = blk/bpi, bx = blk - bi*bpi;
  const int lane = threadIdx.x & 63, unit = bx*4 + (threadIdx.x >> 6);
  __shared__ unsigned waves_done;
  if (arrive != nullptr) {            // the only block barrier: at the start, where every wave still is
    if (threadIdx.x == 0) waves_done = 0u;
    __syncthreads();
  }
  if (unit >= units) return;
  {
  const int nsx = (ws + kSmoothCols - 1)/kSmoothCols;
  const int sxi = unit % nsx, syi = unit/nsx;
  const int r0 = syi*kSmoothRowsMain, r1 = min(r0 + kSmoothRowsMain, hs);
  const int u = sxi*kSmoothCols + lane, uc = min(u, ws - 1);        // lanes right of the image repeat its last column: |d - d| = 0
  const bool live = lane < kSmoothCols && u < ws;
  const float* __restrict__ d = sc.p[s] + (size_t)bi*n;
  const float2* __restrict__ ew = edge_w ? (const float2*)edge_w + edge_offset(sc, b, s) + (size_t)bi*n : nullptr;
  float dr_[kSmoothRowsMain + 1];
  float2 wr_[kSmoothRowsMain];
#pragma unroll
  for (int k = 0; k <= kSmoothRowsMain; ++k) dr_[k] = d[(size_t)min(r0 + k, hs - 1)*ws + uc];   // the last image row pairs with itself
#pragma unroll
  for (int k = 0; k < kSmoothRowsMain; ++k) wr_[k] = ew ? ew[(size_t)min(r0 + k, hs - 1)*ws + uc] : make_float2(1.f, 1.f);
  float accE = 0.f, accD = 0.f;
#pragma unroll
  for (int k = 0; k < kSmoothRowsMain; ++k) {
    const float cur = dr_[k], right = lane_right(cur);
    if (live && r0 + k < r1) {
      accD += cur;
      accE += fabsf(cur - right)*wr_[k].x + fabsf(cur - dr_[k + 1])*wr_[k].y;
    }
  }
  const float totE = wave_sum(accE), totD = wave_sum(accD);
  if (lane == 0) {   // published write-through (agent scope): the block that arrives last reads every partial in this launch
    unsigned long long* pp = (unsigned long long*)partial + ((size_t)s*b + bi)*max_units + unit;
    __hip_atomic_store(pp, ((unsigned long long)__builtin_bit_cast(unsigned, totD) << 32) | __builtin_bit_cast(unsigned, totE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  }
  if (arrive == nullptr) return;      // two-launch form: k_smooth_finalize follows
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int pair = s*b + bi, npairs = sc.S*b;
  const unsigned live = (unsigned)min(4, units - bx*4);
  unsigned flag = 0;
  if (lane == 0) {
    if (__hip_atomic_fetch_add(&waves_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == live - 1u)
      flag = (__hip_atomic_fetch_add(arrive + pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)ceil_div(units, 4) - 1u) ? 1u : 0u;
  }
  if (!__builtin_amdgcn_readfirstlane((int)flag)) return;
  SMD_TAIL_ACQUIRE();
  {  // this wave completed the pair: (mean, E) and the pair's share of the loss
    double e = 0.0, dsum = 0.0;
    const unsigned long long* pp = (const unsigned long long*)partial + (size_t)pair*max_units;
    auto ldp = [&](int c, double& ee, double& dd) {
      const unsigned long long v = __hip_atomic_load(pp + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ee += (double)__builtin_bit_cast(float, (unsigned)v); dd += (double)__builtin_bit_cast(float, (unsigned)(v >> 32));
    };
    int c = lane;
    for (; c + 192 < units; c += 256) {   // four independent loads in flight; the order of the additions is fixed
      double e0 = 0, d0 = 0, e1 = 0, d1 = 0, e2 = 0, d2 = 0, e3 = 0, d3 = 0;
      ldp(c, e0, d0); ldp(c + 64, e1, d1); ldp(c + 128, e2, d2); ldp(c + 192, e3, d3);
      e += (e0 + e1) + (e2 + e3); dsum += (d0 + d1) + (d2 + d3);
    }
    for (; c < units; c += 64) ldp(c, e, dsum);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { e += __shfl_xor(e, off, 64); dsum += __shfl_xor(dsum, off, 64); }
    const float mean = (float)(dsum/n);
    const float E = (float)(e/(double)fmaxf(mean, kEps32));
    flag = 0;
    if (lane == 0) {
      stats[(size_t)pair*2] = mean; stats[(size_t)pair*2 + 1] = E;
      const double share = ldexp((double)E/((double)b*n), -sc.key[s]);   // 2^-key exactly, without the double-precision exp2 routine
      __hip_atomic_store((unsigned long long*)contrib + pair, __builtin_bit_cast(unsigned long long, share), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(arrive + pair, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the slot is free again
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      flag = (__hip_atomic_fetch_add(arrive + npairs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)npairs - 1u) ? 1u : 0u;
    }
  }
  if (!__builtin_amdgcn_readfirstlane((int)flag)) return;
  SMD_TAIL_ACQUIRE();
  double total = 0.0;
  for (int q = lane; q < npairs; q += 64) total += __builtin_bit_cast(double, __hip_atomic_load((const unsigned long long*)contrib + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
  if (lane == 0) {
    const float l_sm = (float)(total/sc.S);
    loss[0] = l_sm;
    __hip_atomic_store(arrive + npairs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (jb.comb.out3) loss_combine_arrive(jb.comb, 2, l_sm);
  }
}

inline int smooth_main_blocks(const ScaleSet& sc, int b) {
  int nb = 0;
  for (int s = 0; s < sc.S; ++s) nb += ceil_div(smooth_units_main(sc.hs[s], sc.ws[s]), 4)*b;
  return nb;
}


// Streaming adjoint (round 4): a wave owns 62 columns (+ one halo lane on each side) and kSmoothRowsMain rows.  With
//   t_x(v,u) = w_x(v,u) sg(dhat(v,u) - dhat(v,u+1))  (0 in the last column),   t_y(v,u) = w_y(v,u) sg(dhat(v,u) - dhat(v+1,u))  (0 in the last row)
// the gradient of E' is G(v,u) = t_x(v,u) - t_x(v,u-1) + t_y(v,u) - t_y(v-1,u): per row ONE disparity load and ONE 8-byte weight load (all of a
// unit's rows requested up front), the horizontal neighbours by DPP, the vertical ones from the previous / next row's registers — the
// per-pixel form gathered 5 disparities and 3 weight pairs per pixel (12.7 -> ~9 us at cfg 2).  Same values: sg() of the same differences.
constexpr int kSmoothBwdCols = 62;
__host__ __device__ inline int smooth_units_bwd(int hs, int ws) { return ((ws + kSmoothBwdCols - 1)/kSmoothBwdCols)*((hs + kSmoothRowsMain - 1)/kSmoothRowsMain); }

// A device function since round 5: block `bx` of (scale s, sample bi); the kernel k_smooth_bwd_stream (smd_smooth.hip) or guest blocks of the K0
// adjoint's first launch (the fused loss path, smd_depth.hip).  g_scale: host-side factor of the incoming gradient (the loss weight);
// accumulate: add to what the level's gradient tensor already holds (a level the reconstruction backward wrote in an earlier launch).
inline __device__ void smooth_bwd_block(const ScaleSet& sc, int b, int s, int bi, int bx, const float* __restrict__ stats, const float* __restrict__ g_loss,
                                        float g_scale, const float* __restrict__ edge_w, bool accumulate) {
  const int hs = sc.hs[s], ws = sc.ws[s], n = hs*ws;
  const int units = smooth_units_bwd(hs, ws);
  const int lane = threadIdx.x & 63, unit = bx*4 + (threadIdx.x >> 6);
  if (unit >= units) return;
  const int nsx = (ws + kSmoothBwdCols - 1)/kSmoothBwdCols;
  const int sxi = unit % nsx, syi = unit/nsx;
  const int r0 = syi*kSmoothRowsMain, r1 = min(r0 + kSmoothRowsMain, hs);
  const int u = sxi*kSmoothBwdCols - 1 + lane, uc = min(max(u, 0), ws - 1);
  const bool store = lane >= 1 && lane <= kSmoothBwdCols && u < ws;
  const float* __restrict__ d = sc.p[s] + (size_t)bi*n;
  float* __restrict__ gd = sc.g[s] + (size_t)bi*n;
  const float2* __restrict__ ew = edge_w ? (const float2*)edge_w + edge_offset(sc, b, s) + (size_t)bi*n : nullptr;
  const float mean = stats[((size_t)s*b + bi)*2], E = stats[((size_t)s*b + bi)*2 + 1];
  const float m = fmaxf(mean, kEps32), inv_m = 1.f/m;
  const float gl = g_loss[0]*g_scale;
  const float gs = gl*exp2f(-(float)sc.key[s])/((float)sc.S*(float)b*(float)n);
  const float mean_term = (mean >= kEps32) ? E*inv_m/(float)n : 0.f;
  auto sg = [](float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); };
  // rows r0-1 .. r0+R (clamped into the image; out-of-image terms are masked below)
  float dr_[kSmoothRowsMain + 2];
  float2 wr_[kSmoothRowsMain + 1];
#pragma unroll
  for (int k = 0; k < kSmoothRowsMain + 2; ++k) dr_[k] = d[(size_t)min(max(r0 - 1 + k, 0), hs - 1)*ws + uc]*inv_m;
#pragma unroll
  for (int k = 0; k < kSmoothRowsMain + 1; ++k) wr_[k] = ew ? ew[(size_t)min(max(r0 - 1 + k, 0), hs - 1)*ws + uc] : make_float2(1.f, 1.f);
  const bool has_right = u >= 0 && u < ws - 1;           // t_x exists for columns 0 .. ws-2 (a halo lane left of the image holds none)
  // t_y of the row above the unit
  float ty_prev = (r0 > 0) ? wr_[0].y*sg(dr_[0] - dr_[1]) : 0.f;
#pragma unroll
  for (int k = 0; k < kSmoothRowsMain; ++k) {
    const int v = r0 + k;
    const float dc = dr_[k + 1], right = lane_right(dc);
    const float tx = has_right ? wr_[k + 1].x*sg(dc - right) : 0.f;
    const float tx_l = lane_left(tx);
    const float ty = (v < hs - 1) ? wr_[k + 1].y*sg(dc - dr_[k + 2]) : 0.f;
    const float G = (tx - tx_l) + (ty - ty_prev);
    ty_prev = ty;
    if (store && v < r1) { const float val = gs*(G*inv_m - mean_term); gd[(size_t)v*ws + u] = accumulate ? gd[(size_t)v*ws + u] + val : val; }
  }
}

// blocks per sample of the streaming adjoint over the whole pyramid, and the (scale, block) of per-sample block index q (coarse scales first)
inline __host__ __device__ int smooth_bwd_blocks_per_sample(const ScaleSet& sc) {
  int nb = 0;
  for (int s = 0; s < sc.S; ++s) nb += ceil_div(smooth_units_bwd(sc.hs[s], sc.ws[s]), 4);
  return nb;
}
inline __device__ void smooth_bwd_decode(const ScaleSet& sc, int q, int& s, int& bx) {
  s = sc.S - 1;
  for (; s > 0; --s) { const int nb = ceil_div(smooth_units_bwd(sc.hs[s], sc.ws[s]), 4); if (q < nb) break; q -= nb; }
  bx = q;
}


}  // namespace smd
