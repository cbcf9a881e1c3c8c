// smd_pose_fin.h — the per-sample epilogue of the fused backward (shared by smd_recon_bwd.hip, which runs it in-launch or as
// a launch of its own, and smd_depth.hip, whose K0-adjoint launch carries it as extra blocks on the training path).
#pragma once
#include "smd_common.h"
#include "smd_kernels.h"

namespace smd {

// ---------------------------------------------------------------------------------------------
// Epilogue of a sample (the former k_pose_finalize launch), run by ONE wave: sum the per-block partials of dL/d(H, a0, a1, tz)
// and push them through
//   H[0:2] = K2 * M,  H[2] = M[2],  M = R * Ki3,  (a0, a1) = K2 * t,  tz = t[2]
// to dL/dT (n,b,4,4), dL/dK (b,4,4), dL/dKinv (b,4,4).  fp64, fixed order -> deterministic whichever wave does it.
// The wave is a chain of latencies (it runs alone: at the end of a sample in-launch, or as a guest block of the K0 adjoint),
// so it is built around the number of round trips, not throughput:
//   * the partials of a (support, sample) are `entries` x 12 contiguous floats, swept with 16-byte agent-scope loads, sixteen per
//     lane issued before the first is used (one trip per 16 KB = 341 entries); lane l owns floats 4l .. 4l+3 of every 256-float
//     chunk, and as 12 does not divide 256 the sum its j-th float belongs to rotates with the chunk: k = (4 (c + l) + j) mod 12
//     — three accumulator sets (c mod 3);
//   * the 64 x 12 lane accumulators go through LDS: lane k < 12 adds up its column over the lanes in lane order (64 ds_read_b64,
//     ~0.3 us) instead of 72 dependent cross-lane exchanges of doubles.
// scratch: kFinScratchDoubles doubles of LDS owned by this wave.
// ---------------------------------------------------------------------------------------------
constexpr int kFinDump = 64*kPoseSums;                                   // lane accumulators of ONE wave, [lane][set][component]
constexpr int kFinTot = SMD_MAX_SUPPORTS*kPoseSums;                      // totals, [support][12]
constexpr int kFinShared = kFinTot + SMD_MAX_SUPPORTS*(9 + 9 + 6 + 9);   // + per support: M, gM, 6 of dL/dK, 9 of dL/dKinv
constexpr int kFinScratchDoubles = kFinDump + kFinShared;                // scratch of a single-wave epilogue
__host__ __device__ constexpr int fin_scratch_doubles(int nwaves) { return nwaves*kFinDump + kFinShared; }

__device__ __forceinline__ f4 ld4_agent(rsrc_t r, unsigned byte_off) {   // 16 bytes another workgroup published in this launch (sc1: agent scope)
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}
__device__ __forceinline__ void wave_lds_sync() {   // LDS written by some lanes of THIS wave is read by others: program order + a wait
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// MULTI: every wave of the block calls this (wave `wv` of `nwaves` sweeps supports wv, wv + nwaves, ...; block barriers);
// otherwise ONE wave calls it with wv = 0, nwaves = 1.  scratch: fin_scratch_doubles(nwaves) doubles of LDS.
// lds_out (optional, LDS): wave 0 also leaves dL/dT [n][16], dL/dK [16], dL/dK^-1 [16] there for a chain rule that continues in the same wave
// (the fused loss path: smd_depth.hip).
// Written to stay SMALL in registers (it is also guest code inside the K0 adjoint, whose occupancy it must not cost): the
// sweep keeps four 16-byte loads in flight, and the 3x3 algebra is spread over the lanes one output element each, with the
// intermediate matrices in LDS, instead of one lane holding ~50 doubles.
template <bool MULTI>
inline __device__ void pose_finalize(const ReconBwdArgs& a, int bi, int entries, double* scratch, int wv, int nwaves, float* lds_out = nullptr) {
  const int lane = threadIdx.x & 63;
  const int n = a.n, b = a.b;
  const unsigned F4 = (unsigned)entries*kPoseSums*4u;        // bytes per (support, sample); a multiple of 16
  double* dump = scratch + wv*kFinDump;
  double* tot = scratch + nwaves*kFinDump;                   // [n][12]: gH (3x3), ga (2), gtz
  double* Ms = tot + kFinTot;                                // [n][9]  M = R Ki3
  double* gMs = Ms + SMD_MAX_SUPPORTS*9;                     // [n][9]
  double* gKs = gMs + SMD_MAX_SUPPORTS*9;                    // [n][6]
  double* gKis = gKs + SMD_MAX_SUPPORTS*6;                   // [n][9]
  for (int i = wv; i < n && wv < nwaves; i += nwaves) {   // (waves beyond `nwaves` of a MULTI block only take part in the barrier)
    const float* pp = a.pose_partial + ((size_t)i*b + bi)*(size_t)a.pose_stride*kPoseSums;
    const rsrc_t rs = make_rsrc(pp, F4);                     // loads beyond the last entry read 0
    double acc[3][4];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[m][j] = 0.0;
    for (unsigned base = 0; base < F4; base += 12u*1024u) {  // 12 chunks per outer trip keep the set index (chunk mod 3) static
#pragma unroll
      for (int part = 0; part < 3; ++part) {
        const unsigned o = base + (unsigned)part*4u*1024u;
        if (o >= F4) break;
        f4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ld4_agent(rs, o + (unsigned)q*1024u + (unsigned)lane*16u);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[(part*4 + q) % 3][j] += (double)v[q][j];
      }
    }
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) dump[lane*kPoseSums + m*4 + j] = acc[m][j];
    wave_lds_sync();
    if (lane < kPoseSums) {   // sum k = lane: in lane l it sits in set (k/4 - l) mod 3, component k mod 4
      const int k = lane;
      double t = 0.0;
#pragma unroll 8
      for (int l = 0; l < 64; ++l) t += dump[l*kPoseSums + ((k/4 + 3 - l % 3) % 3)*4 + (k & 3)];
      tot[i*kPoseSums + k] = t;
    }
    wave_lds_sync();
  }
  if (MULTI) __syncthreads();
  if (wv != 0) return;
  // ---- the chain rule, one output element per lane: lane = 16*(support in the group of four) + e
  const float* Km = a.K + (size_t)bi*16;
  const float* Ki = a.Kinv + (size_t)bi*16;
  const int e = lane & 15;
  for (int i0 = 0; i0 < n; i0 += 4) {
    const int i = i0 + (lane >> 4);
    const bool on = i < n;
    const float* Tm = a.T + ((size_t)(on ? i : 0)*b + bi)*16;
    const double* gH = tot + (on ? i : 0)*kPoseSums;
    if (on && e < 9) {
      const int r = e/3, c = e - r*3;
      double m = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) m += (double)Tm[r*4 + k]*(double)Ki[k*4 + c];      // M = R Ki3
      Ms[i*9 + e] = m;
      double gm = (double)Km[r]*gH[c] + (double)Km[4 + r]*gH[3 + c];                 // gM[m][c], m = r: K2[0][m] gH[0][c] + K2[1][m] gH[1][c]
      if (r == 2) gm += gH[6 + c];
      gMs[i*9 + e] = gm;
    }
    wave_lds_sync();
    if (on) {
      const int r = e >> 2, m = e & 3;
      double gt = 0.0;                                                                // dL/dT[r][m]
      if (r < 3 && m < 3) {
#pragma unroll
        for (int q = 0; q < 3; ++q) gt += gMs[i*9 + r*3 + q]*(double)Ki[m*4 + q];
      } else if (r < 3) {
        gt = (double)Km[r]*gH[9] + (double)Km[4 + r]*gH[10];                          // K2[:, r] . ga
        if (r == 2) gt += gH[11];
      }
      a.g_T[((size_t)i*b + bi)*16 + e] = (float)gt;
      if (lds_out) lds_out[i*16 + e] = (float)gt;
      if (e < 6) {                                                                    // dL/dK[r2][m2] share of this support
        const int r2 = e/3, m2 = e - r2*3;
        double v = gH[9 + r2]*(double)Tm[m2*4 + 3];
#pragma unroll
        for (int q = 0; q < 3; ++q) v += gH[r2*3 + q]*Ms[i*9 + m2*3 + q];
        gKs[i*6 + e] = v;
      }
      if (e < 9) {                                                                    // dL/dKinv[m3][c3] share
        const int m3 = e/3, c3 = e - m3*3;
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < 3; ++q) v += (double)Tm[q*4 + m3]*gMs[i*9 + q*3 + c3];
        gKis[i*9 + e] = v;
      }
    }
    wave_lds_sync();
  }
  if (lane < 16 && (a.g_K || a.g_Kinv)) {   // lane q writes element q of the two 4x4 outputs: the supports' shares added in index order
    const int r = lane >> 2, c = lane & 3;
    double sK = 0.0, sKi = 0.0;
    for (int i = 0; i < n; ++i) {
      if (r < 2 && c < 3) sK += gKs[i*6 + r*3 + c];
      if (r < 3 && c < 3) sKi += gKis[i*9 + r*3 + c];
    }
    if (a.g_K) a.g_K[(size_t)bi*16 + lane] = (float)sK;
    if (a.g_Kinv) a.g_Kinv[(size_t)bi*16 + lane] = (float)sKi;
    if (lds_out) { lds_out[n*16 + lane] = (float)sK; lds_out[n*16 + 16 + lane] = (float)sKi; }
  }
}

inline __device__ void pose_finalize_wave(const ReconBwdArgs& a, int bi, int entries, double* scratch) { pose_finalize<false>(a, bi, entries, scratch, 0, 1); }

}  // namespace smd
