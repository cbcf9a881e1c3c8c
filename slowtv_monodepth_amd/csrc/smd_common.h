// smd_common.h — device-side building blocks shared by the gfx950 kernels of the view-synthesis loss path.
//
// Written for CDNA4 only: wave64, DPP wave shifts for the horizontal stencil taps, SGPR-resident camera
// constants, unaligned 8-byte gathers from the planar (NCHW) support frames.  No LDS is needed by the fused
// kernels: one wave owns a 64-column strip and streams down the rows, so the 3x3 windows live in registers
// (horizontal taps = neighbouring lanes, vertical taps = forward-accumulated row sums).
#pragma once
// Experiment / diagnosis switches (ablations, wave traces, dropped kernel variants, launch-shape knobs read from the environment) exist
// only in -DSMD_EXPERIMENTS builds (`make EXPERIMENTS=1`; scripts/dev builds those).  The product library ignores them.
#ifndef SMD_EXPERIMENTS
#undef SMD_ABLATE
#undef SMD_ABLATE_BWD
#undef SMD_FWD_PRIO
#undef SMD_BWD_FASTPATH
#undef SMD_NO_DPP
#undef SMD_BWD_PEEL
#undef SMD_BWD_ROWSEL
#undef SMD_BWD_STATEFUL
#undef SMD_FWD_CAM_REGS
#undef SMD_TRACE_WAVES
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <hip/hip_bf16.h>

namespace smd {

// Typed element access for kernels whose I/O tensors may be bf16 at an autocast boundary (arithmetic stays fp32).
typedef __hip_bfloat16 bf16;
template <typename T> __device__ __forceinline__ float ld_as_float(const T* p, size_t i);
template <> __device__ __forceinline__ float ld_as_float<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld_as_float<bf16>(const bf16* p, size_t i) { return __bfloat162float(p[i]); }
template <typename T> __device__ __forceinline__ void st_from_float(T* p, size_t i, float v);
template <> __device__ __forceinline__ void st_from_float<float>(float* p, size_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st_from_float<bf16>(bf16* p, size_t i, float v) { p[i] = __float2bfloat16(v); }


constexpr float kEps32 = 1.1920929e-07f;  // torch.finfo(float32).eps  (src/tools/ops.py:63-66)
constexpr float kC1 = 1e-4f;              // SSIM eps1 = 0.01^2        (src/losses/photometric.py:30)
constexpr float kC2 = 9e-4f;              // SSIM eps2 = 0.03^2        (src/losses/photometric.py:31)
constexpr float kWSsim = 0.85f;           // PhotoError(weight_ssim)   (src/losses/reconstruction.py:38)
constexpr float kZMin = 0.1f;             // z.clamp(min=0.1)          (src/tools/geometry.py:341)
constexpr int kWave = 64;

typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));  // 8-byte load that is only 4-byte aligned
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f3 __attribute__((ext_vector_type(3)));               // one RGB texel of a repacked support frame

// ---------------------------------------------------------------------------------------------
// Cross-lane neighbours (DPP wave shifts; one VALU op each, no LDS traffic).
//   lane_left(x)[l]  = x[l-1]  (0 for lane 0)
//   lane_right(x)[l] = x[l+1]  (0 for lane 63)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float lane_left(float x) {
#ifdef SMD_NO_DPP
  float v = __shfl_up(x, 1, 64);
  return (threadIdx.x & 63) == 0 ? 0.f : v;
#else
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138 /*wave_shr:1*/, 0xf, 0xf, true));
#endif
}
__device__ __forceinline__ float lane_right(float x) {
#ifdef SMD_NO_DPP
  float v = __shfl_down(x, 1, 64);
  return (threadIdx.x & 63) == 63 ? 0.f : v;
#else
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130 /*wave_shl:1*/, 0xf, 0xf, true));
#endif
}

// Weighted horizontal 3-tap: q + wl*q[l-1] + wr*q[l+1].  With (wl, wr) = reflect_weights() this is the
// ReflectionPad2d(1) + 3-wide box sum of src/losses/photometric.py:27-28 along x.
__device__ __forceinline__ float hsum3(float q, float wl, float wr) {
  return fmaf(wr, lane_right(q), fmaf(wl, lane_left(q), q));
}

// Reflection-padded 3-tap weights for position i in [0, n): neighbour i-1 has weight `lo`, i+1 has `hi`.
//   i == 0   : window {-1->1, 0, 1}     -> lo 0, hi 2
//   i == n-1 : window {n-2, n-1, n->n-2} -> lo 2, hi 0
__device__ __forceinline__ void reflect_weights(int i, int n, float& lo, float& hi) {
  lo = (i == 0) ? 0.f : ((i == n - 1) ? 2.f : 1.f);
  hi = (i == n - 1) ? 0.f : ((i == 0) ? 2.f : 1.f);
}
// Adjoint of the above (weights with which position i RECEIVES from coefficient maps at i-1 / i+1):
//   from i-1: reflect hi-weight of (i-1) = 2 if i-1 == 0;  from i+1: reflect lo-weight of (i+1) = 2 if i+1 == n-1.
__device__ __forceinline__ void reflect_weights_adj(int i, int n, float& lo, float& hi) {
  lo = (i == 0) ? 0.f : ((i == 1) ? 2.f : 1.f);
  hi = (i == n - 1) ? 0.f : ((i == n - 2) ? 2.f : 1.f);
  if (n == 2) { lo = (i == 1) ? 2.f : 0.f; hi = (i == 0) ? 2.f : 0.f; }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------
// Per-(support, sample) camera constants, folded so that a pixel costs 6 FMAs + 1 rcp + 2 mul:
//   [hx hy hz]^T = H * (u, v, 1)^T          H rows 0,1 = K[:2,:3] * R * Kinv3,  row 2 = (R * Kinv3)[2]
//   (nx, ny, Yz) = D * (hx, hy, hz) + (a0, a1, tz)     a = K[:2,:3] * t
//   (px, py) = (nx, ny) / max(Yz, 0.1)
// which equals K[:, :3, :3] @ ((T @ [D * Kinv3 @ pix; 1])[:3] / z.clamp(eps).clamp(0.1)) of
// src/tools/geometry.py:312-316, :386, :339-341 up to fp32 re-association.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float uniform(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

struct Cam {
  float H[9];
  float a0, a1, tz;
};

__device__ __forceinline__ void make_cam(Cam& c, const float* __restrict__ T, const float* __restrict__ K,
                                         const float* __restrict__ Ki) {
  float M[9];  // R * Kinv3
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int q = 0; q < 3; ++q)
      M[r*3 + q] = T[r*4 + 0]*Ki[0*4 + q] + T[r*4 + 1]*Ki[1*4 + q] + T[r*4 + 2]*Ki[2*4 + q];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    c.H[0*3 + q] = K[0]*M[q] + K[1]*M[3 + q] + K[2]*M[6 + q];
    c.H[1*3 + q] = K[4]*M[q] + K[5]*M[3 + q] + K[6]*M[6 + q];
    c.H[2*3 + q] = M[6 + q];
  }
  c.a0 = K[0]*T[3] + K[1]*T[7] + K[2]*T[11];
  c.a1 = K[4]*T[3] + K[5]*T[7] + K[6]*T[11];
  c.tz = T[11];
  // The inputs are wave-uniform but fp math runs on the VALU: pin the results into SGPRs so that the twelve constants
  // per support do not occupy vector registers for the whole kernel.
#pragma unroll
  for (int q = 0; q < 9; ++q) c.H[q] = uniform(c.H[q]);
  c.a0 = uniform(c.a0); c.a1 = uniform(c.a1); c.tz = uniform(c.tz);
}

// The same constants with the pixel -> sampling-grid scale (w/(w-1), h/(h-1): the reference normalises by (w-1) and grid_sample
// un-normalises with align_corners=False) folded into rows 0/1, so that a source coordinate is fma(nx, 1/z, -0.5).  Only the
// row-dependent part stays wave-uniform (SGPRs); the column part is folded into three per-lane constants.  Both fused kernels
// build their coordinates from this one function: the backward re-derives exactly the taps the forward blended.
struct Cam2 {
  float H1, H4, H7;        // d(hx, hy, hz)/dv
  float a0, a1, tz;
};
__device__ __forceinline__ void make_cam2(Cam2& c, float& hx0, float& hy0, float& hz0, const float* __restrict__ T,
                                          const float* __restrict__ K, const float* __restrict__ Ki, float wscale, float hscale, float uf) {
  float M[9];  // R * Kinv3
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int q = 0; q < 3; ++q) M[r*3 + q] = T[r*4 + 0]*Ki[0*4 + q] + T[r*4 + 1]*Ki[1*4 + q] + T[r*4 + 2]*Ki[2*4 + q];
  float H[9];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    H[q] = (K[0]*M[q] + K[1]*M[3 + q] + K[2]*M[6 + q])*wscale;
    H[3 + q] = (K[4]*M[q] + K[5]*M[3 + q] + K[6]*M[6 + q])*hscale;
    H[6 + q] = M[6 + q];
  }
  c.H1 = uniform(H[1]); c.H4 = uniform(H[4]); c.H7 = uniform(H[7]);
  c.a0 = uniform((K[0]*T[3] + K[1]*T[7] + K[2]*T[11])*wscale);
  c.a1 = uniform((K[4]*T[3] + K[5]*T[7] + K[6]*T[11])*hscale);
  c.tz = uniform(T[11]);
  hx0 = fmaf(uniform(H[0]), uf, uniform(H[2]));
  hy0 = fmaf(uniform(H[3]), uf, uniform(H[5]));
  hz0 = fmaf(uniform(H[6]), uf, uniform(H[8]));
}

// Bilinear, border-clamped 4-tap gather setup (grid_sample(bilinear, border, align_corners=False)).
struct Taps {
  int off;        // iy*stride + ix of the north-west tap (ix <= w-2, iy <= h-2 so the 2x2 block is in range)
  float fx, fy;   // fractional weights of the east / south taps
  float mx, my;   // d(clamped coord)/d(unclamped coord): 1 strictly inside (0, size-1), else 0
};

__device__ __forceinline__ Taps make_taps(float sx, float sy, int h, int w, int stride) {
  Taps t;
  const float xmax = (float)(w - 1), ymax = (float)(h - 1);
  t.mx = (sx > 0.f && sx < xmax) ? 1.f : 0.f;
  t.my = (sy > 0.f && sy < ymax) ? 1.f : 0.f;
  // NaN-safe clamp (fmaxf/fminf drop NaNs -> 0), then split into integer and fractional parts.
  float cx = fminf(fmaxf(sx, 0.f), xmax), cy = fminf(fmaxf(sy, 0.f), ymax);
  float x0 = fminf(floorf(cx), xmax - 1.f), y0 = fminf(floorf(cy), ymax - 1.f);
  t.fx = cx - x0; t.fy = cy - y0;
  t.off = (int)y0*stride + (int)x0;
  return t;
}

__device__ __forceinline__ float bilerp(const float* __restrict__ plane, const Taps& t, int w, float& ddx, float& ddy) {
  f2u n = *(const f2u*)(plane + t.off);
  f2u s = *(const f2u*)(plane + t.off + w);
  float top = fmaf(t.fx, n.y - n.x, n.x), bot = fmaf(t.fx, s.y - s.x, s.x);
  ddx = fmaf(t.fy, (s.y - s.x) - (n.y - n.x), n.y - n.x);  // d/dfx
  ddy = bot - top;                                          // d/dfy
  return fmaf(t.fy, bot - top, top);
}
__device__ __forceinline__ float bilerp(const float* __restrict__ plane, const Taps& t, int w) {
  f2u n = *(const f2u*)(plane + t.off);
  f2u s = *(const f2u*)(plane + t.off + w);
  float top = fmaf(t.fx, n.y - n.x, n.x), bot = fmaf(t.fx, s.y - s.x, s.x);
  return fmaf(t.fy, bot - top, top);
}

// Uniform base + 32-bit lane offset: lets the compiler use the saddr form (no 64-bit VALU address arithmetic).
__device__ __forceinline__ float ld1(const float* base, unsigned idx) { return *(const float*)((const char*)base + (size_t)(idx*4u)); }
__device__ __forceinline__ f4 ld4(const float* base, unsigned texel) { return *(const f4*)((const char*)base + (size_t)(texel*16u)); }
typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
__device__ __forceinline__ f3 ld3(const float* base, unsigned texel) { const f3u v = *(const f3u*)((const char*)base + (size_t)(texel*12u)); return v; }

// Uniform base (SGPR pair) + per-lane 32-bit BYTE offset: the saddr addressing form, no 64-bit VALU address arithmetic.
__device__ __forceinline__ float ldu(const float* base, unsigned byte_off) { return *(const float*)((const char*)base + (size_t)byte_off); }
// Buffer resources (SRSRC): base + size in four SGPRs.  A row access is `buffer_load v, v_lane_offset, s[rsrc], s_row_offset offen`:
// the per-lane column offset is a loop-invariant VGPR, the row / plane offset is a scalar the SALU advances, so the coalesced
// loads and stores of the streaming kernels cost no VALU address arithmetic at all (with plain pointers LLVM re-associates
// `row pointer + lane offset`, hoists `base + lane offset` out of the row loop as a 64-bit VGPR pair per array and pays
// 64-bit VALU adds per access).  Out-of-range accesses read 0 / are dropped.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(bytes > 0xffffffffull ? 0xffffffffull : bytes), 0x00020000);
}
__device__ __forceinline__ float bld(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ unsigned bld8(rsrc_t r, unsigned voff, unsigned soff) { return __builtin_amdgcn_raw_buffer_load_b8(r, voff, soff, 0); }
__device__ __forceinline__ f4 bld4(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 bld2(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
// 12-byte load of an RGB texel.  (Never a 16-byte load with an ignored fourth lane: the register allocator treats the
// never-read .w register of an in-flight load as free, reuses it for address arithmetic and must then wait (s_waitcnt) for
// that load to land first — which serialises a whole gather batch behind its first load.)
__device__ __forceinline__ f3 bld3(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f3, __builtin_amdgcn_raw_buffer_load_b96(r, voff, soff, 0));
}
__device__ __forceinline__ void bst(rsrc_t r, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
__device__ __forceinline__ void bst8(rsrc_t r, unsigned voff, unsigned soff, unsigned v) {
  __builtin_amdgcn_raw_buffer_store_b8((unsigned char)v, r, voff, soff, 0);
}
__device__ __forceinline__ void bst3(rsrc_t r, unsigned voff, unsigned soff, f3 v) {
  typedef unsigned u3 __attribute__((ext_vector_type(3)));
  __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(u3, v), r, voff, soff, 0);
}
__device__ __forceinline__ void bst4(rsrc_t r, unsigned voff, unsigned soff, f4 v) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, voff, soff, 0);
}

// Horizontal 3-tap sum through DPP wave shifts (each shift rides on a v_add_f32_dpp).  No weights are needed for the
// reflection padding: the halo lane left of column 0 (right of column w-1) synthesises column 1 (w-2) itself, i.e. it
// holds the reflected value.  Written as asm blocks so that each shift stays fused into its add and the pairs stay
// adjacent: left to itself the compiler emits all the shifts of a row first (as v_mov_b32_dpp) and keeps their results
// live, which costs a wave of occupancy.  Two or three independent sums share one block: one s_nop covers the
// VALU-write -> DPP-read hazard (2 wait states) that the compiler cannot see inside asm, and the interleaving hides the
// add latency.
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

// SSIM error of one channel from UN-normalised (x9) window sums: ssim = N/D with N and D both scaled by 81*81.
//   sx = S_x, sxx = S_xx, sxy = S_xy;  sy = S_y, cy1 = S_y^2 + 81 C1, cy2 = 9 S_yy - S_y^2 + 81 C2     (photometric.py:40-50)
__device__ __forceinline__ float ssim_err81(float sx, float sxx, float sxy, float sy, float cy1, float cy2) {
  constexpr float c1 = 81.f*kC1, c2 = 81.f*kC2;
  const float t = sx*sy;
  const float num = fmaf(2.f, t, c1)*fmaf(2.f, fmaf(9.f, sxy, -t), c2);
  const float sx2 = sx*sx;
  const float den = (sx2 + cy1)*(fmaf(9.f, sxx, -sx2) + cy2);
  const float val = fmaf(-0.5f, num*__builtin_amdgcn_rcpf(den), 0.5f);
  return fminf(fmaxf(val, 0.f), 1.f);
}

// SSIM error of one channel from the nine-tap window sums (already divided by 9 where noted).
//   mx = E[x], exx = E[x^2], exy = E[xy];  my = E[y], cy1 = my^2 + C1, cy2 = var(y) + C2
__device__ __forceinline__ float ssim_err(float mx, float exx, float exy, float my, float cy1, float cy2) {
  float sx = exx - mx*mx, sxy = exy - mx*my;
  float num = fmaf(2.f*mx, my, kC1)*fmaf(2.f, sxy, kC2);
  float den = fmaf(mx, mx, cy1)*(sx + cy2);
  float v = fmaf(-0.5f, num*__builtin_amdgcn_rcpf(den), 0.5f);
  return fminf(fmaxf(v, 0.f), 1.f);
}

// Partials of the (unclamped) SSIM error e = (1 - num/den)/2 w.r.t. the window means (E[x], E[x^2], E[xy]);
// zero where the clamp(0, 1) of photometric.py:50 is active.
__device__ __forceinline__ void ssim_err_grad(float mx, float exx, float exy, float my, float cy1, float cy2,
                                              float& d_mx, float& d_exx, float& d_exy) {
  float sx = exx - mx*mx, sxy = exy - mx*my;
  float a1 = fmaf(2.f*mx, my, kC1), a2 = fmaf(2.f, sxy, kC2);
  float b1 = fmaf(mx, mx, cy1), b2 = sx + cy2;
  float rden = __builtin_amdgcn_rcpf(b1*b2);
  float val = a1*a2*rden;
  float e = fmaf(-0.5f, val, 0.5f);
  float pass = (e >= 0.f && e <= 1.f) ? -0.5f : 0.f;  // de/dval, gated by the clamp
  // dval/dmx  (treating mx, exx, exy as independent): [2 my (a2 - a1) - 2 mx val (b2 - b1)] / den
  d_mx = pass*(2.f*my*(a2 - a1) - 2.f*mx*val*(b2 - b1))*rden;
  d_exx = pass*(-val*b1)*rden;       // dval/dexx = -val / b2
  d_exy = pass*(2.f*a1)*rden;        // dval/dexy = 2 a1 / den
}

// Reader side of the in-launch hand-offs (the wave that arrived last reads what the others published).  Every such read in this
// library is an agent-scope load (sc1: `__hip_atomic_load(.., AGENT)` or a buffer load with aux = 16), which is coherent at agent
// scope by itself — the two-granule form R2 of cdna_hip_programming.md Guideline 16 — so no acquire fence (`buffer_inv sc1`,
// ~1.7 us on the critical path of a launch's last wave) is needed; it would be before PLAIN loads.  -DSMD_TAIL_FENCE restores it.
#ifdef SMD_TAIL_FENCE
#define SMD_TAIL_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#else
#define SMD_TAIL_ACQUIRE() ((void)0)
#endif

// Counter-based Gaussian for the automask tie-break when the caller supplies no noise tensor.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float gauss_noise(uint32_t seed_lo, uint32_t seed_hi, uint32_t idx) {
  uint32_t a = hash32(idx ^ seed_lo), b = hash32(a ^ seed_hi ^ 0x9e3779b9U);
  float u1 = (float)(a >> 8)*(1.0f/16777216.0f) + (0.5f/16777216.0f);
  float u2 = (float)(b >> 8)*(1.0f/16777216.0f);
  return sqrtf(-2.f*__logf(u1))*__cosf(6.2831853f*u2);
}

}  // namespace smd
