// Issue-rate microbenchmark #2 for gfx950 (round 2): packed fp32, 64-bit moves, DPP flavours, conversions, mixed streams,
// cross-lane through the LDS pipe, and the gather path.  Every kernel runs 8 (or 4 packed) independent dependency chains.
// build: hipcc --offload-arch=gfx950 -O3 valu_rate2.hip -o valu_rate2 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define REP8(x) x x x x x x x x
#define IO8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define BODY(NAME, ASM)                                                                    \
  __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                     \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    float b = 1.0001f, c = 0.5f;                                                           \
    for (int i = 0; i < iters; ++i) { REP8(asm volatile(ASM : IO8 : "v"(b), "v"(c));) }    \
    out[blockIdx.x*256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;            \
  }
#define BODY2(NAME, ASM)  /* four packed chains */                                         \
  __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                     \
    f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b = {1.0001f, 0.999f}, c = {0.5f, 0.25f}; \
    for (int i = 0; i < iters; ++i) { REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) } \
    f2 r = a0 + a1 + a2 + a3; out[blockIdx.x*256 + threadIdx.x] = r.x + r.y;              \
  }
#define OP8(op) op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n"
#define OP8_2(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
#define OP8_2S(op, suf) op " %0, %0, %8 " suf "\n" op " %1, %1, %8 " suf "\n" op " %2, %2, %8 " suf "\n" op " %3, %3, %8 " suf "\n" op " %4, %4, %8 " suf "\n" op " %5, %5, %8 " suf "\n" op " %6, %6, %8 " suf "\n" op " %7, %7, %8 " suf "\n"
#define OP8_1(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
#define DPPSUF(ctl) " " ctl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define OP8_DPP(op, ctl) op " %0, %1, %0" DPPSUF(ctl) op " %1, %2, %1" DPPSUF(ctl) op " %2, %3, %2" DPPSUF(ctl) op " %3, %4, %3" DPPSUF(ctl) op " %4, %5, %4" DPPSUF(ctl) op " %5, %6, %5" DPPSUF(ctl) op " %6, %7, %6" DPPSUF(ctl) op " %7, %0, %7" DPPSUF(ctl)
#define PK4(op3) op3(0) op3(1) op3(2) op3(3) op3(0) op3(1) op3(2) op3(3)
#define PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %4, %5\n"
#define PKFMA_BC(i) "v_pk_fma_f32 %" #i ", %" #i ", %4, %5 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"
#define PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %4\n"
#define PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %4\n"
#define PKADD_NEG(i) "v_pk_add_f32 %" #i ", %" #i ", %4 neg_lo:[0,1] neg_hi:[0,1]\n"
#define PKFMA_CL(i) "v_pk_fma_f32 %" #i ", %" #i ", %4, %5 clamp\n"
#define MOV64(i) "v_mov_b64 %" #i ", %" #i "\n"
BODY(k_fma, OP8("v_fma_f32"))
BODY(k_add, OP8_2("v_add_f32"))
BODY(k_mul, OP8_2("v_mul_f32"))
BODY(k_max, OP8_2("v_max_f32"))
BODY(k_min, OP8_2("v_min_f32"))
BODY(k_med3, OP8("v_med3_f32"))
BODY(k_mov, OP8_1("v_mov_b32"))
BODY(k_rcp, OP8_1("v_rcp_f32"))
BODY(k_floor, OP8_1("v_floor_f32"))
BODY(k_fract, OP8_1("v_fract_f32"))
BODY(k_cvt_i, OP8_1("v_cvt_i32_f32"))
BODY(k_cvt_u, OP8_1("v_cvt_u32_f32"))
BODY(k_lshl, OP8_2S("v_lshlrev_b32", ""))
BODY(k_addu, OP8_2("v_add_u32"))
BODY(k_mad24, OP8("v_mad_u32_u24"))
BODY(k_lshl_add, OP8("v_lshl_add_u32"))
BODY(k_fma_clamp, "v_fma_f32 %0, %0, %8, %9 clamp\n v_fma_f32 %1, %1, %8, %9 clamp\n v_fma_f32 %2, %2, %8, %9 clamp\n v_fma_f32 %3, %3, %8, %9 clamp\n v_fma_f32 %4, %4, %8, %9 clamp\n v_fma_f32 %5, %5, %8, %9 clamp\n v_fma_f32 %6, %6, %8, %9 clamp\n v_fma_f32 %7, %7, %8, %9 clamp\n")
BODY(k_add_abs, "v_add_f32_e64 %0, %0, |%8|\n v_add_f32_e64 %1, %1, |%8|\n v_add_f32_e64 %2, %2, |%8|\n v_add_f32_e64 %3, %3, |%8|\n v_add_f32_e64 %4, %4, |%8|\n v_add_f32_e64 %5, %5, |%8|\n v_add_f32_e64 %6, %6, |%8|\n v_add_f32_e64 %7, %7, |%8|\n")
BODY(k_add_dpp_wave, OP8_DPP("v_add_f32_dpp", "wave_shr:1"))
BODY(k_add_dpp_row, OP8_DPP("v_add_f32_dpp", "row_shr:1"))
BODY(k_add_dpp_quad, OP8_DPP("v_add_f32_dpp", "quad_perm:[1,0,3,2]"))
BODY(k_add_dpp_ror, OP8_DPP("v_add_f32_dpp", "wave_ror:1"))
BODY(k_fmac_dpp_wave, OP8_DPP("v_fmac_f32_dpp", "wave_shr:1"))
// mixed streams: 4 plain + 4 DPP; 7 fma + 1 rcp; 6 fma + 2 rcp
BODY(k_mix_fma_dpp, "v_fma_f32 %0, %0, %8, %9\n v_add_f32_dpp %1, %2, %1" DPPSUF("wave_shr:1") "v_fma_f32 %2, %2, %8, %9\n v_add_f32_dpp %3, %4, %3" DPPSUF("wave_shr:1") "v_fma_f32 %4, %4, %8, %9\n v_add_f32_dpp %5, %6, %5" DPPSUF("wave_shr:1") "v_fma_f32 %6, %6, %8, %9\n v_add_f32_dpp %7, %0, %7" DPPSUF("wave_shr:1"))
BODY(k_mix_fma7_rcp1, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_rcp_f32 %3, %3\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
BODY(k_mix_fma6_rcp2, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_rcp_f32 %3, %3\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_rcp_f32 %7, %7\n")
BODY(k_mix_fma4_max4, "v_fma_f32 %0, %0, %8, %9\n v_max_f32 %1, %1, %8\n v_fma_f32 %2, %2, %8, %9\n v_max_f32 %3, %3, %8\n v_fma_f32 %4, %4, %8, %9\n v_max_f32 %5, %5, %8\n v_fma_f32 %6, %6, %8, %9\n v_max_f32 %7, %7, %8\n")
BODY2(k_pkfma, PK4(PKFMA))
BODY2(k_pkfma_bc, PK4(PKFMA_BC))
BODY2(k_pkfma_clamp, PK4(PKFMA_CL))
BODY2(k_pkmul, PK4(PKMUL))
BODY2(k_pkadd, PK4(PKADD))
BODY2(k_pkadd_neg, PK4(PKADD_NEG))
BODY2(k_mov64, PK4(MOV64))
// packed + plain mixed: 2 pk_fma + 2 v_fma alternating (does the mix matter?)
__global__ __launch_bounds__(256) void k_mix_pk_plain(float* out, int iters) {
  f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, b = {1.0001f, 0.999f}, c = {0.5f, 0.25f};
  float s0 = threadIdx.x, s1 = s0 + 1.f, s2 = s0 + 2.f, s3 = s0 + 3.f;
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("v_pk_fma_f32 %0, %0, %6, %7\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %1, %1, %6, %7\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                      "v_pk_fma_f32 %0, %0, %6, %7\n v_fma_f32 %2, %2, %8, %9\n"
                      : "+v"(a0), "+v"(a1), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(b), "v"(c), "v"(b.x), "v"(c.x));)
  }
  f2 r = a0 + a1; out[blockIdx.x*256 + threadIdx.x] = r.x + r.y + s0 + s1 + s2 + s3;
}
// cross-lane through the LDS pipe next to plain VALU: 8 fma + 2 ds_swizzle / ds_bpermute per group
__global__ __launch_bounds__(256) void k_swz_only(float* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %1, %1 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %2, %2 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %3, %3 offset:swizzle(SWAP,1)\n"
                      "s_waitcnt lgkmcnt(0)\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
  }
  out[blockIdx.x*256 + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(256) void k_bperm_only(float* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  int idx = ((threadIdx.x + 1) & 63)*4;
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n"
                      "s_waitcnt lgkmcnt(0)\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(idx));)
  }
  out[blockIdx.x*256 + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(256) void k_fma8_bperm2(float* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float p0 = a0*3.f, p1 = a1*3.f, b = 1.0001f, c = 0.5f;
  int idx = ((threadIdx.x + 1) & 63)*4;
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("ds_bpermute_b32 %8, %12, %8\n ds_bpermute_b32 %9, %12, %9\n"
                      "v_fma_f32 %0, %0, %10, %11\n v_fma_f32 %1, %1, %10, %11\n v_fma_f32 %2, %2, %10, %11\n v_fma_f32 %3, %3, %10, %11\n"
                      "v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n v_fma_f32 %6, %6, %10, %11\n v_fma_f32 %7, %7, %10, %11\n"
                      "s_waitcnt lgkmcnt(0)\n" : IO8, "+v"(p0), "+v"(p1) : "v"(b), "v"(c), "v"(idx));)
  }
  out[blockIdx.x*256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0 + p1;
}
// the gather path: 4 x 16-byte taps of a 2x2 texel block per lane per step, pseudo-random but local offsets
template <int MODE>   // 0: global_load via pointer arithmetic; 1: buffer load with idxen (stride 16) + soffset for the second row
__global__ __launch_bounds__(256) void k_gather(const f4* __restrict__ tex, float* out, int iters, unsigned mask, unsigned rowtex) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  unsigned o = (blockIdx.x*977u + threadIdx.x*3u) & mask;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tex, 16, (mask + rowtex + 4)*16, 0x00020000 | (1 << 27));
  (void)rs;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f4 t0, t1, t2, t3;
      if (MODE == 0) { t0 = tex[o]; t1 = tex[o + 1]; t2 = tex[o + rowtex]; t3 = tex[o + rowtex + 1]; }
      else {
        asm volatile("buffer_load_dwordx4 %0, %4, %5, 0 idxen\n buffer_load_dwordx4 %1, %4, %5, 0 idxen offset:16\n"
                     "buffer_load_dwordx4 %2, %4, %5, %6 idxen\n buffer_load_dwordx4 %3, %4, %5, %6 idxen offset:16\n s_waitcnt vmcnt(0)"
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(o), "s"(rs), "s"(rowtex*16u) : "memory");
      }
      acc += (t0 + t1) + (t2 + t3);
      o = (o + 67u + (unsigned)r) & mask;
    }
  }
  out[blockIdx.x*256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <typename K> double run(const char* name, K kern, float* d, int waves_per_simd, int per_iter = 64) {
  const int iters = 2000, blocks = 256*waves_per_simd;  // 4 waves per block -> waves_per_simd blocks per CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_simd = (double)iters*per_iter*waves_per_simd;
  double ns_per_instr = ms*1e6/instr_per_simd;
  printf("%-22s waves/SIMD %d: %8.3f ms  -> %.3f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, waves_per_simd, ms, ns_per_instr, ns_per_instr*2.4);
  return ns_per_instr;
}
template <int MODE> void run_gather(const char* name, const f4* tex, float* d, unsigned texels, unsigned rowtex, int waves_per_simd) {
  const int iters = 500, blocks = 256*waves_per_simd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(256), 0, 0, tex, d, 10, texels - 1, rowtex);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(256), 0, 0, tex, d, iters, texels - 1, rowtex);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double loads_per_simd = (double)iters*16*waves_per_simd;
  printf("%-22s table %6u KB waves/SIMD %d: %8.3f ms -> %.2f ns per 16-B wave-load per SIMD (%.1f cycles), %.2f TB/s of taps\n", name, texels*16/1024, waves_per_simd, ms,
         ms*1e6/loads_per_simd, ms*1e6/loads_per_simd*2.4, (double)blocks*256*iters*16*16/(ms*1e-3)/1e12);
}
int main() {
  float* d; hipMalloc(&d, 256*8*256*sizeof(float)*4);
  f4* tex; const unsigned maxtex = 1u << 23; hipMalloc(&tex, (size_t)(maxtex + 4096)*16); hipMemset(tex, 0, (size_t)(maxtex + 4096)*16);
  for (int w : {1, 2, 4}) {
    run("v_fma_f32", k_fma, d, w); run("v_add_f32", k_add, d, w); run("v_mul_f32", k_mul, d, w);
    run("v_max_f32", k_max, d, w); run("v_min_f32", k_min, d, w); run("v_med3_f32", k_med3, d, w); run("v_mov_b32", k_mov, d, w);
    run("v_rcp_f32", k_rcp, d, w); run("v_floor_f32", k_floor, d, w); run("v_fract_f32", k_fract, d, w);
    run("v_cvt_i32_f32", k_cvt_i, d, w); run("v_cvt_u32_f32", k_cvt_u, d, w); run("v_lshlrev_b32", k_lshl, d, w);
    run("v_add_u32", k_addu, d, w); run("v_mad_u32_u24", k_mad24, d, w); run("v_lshl_add_u32", k_lshl_add, d, w);
    run("v_fma_f32 clamp", k_fma_clamp, d, w); run("v_add_f32 |abs|", k_add_abs, d, w);
    run("add_dpp wave_shr", k_add_dpp_wave, d, w); run("add_dpp row_shr", k_add_dpp_row, d, w); run("add_dpp quad_perm", k_add_dpp_quad, d, w);
    run("add_dpp wave_ror", k_add_dpp_ror, d, w); run("fmac_dpp wave_shr", k_fmac_dpp_wave, d, w);
    run("mix 4fma+4dpp", k_mix_fma_dpp, d, w); run("mix 7fma+1rcp", k_mix_fma7_rcp1, d, w); run("mix 6fma+2rcp", k_mix_fma6_rcp2, d, w);
    run("mix 4fma+4max", k_mix_fma4_max4, d, w);
    run("v_pk_fma_f32", k_pkfma, d, w); run("v_pk_fma_f32 op_sel", k_pkfma_bc, d, w); run("v_pk_fma_f32 clamp", k_pkfma_clamp, d, w);
    run("v_pk_mul_f32", k_pkmul, d, w); run("v_pk_add_f32", k_pkadd, d, w); run("v_pk_add_f32 neg", k_pkadd_neg, d, w); run("v_mov_b64", k_mov64, d, w);
    run("mix 3pk+5fma", k_mix_pk_plain, d, w);
    run("ds_swizzle only", k_swz_only, d, w, 32); run("ds_bpermute only", k_bperm_only, d, w, 32); run("8fma+2bperm (per 10)", k_fma8_bperm2, d, w, 80);
  }
  for (int w : {2, 4}) {
    for (unsigned texels : {4096u, 131072u, 4194304u}) {
      run_gather<0>("gather global_load", tex, d, texels, 641, w);
      run_gather<1>("gather buffer idxen", tex, d, texels, 641, w);
    }
  }
  return 0;
}
