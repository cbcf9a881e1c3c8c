// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction at full occupancy.
// build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define BODY(NAME, ASM)                                                                    \
  __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                     \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    float b = 1.0001f, c = 0.5f;                                                           \
    for (int i = 0; i < iters; ++i) {                                                      \
      REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) \
    }                                                                                      \
    out[blockIdx.x*256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;            \
  }
#define OP8(op) op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n"
#define OP8_2(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
#define OP8_1(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
#define OP8_DPP(op) op " %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" op " %1, %2, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" op " %2, %3, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" op " %3, %4, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" op " %4, %5, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" op " %5, %6, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" op " %6, %7, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" op " %7, %0, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define OP8_ROWDPP(op) op " %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" op " %1, %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" op " %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" op " %3, %4, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" op " %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" op " %5, %6, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" op " %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" op " %7, %0, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define OP8_MOVDPP "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
BODY(k_fma, OP8("v_fma_f32"))
BODY(k_fmac, OP8_2("v_fmac_f32"))   // VOP2: a += a*b  (v_fmac_f32 dst, src0, src1 : dst += src0*src1)
BODY(k_add, OP8_2("v_add_f32"))
BODY(k_mul, OP8_2("v_mul_f32"))
BODY(k_max, OP8_2("v_max_f32"))
BODY(k_med3, OP8("v_med3_f32"))
BODY(k_mov, OP8_1("v_mov_b32"))
BODY(k_rcp, OP8_1("v_rcp_f32"))
BODY(k_floor, OP8_1("v_floor_f32"))
BODY(k_cvt, OP8_1("v_cvt_i32_f32"))
BODY(k_add_dpp_wave, OP8_DPP("v_add_f32_dpp"))
BODY(k_add_dpp_row, OP8_ROWDPP("v_add_f32_dpp"))
BODY(k_mov_dpp_wave, OP8_MOVDPP)
BODY(k_addu, OP8_2("v_add_u32"))
BODY(k_mad24, OP8("v_mad_u32_u24"))
__global__ __launch_bounds__(256) void k_pkfma(float* out, int iters) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b = {1.0001f, 0.999f}, c = {0.5f, 0.25f};
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
  }
  f2 r = a0 + a1 + a2 + a3;
  out[blockIdx.x*256 + threadIdx.x] = r.x + r.y;
}
__global__ __launch_bounds__(256) void k_pkadd(float* out, int iters) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b = {1.0001f, 0.999f};
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
  }
  f2 r = a0 + a1 + a2 + a3;
  out[blockIdx.x*256 + threadIdx.x] = r.x + r.y;
}
template <typename K> void run(const char* name, K kern, float* d, int waves_per_simd) {
  const int iters = 2000, blocks = 256*waves_per_simd;  // 4 waves per block -> waves_per_simd blocks per CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_simd = (double)iters*64*waves_per_simd;        // each SIMD runs waves_per_simd waves
  double ns_per_instr = ms*1e6/instr_per_simd;
  printf("%-18s waves/SIMD %d: %8.3f ms  -> %.3f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, waves_per_simd, ms, ns_per_instr, ns_per_instr*2.4);
}
int main() {
  float* d; hipMalloc(&d, 256*8*256*sizeof(float)*4);
  for (int w : {1, 4}) {
    run("v_fma_f32", k_fma, d, w); run("v_fmac_f32", k_fmac, d, w); run("v_add_f32", k_add, d, w); run("v_mul_f32", k_mul, d, w);
    run("v_max_f32", k_max, d, w); run("v_med3_f32", k_med3, d, w); run("v_mov_b32", k_mov, d, w); run("v_rcp_f32", k_rcp, d, w);
    run("v_floor_f32", k_floor, d, w); run("v_cvt_i32_f32", k_cvt, d, w); run("v_add_f32_dpp wave", k_add_dpp_wave, d, w);
    run("v_add_f32_dpp row", k_add_dpp_row, d, w); run("v_mov_b32_dpp wave", k_mov_dpp_wave, d, w); run("v_add_u32", k_addu, d, w);
    run("v_mad_u32_u24", k_mad24, d, w); run("v_pk_fma_f32", k_pkfma, d, w); run("v_pk_add_f32", k_pkadd, d, w);
  }
  return 0;
}
