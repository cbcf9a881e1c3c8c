// What does a buffer gather cost the texture path when lanes are out of range or masked off?  (GPU box)
// 16 waves per CU, each issuing `iters` x 8 independent 12-byte buffer loads over an L2-resident 1 MB array.
// build: hipcc --offload-arch=gfx950 -O3 ta_probe.hip -o ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef float f3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ f3 bld3(rsrc_t r, unsigned voff, unsigned soff) { return __builtin_bit_cast(f3, __builtin_amdgcn_raw_buffer_load_b96(r, voff, soff, 0)); }
// mode 0: all lanes in range, contiguous texels;  1: all lanes out of range;  2: odd QUADS out of range;  3: odd LANES out of range;
// 4: exec-masked odd quads (divergent branch);  5: exec-masked odd lanes;  6: all lanes, one dword instead of three
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* p, float* out, int iters) {
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 1 << 20, 0x00020000);
  const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x*4 + (threadIdx.x >> 6));
  unsigned base = (wave*977u % 1000u)*768u + lane*12u;
  const bool odd_quad = (lane >> 2) & 1, odd_lane = lane & 1;
  if (MODE == 1) base = 0x40000000u;
  if (MODE == 2 && odd_quad) base = 0x40000000u;
  if (MODE == 3 && odd_lane) base = 0x40000000u;
  f3 acc = {0.f, 0.f, 0.f};
  for (int i = 0; i < iters; ++i) {
    const unsigned so = (unsigned)(i & 63)*3072u;
    if ((MODE == 4 && odd_quad) || (MODE == 5 && odd_lane)) continue;
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) {
      if (MODE == 6) acc.x += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, base + k2*12u, so, 0));
      else acc += bld3(r, base + k2*12u, so);
    }
  }
  if (acc.x + acc.y + acc.z == 12345.678f) out[0] = acc.x;
}
template <typename K> void run(const char* name, K kern, const float* p, float* o) {
  const int iters = 2000, blocks = 256*4;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, p, o, 10); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, p, o, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  // per CU: 16 waves x iters x 8 wave-loads
  printf("%-44s %.3f ms -> %.1f cycles@2.4GHz per wave-load per CU\n", name, ms, ms*1e-3*2.4e9/((double)iters*8*16));
}
int main() {
  float *p, *o; (void)hipMalloc(&p, 1 << 20); (void)hipMalloc(&o, 64); (void)hipMemset(p, 0, 1 << 20);
  run("12-B gather, all lanes in range", k<0>, p, o);
  run("all lanes out of range", k<1>, p, o);
  run("odd quads out of range", k<2>, p, o);
  run("odd lanes out of range", k<3>, p, o);
  run("odd quads masked off (exec)", k<4>, p, o);
  run("odd lanes masked off (exec)", k<5>, p, o);
  run("4-B load, all lanes in range", k<6>, p, o);
  return 0;
}
