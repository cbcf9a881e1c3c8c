// lds_dma_probe.hip — where does `buffer_load_dwordx3/x4 ... lds` (LDS-DMA, gfx950) put a lane's data?  Finding: both at M0 + 16*lane —
// the 12-byte form keeps the 16-byte lane stride and leaves the fourth dword of each slot untouched.  (GPU box)
// build + run: hipcc --offload-arch=gfx950 -O3 scripts/dev/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int rsrc_t __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* dst, int n) {
  __shared__ __attribute__((aligned(16))) float ring[256 + 256];
  rsrc_t r;
  unsigned long long p = (unsigned long long)src;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p); r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(p >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane(n*4); r.w = __builtin_amdgcn_readfirstlane(0x00020000);
  const unsigned lane = threadIdx.x & 63;
  const unsigned src_lane = 63u - lane;                      // per-lane source address: reversed, to tell it from the LDS order
  unsigned voff = src_lane*16u, voff3 = src_lane*12u;
  unsigned soff = __builtin_amdgcn_readfirstlane(blockIdx.x*1024u);
  unsigned ldsb = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)ring);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(r), "s"(soff), "s"(ldsb) : "memory");
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx3 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff3), "s"(r), "s"(soff), "s"(ldsb + 1024u) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = threadIdx.x; e < 512; e += 64) dst[blockIdx.x*512 + e] = ring[e];
}
int main() {
  const int n = 1024;
  std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *s, *d; hipMalloc(&s, n*4); hipMalloc(&d, 512*4*2); hipMemcpy(s, h.data(), n*4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(2), dim3(64), 0, 0, s, d, n);
  std::vector<float> o(512*2); hipMemcpy(o.data(), d, 512*4*2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int blk = 0; blk < 2; ++blk) for (int l = 0; l < 64; ++l) {
    for (int c = 0; c < 4; ++c) if (o[blk*512 + l*4 + c] != (float)(blk*256 + (63 - l)*4 + c)) ++bad;
    for (int c = 0; c < 3; ++c) if (o[blk*512 + 256 + l*4 + c] != (float)(blk*256 + (63 - l)*3 + c)) ++bad;   // 16-byte lane stride
  }
  printf("lds-dma probe: %d mismatches against {x4: lane l -> LDS[16 l .. 16 l + 15], x3: lane l -> LDS[1024 + 16 l .. + 11]}; x4 lane 0 = %g %g %g %g, x3 lane 0 = %g %g %g, x3 lane 1 = %g %g %g\n",
         bad, o[0], o[1], o[2], o[3], o[256], o[257], o[258], o[260], o[261], o[262]);
  return bad != 0;
}
