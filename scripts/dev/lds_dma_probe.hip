// lds_dma_probe.hip — does `buffer_load_dwordx3/x4 ... lds` (LDS-DMA, gfx950) land lane-linear at M0 + lane*size?  (GPU box)
// build + run: hipcc --offload-arch=gfx950 -O3 scripts/dev/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int rsrc_t __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* dst, int n) {
  __shared__ __attribute__((aligned(16))) float ring[256 + 192];
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
  for (int e = threadIdx.x; e < 448; e += 64) dst[blockIdx.x*448 + e] = ring[e];
}
int main() {
  const int n = 1024;
  std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *s, *d; hipMalloc(&s, n*4); hipMalloc(&d, 448*4*2); hipMemcpy(s, h.data(), n*4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(2), dim3(64), 0, 0, s, d, n);
  std::vector<float> o(448*2); hipMemcpy(o.data(), d, 448*4*2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int blk = 0; blk < 2; ++blk) for (int l = 0; l < 64; ++l) {
    for (int c = 0; c < 4; ++c) if (o[blk*448 + l*4 + c] != (float)(blk*256 + (63 - l)*4 + c)) ++bad;
    for (int c = 0; c < 3; ++c) if (o[blk*448 + 256 + l*3 + c] != (float)(blk*256 + (63 - l)*3 + c)) ++bad;
  }
  for (int blk = 0, shown = 0; blk < 2; ++blk) for (int e = 0; e < 448 && shown < 24; ++e) {
    const int l = e < 256 ? e/4 : (e - 256)/3, c = e < 256 ? e%4 : (e - 256)%3;
    const float want = (float)(blk*256 + (63 - l)*(e < 256 ? 4 : 3) + c);
    if (o[blk*448 + e] != want) { printf("  blk %d e %d (lane %d c %d): got %g want %g\n", blk, e, l, c, o[blk*448 + e], want); ++shown; }
  }
  printf("lds-dma probe: %d mismatches (x4: lane l -> LDS[l*16], x3: lane l -> LDS[1024 + l*12]); sample x4 lane0 = %g %g %g %g, x3 lane0 = %g %g %g\n", bad,
         o[0], o[1], o[2], o[3], o[256], o[257], o[258]);
  return bad != 0;
}
