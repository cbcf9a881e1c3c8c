// Does packed fp32 pay in a REALISTIC instruction mix?  The SSIM window algebra + sliding sums of one pixel row for 6 (support,
// channel) pairs, once with scalar floats and once with float2 (v_pk_*), no memory traffic, 4 or 8 waves per SIMD... (run on GPU).
// build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize pk_probe.hip -o pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ T rcpT(T x);
template <> __device__ __forceinline__ float rcpT<float>(float x) { return __builtin_amdgcn_rcpf(x); }
template <> __device__ __forceinline__ f2 rcpT<f2>(f2 x) { return f2{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
template <typename T> __device__ __forceinline__ T fmaT(T a, T b, T c);
template <> __device__ __forceinline__ float fmaT<float>(float a, float b, float c) { return fmaf(a, b, c); }
template <> __device__ __forceinline__ f2 fmaT<f2>(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
template <typename T> __device__ __forceinline__ T clampT(T v);
template <> __device__ __forceinline__ float clampT<float>(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
template <> __device__ __forceinline__ f2 clampT<f2>(f2 v) { return f2{fminf(fmaxf(v.x, 0.f), 1.f), fminf(fmaxf(v.y, 0.f), 1.f)}; }
template <typename T> __device__ __forceinline__ T bc(float x);
template <> __device__ __forceinline__ float bc<float>(float x) { return x; }
template <> __device__ __forceinline__ f2 bc<f2>(float x) { return f2{x, x}; }

template <typename T, int NQ>   // NQ values of type T = 6 scalar pairs or 3 packed ones
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  T xo[NQ], Px[NQ], Pxx[NQ], Pxy[NQ], yo[NQ], acc[NQ];
  const float s = threadIdx.x*1e-3f;
  for (int q = 0; q < NQ; ++q) { xo[q] = bc<T>(0.3f + s + q*0.01f); Px[q] = bc<T>(0.6f); Pxx[q] = bc<T>(0.2f); Pxy[q] = bc<T>(0.21f); yo[q] = bc<T>(0.31f + q*0.02f); acc[q] = bc<T>(0.f); }
  const T c1 = bc<T>(0.0081f), c2 = bc<T>(0.0729f), two = bc<T>(2.f), nine = bc<T>(9.f), mh = bc<T>(-0.5f), half = bc<T>(0.5f);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const T xn = xo[q]*bc<T>(0.999f) + bc<T>(1e-4f), yn = yo[q];
      const T xx = xn*xn, xy = xn*yn;
      const T Vx = Px[q] + xn, Vxx = Pxx[q] + xx, Vxy = Pxy[q] + xy;
      Px[q] = xo[q] + xn; Pxx[q] = fmaT<T>(xo[q], xo[q], xx); Pxy[q] = fmaT<T>(xo[q], yo[q], xy);
      xo[q] = xn;
      const T sy = yn*nine, cy1 = fmaT<T>(sy, sy, c1), cy2 = c2 + sy;
      const T t = Vx*sy;
      const T num = fmaT<T>(two, t, c1)*fmaT<T>(two, fmaT<T>(nine, Vxy, -t), c2);
      const T sx2 = Vx*Vx;
      const T den = (sx2 + cy1)*(fmaT<T>(nine, Vxx, -sx2) + cy2);
      acc[q] += clampT<T>(fmaT<T>(mh, num*rcpT<T>(den), half));
    }
  }
  T r = acc[0];
  for (int q = 1; q < NQ; ++q) r += acc[q];
  float rr; if constexpr (sizeof(T) == 8) rr = r.x + r.y; else rr = r;
  out[blockIdx.x*256 + threadIdx.x] = rr;
}
template <typename K> void run(const char* name, K kern, float* d, int wps) {
  const int iters = 4000, blocks = 256*wps;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 10); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-10s waves/SIMD %d: %.3f ms -> %.1f cycles@2.4GHz per iteration (6 channel-pairs) per SIMD-wave-slot\n", name, wps, ms, ms*1e-3*2.4e9/(double)iters/wps);
}
int main() {
  float* d; (void)hipMalloc(&d, 256*8*256*sizeof(float));
  for (int w : {1, 2, 4, 8}) { run("scalar", k<float, 6>, d, w); run("packed", k<f2, 3>, d, w); }
  return 0;
}
