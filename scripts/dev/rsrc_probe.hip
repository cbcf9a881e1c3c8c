// Does the hardware bounds check of a raw buffer load include the scalar offset (soffset)?  (GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__global__ void k(const float* p, float* out) {
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 64, 0x00020000);   // 16 floats
  out[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 8, 0, 0));      // in range: p[2]
  out[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 128, 0, 0));    // voffset out of range
  out[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 8, 128, 0));    // soffset out of range: p[34] or 0?
  out[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 60, 8, 0));     // voffset in range, sum out: p[17] or 0?
  out[4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 0, 0, 128));    // inst offset out of range
}
int main() {
  float h[64]; for (int i = 0; i < 64; ++i) h[i] = 100.f + i;
  float *d, *o; (void)hipMalloc(&d, sizeof h); (void)hipMalloc(&o, 32); (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, o);
  float r[5]; (void)hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  printf("in-range %g | voffset OOB %g | soffset OOB %g | sum OOB %g | inst OOB %g\n", r[0], r[1], r[2], r[3], r[4]);
  return 0;
}
