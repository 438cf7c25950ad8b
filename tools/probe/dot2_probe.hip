// Is v_dot2_f32_f16(a, b, c) == fmaf(a.y, b.y, fmaf(a.x, b.x, c)) bit for bit? (or the other order, or neither)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ uint64_t mix64(uint64_t x){ x+=0x9E3779B97F4A7C15ull; x=(x^(x>>30))*0xBF58476D1CE4E5B9ull; x=(x^(x>>27))*0x94D049BB133111EBull; return x^(x>>31);}
__global__ void k(unsigned long long* out, int iters) {
  uint64_t s = mix64(blockIdx.x * 1024ull + threadIdx.x);
  unsigned long long bad_xy = 0, bad_yx = 0, bad_chain = 0;
  for (int it = 0; it < iters; ++it) {
    // a 64-dim chain like the kernel's: values ~ N(0, 0.1)-ish from random bits
    float c1 = 0.f, c2 = 0.f;
    for (int i = 0; i < 32; ++i) {
      s = mix64(s);
      // two halfs in [-1,1): take random mantissas with exponent in a small range
      uint16_t hx = (uint16_t)((s & 0x83FF) | (((s >> 16) % 6 + 9) << 10));
      uint16_t hy = (uint16_t)(((s >> 24) & 0x83FF) | (((s >> 40) % 6 + 9) << 10));
      h2 v; v.x = __builtin_bit_cast(_Float16, hx); v.y = __builtin_bit_cast(_Float16, hy);
      float r_xy = __builtin_fmaf((float)v.y, (float)v.y, __builtin_fmaf((float)v.x, (float)v.x, c1));
      float r_yx = __builtin_fmaf((float)v.x, (float)v.x, __builtin_fmaf((float)v.y, (float)v.y, c1));
      float d = __builtin_amdgcn_fdot2(v, v, c2, false);
      float d1 = __builtin_amdgcn_fdot2(v, v, c1, false);
      bad_xy += (__float_as_uint(d1) != __float_as_uint(r_xy));
      bad_yx += (__float_as_uint(d1) != __float_as_uint(r_yx));
      c1 = r_xy; c2 = d;
    }
    bad_chain += (__float_as_uint(c1) != __float_as_uint(c2));
  }
  atomicAdd(&out[0], bad_xy); atomicAdd(&out[1], bad_yx); atomicAdd(&out[2], bad_chain);
}
int main() {
  unsigned long long* d; hipMalloc(&d, 24); hipMemset(d, 0, 24);
  int iters = 64;
  hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d, iters);
  unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  double n = 4096.0 * 256 * iters * 32;
  printf("steps=%.0f dot2!=fma(x then y): %llu  dot2!=fma(y then x): %llu  chains(64 dims) differing: %llu of %.0f\n", n, h[0], h[1], h[2], 4096.0*256*iters);
  return 0;
}
