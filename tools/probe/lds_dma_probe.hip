// Does global_load_lds_dwordx4 put lane l's 16 bytes at LDS address M0 + 16 l?  (k_centroid_scores_stream relies on it.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint4* __restrict__ g, uint4* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t voff = (uint32_t)(threadIdx.x ^ 5u) * 16u;   // a permuted source piece per lane
  const uint32_t m0v = lds0 + 4096u + wave * 1024u;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(g), "s"(m0v) : "memory", "m0");
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  out[threadIdx.x] = *reinterpret_cast<uint4*>(smem + 4096 + threadIdx.x * 16);
}
int main() {
  const int n = 256;
  std::vector<uint4> h(n), o(n);
  for (int i = 0; i < n; ++i) h[i] = make_uint4(i, i * 3 + 1, ~i, i * 7);
  uint4 *d, *dout;
  hipMalloc(&d, n * 16); hipMalloc(&dout, n * 16);
  hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(n), 16384, 0, d, dout);
  hipMemcpy(o.data(), dout, n * 16, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) { const uint4 e = h[i ^ 5]; if (o[i].x != e.x || o[i].y != e.y || o[i].z != e.z || o[i].w != e.w) ++bad; }
  printf("lds_dma_probe: %s (%d mismatches of %d)\n", bad ? "FAIL" : "OK", bad, n);
  return bad != 0;
}
