// Gather rate of short rows from an L2-sized table (the S4 access pattern): per "document" 32 random rows,
// 8 gathers in flight, max-reduced.  Rows/s for
//  a: 64-B rows (8 MiB table at C=131072), 4 lanes x 16 B per document   (the shipped k_approx pattern)
//  b: 32-B rows (4 MiB table),            2 lanes x 16 B per document
//  c: 32-B rows (4 MiB table),            4 lanes x  8 B per document
//  d: 16-B rows (2 MiB table),            1 lane  x 16 B per document
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int ROWB, int LPD>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ tab, const int* __restrict__ codes, int ndocs, int rows_per_doc, uint32_t* out) {
  constexpr int PB = ROWB / LPD;   // bytes per lane
  const int sub = threadIdx.x % LPD;
  const int dpb = 256 / LPD;
  uint32_t acc = 0;
  for (int d = blockIdx.x * dpb + threadIdx.x / LPD; d < ndocs; d += gridDim.x * dpb) {
    const int* cp = codes + (size_t)d * rows_per_doc;
    for (int t = 0; t < rows_per_doc; t += 8) {
      int c[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = cp[t + j];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint8_t* p = tab + (size_t)c[j] * ROWB + sub * PB;
        if constexpr (PB == 16) { const uint4 v = *reinterpret_cast<const uint4*>(p); acc = max(acc, v.x ^ v.y ^ v.z ^ v.w); }
        else { const uint2 v = *reinterpret_cast<const uint2*>(p); acc = max(acc, v.x ^ v.y); }
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int ROWB, int LPD>
void run(const char* name, int C, const int* codes, int ndocs, int rpd) {
  uint8_t* tab; uint32_t* out;
  hipMalloc(&tab, (size_t)C * ROWB); hipMemset(tab, 3, (size_t)C * ROWB);
  const int blocks = 2048;
  hipMalloc(&out, blocks * 256 * 4);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<ROWB, LPD>), dim3(blocks), dim3(256), 0, 0, tab, codes, ndocs, rpd, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double rows = (double)ndocs * rpd;
  printf("%-34s C=%7d table %5.1f MiB  %.3f ms  %.1f Grows/s  %.2f TB/s useful\n", name, C, C * (double)ROWB / 1048576.0, best, rows / best / 1e6, rows * ROWB / best / 1e9);
  hipFree(tab); hipFree(out);
}
int main() {
  const int ndocs = 4 << 20, rpd = 32;
  for (int C : {131072, 65536}) {
    std::vector<int> h((size_t)ndocs * rpd); uint64_t s = 88172645463325252ull;
    for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (int)(s % C); }
    int* codes; hipMalloc(&codes, h.size() * 4); hipMemcpy(codes, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<64, 4>("a 64B rows, 4 lanes x 16B", C, codes, ndocs, rpd);
    run<32, 2>("b 32B rows, 2 lanes x 16B", C, codes, ndocs, rpd);
    run<32, 4>("c 32B rows, 4 lanes x 8B", C, codes, ndocs, rpd);
    run<16, 1>("d 16B rows, 1 lane x 16B", C, codes, ndocs, rpd);
    run<16, 2>("e 16B rows, 2 lanes x 8B", C, codes, ndocs, rpd);
    hipFree(codes);
  }
  return 0;
}
