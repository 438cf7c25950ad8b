// Does a SYNCHRONISED ascending sweep help?  One document per lane pair (or lane), all workgroups co-resident and
// started together, 32 codes per document walked 8 at a time, 32-B rows from a 4 MiB table.
//  unsorted : codes in random order   (accesses cover the whole table all the time)
//  sorted   : codes ascending         (at step k every document is in the same ~third of the table)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
template <int LPD>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ tab, const int* __restrict__ codes, int ndocs, int rpd, uint32_t* out) {
  constexpr int PB = 32 / LPD;
  const int sub = threadIdx.x % LPD;
  const int d = blockIdx.x * (256 / LPD) + threadIdx.x / LPD;
  uint32_t acc = 0;
  if (d < ndocs) {
    const int* cp = codes + (size_t)d * rpd;
    for (int t = 0; t < rpd; t += 8) {
      int c[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = cp[t + j];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint8_t* p = tab + (size_t)c[j] * 32 + sub * PB;
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        acc = max(acc, v.x ^ v.y ^ v.z ^ v.w);
        if constexpr (PB == 32) { const uint4 w = *reinterpret_cast<const uint4*>(p + 16); acc = max(acc, w.x ^ w.y ^ w.z ^ w.w); }
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int LPD>
void run(const char* name, const uint8_t* tab, const int* codes, int ndocs, int rpd, int launches) {
  uint32_t* out;
  const int blocks = (ndocs + 256 / LPD - 1) / (256 / LPD);
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((k<LPD>), dim3(blocks), dim3(256), 0, 0, tab, codes, ndocs, rpd, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double rows = (double)ndocs * rpd * launches;
  printf("%-44s docs/launch %7d blocks %5d  %.3f ms/launch  %.1f Grows/s\n", name, ndocs, blocks, best / launches, rows / best / 1e6);
  hipFree(out);
}
int main() {
  const int C = 131072, rpd = 32;
  uint8_t* tab; hipMalloc(&tab, (size_t)C * 32); hipMemset(tab, 3, (size_t)C * 32);
  for (int ndocs : {163840, 262144, 327680, 1048576}) {
    std::vector<int> h((size_t)ndocs * rpd); uint64_t s = 88172645463325252ull;
    for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (int)(s % C); }
    int* cu; hipMalloc(&cu, h.size() * 4); hipMemcpy(cu, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int d = 0; d < ndocs; ++d) std::sort(h.begin() + (size_t)d * rpd, h.begin() + (size_t)(d + 1) * rpd);
    int* cs; hipMalloc(&cs, h.size() * 4); hipMemcpy(cs, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<2>("unsorted, 2 lanes x 16B", tab, cu, ndocs, rpd, 16);
    run<2>("sorted,   2 lanes x 16B", tab, cs, ndocs, rpd, 16);
    run<1>("unsorted, 1 lane x 32B", tab, cu, ndocs, rpd, 16);
    run<1>("sorted,   1 lane x 32B", tab, cs, ndocs, rpd, 16);
    hipFree(cu); hipFree(cs);
  }
  return 0;
}
