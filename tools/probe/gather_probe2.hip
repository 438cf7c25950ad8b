// How does the L1/TA cost of a 16B-per-lane gather depend on which lanes share a cache line?
// Every variant fetches, per wave iteration, 64 lanes x 16 B = 1 KiB out of 256-B rows of a 32 MiB table.
//  A: 16 rows/instr, 4 ADJACENT lanes read one row's 64-B piece      (lane>>2 = row slot, lane&3 = 16-B chunk)
//  B: 16 rows/instr, the 4 lanes of a row are 16 lanes apart          (lane&15 = row slot, lane>>4 = chunk)
//  C: 64 rows/instr, every lane reads 16 B of its own row             (one token per lane)
//  D: 4 rows/instr, 16 adjacent lanes read a whole 256-B row
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ tab, const int* __restrict__ rows, int nrows_mask, int iters, int mode, uint4* out) {
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint4 acc = make_uint4(0,0,0,0);
  for (int it = 0; it < iters; ++it) {
    const int base = (gw * iters + it) * 64;
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {   // 16 instrs = 16 KiB per wave per iteration in every mode
      int slot, chunk;
      if (mode == 0) { slot = (lane >> 2) + 16 * (j & 3); chunk = (lane & 3) + 4 * (j >> 2); }
      else if (mode == 1) { slot = (lane & 15) + 16 * (j & 3); chunk = (lane >> 4) + 4 * (j >> 2); }
      else if (mode == 2) { slot = lane; chunk = j; }
      else { slot = (lane >> 4) + 4 * j; chunk = lane & 15; }
      const int row = rows[(base + slot) & nrows_mask];
      const uint4 v = tab[(size_t)row * 16 + chunk];
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int run(int C) { const size_t tb = (size_t)C * 256;
  uint4* tab; int* rows; uint4* out;
  hipMalloc(&tab, tb); hipMemset(tab, 1, tb);
  const int NR = 1 << 22; std::vector<int> h(NR); uint64_t s = 88172645463325252ull;
  for (int i = 0; i < NR; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s % C); }
  hipMalloc(&rows, NR * 4); hipMemcpy(rows, h.data(), NR * 4, hipMemcpyHostToDevice);
  const int blocks = 256 * 12 / 4 * 4, iters = 16;   // 3072 blocks x 4 waves
  hipMalloc(&out, (size_t)blocks * 256 * 16);
  const char* names[4] = {"A adjacent4/64B", "B stride16/64B", "C lane=row", "D adjacent16/256B"};
  printf("table %.1f MiB\n", tb / 1048576.0);
  for (int rep = 0; rep < 2; ++rep)
  for (int mode = 0; mode < 4; mode += 2) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, tab, rows, NR - 1, iters, mode, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bytes = (double)blocks * 4 * iters * 16 * 1024;
    if (rep) printf("%-18s %.3f ms  %.1f GB/s  (%.2f B/clk/CU @2.25GHz)\n", names[mode], ms, bytes / ms / 1e6, bytes / (ms * 1e-3) / 256 / 2.25e9);
  }
  hipFree(tab); hipFree(rows); hipFree(out);
  return 0;
}
int main() { for (int C : {4096, 8192, 16384, 32768, 131072, 1048576}) run(C); return 0; }
