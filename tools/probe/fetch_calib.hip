// Calibration of rocprofv3's FETCH_SIZE for the access patterns of this library (MI355X_MICROARCH.md: the counter reports half
// of a wide coalesced streaming read; "other access widths are uncalibrated").  Known byte counts:
//   stream16   : T bytes read once, 16 B per lane, coalesced                         -> T
//   gather32   : N random 32-byte rows, 2 lanes x 16 B, every row in its own 128-byte line of a table far beyond the 256 MiB
//                Infinity Cache (so every gather misses L2 and the cache)            -> N x (fabric request size)
//   gather64   : the same with 64-byte rows (4 lanes x 16 B)
//   gather256  : the same with 256-byte rows (16 lanes x 16 B; the centroid-row gathers of the MaxSim kernel)
// Run:  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o run -- ./fetch_calib.bin ; the driver script divides the counter by
// the launch's known bytes.  The kernel also prints its own wall time (rows/s x bytes must stay below the ~6.3 TB/s the fabric
// delivers, which bounds the true request size from above).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k_calib_stream16(const uint4* __restrict__ p, size_t n16, uint32_t* out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
template <int ROWB>
__global__ __launch_bounds__(256) void k_calib_gather(const uint8_t* __restrict__ tab, const uint32_t* __restrict__ idx, size_t nrows, size_t line_stride, uint32_t* out) {
  constexpr int LPR = ROWB / 16;
  const int sub = threadIdx.x % LPR;
  uint32_t acc = 0;
  for (size_t r = (size_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR; r < nrows; r += (size_t)gridDim.x * (256 / LPR)) {
    const uint4 v = *reinterpret_cast<const uint4*>(tab + (size_t)idx[r] * line_stride + sub * 16);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
static float timed(void (*launch)()) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms; }
static uint8_t* g_tab; static uint32_t* g_idx; static uint32_t* g_out; static size_t g_n, g_stride, g_bytes;
int main() {
  const size_t table = (size_t)4 << 30;            // 4 GiB
  const size_t nrows = (size_t)32 << 20;           // 32 Mi gathers
  hipMalloc(&g_tab, table); hipMemset(g_tab, 1, table);
  hipMalloc(&g_out, 64);
  std::vector<uint32_t> h(nrows); uint64_t s = 88172645463325252ull;
  printf("# name known_bytes ms\n");
  for (int stride : {128, 256, 512}) {
    const size_t lines = table / stride;
    for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)(s % lines); }
    if (!g_idx) hipMalloc(&g_idx, nrows * 4);
    hipMemcpy(g_idx, h.data(), nrows * 4, hipMemcpyHostToDevice);
    g_n = nrows; g_stride = stride;
    if (stride == 128) {
      float ms = timed([] { hipLaunchKernelGGL(k_calib_gather<32>, dim3(8192), dim3(256), 0, 0, g_tab, g_idx, g_n, g_stride, g_out); });
      printf("k_calib_gather<32> rows=%zu row_bytes=32 line_stride=128 %.3f ms %.1f Grows/s\n", nrows, ms, nrows / ms / 1e6);
      ms = timed([] { hipLaunchKernelGGL(k_calib_gather<64>, dim3(8192), dim3(256), 0, 0, g_tab, g_idx, g_n, g_stride, g_out); });
      printf("k_calib_gather<64> rows=%zu row_bytes=64 line_stride=128 %.3f ms %.1f Grows/s\n", nrows, ms, nrows / ms / 1e6);
    } else if (stride == 256) {
      float ms = timed([] { hipLaunchKernelGGL(k_calib_gather<256>, dim3(8192), dim3(256), 0, 0, g_tab, g_idx, g_n, g_stride, g_out); });
      printf("k_calib_gather<256> rows=%zu row_bytes=256 line_stride=256 %.3f ms %.1f Grows/s\n", nrows, ms, nrows / ms / 1e6);
    } else {
      float ms = timed([] { hipLaunchKernelGGL(k_calib_gather<128>, dim3(8192), dim3(256), 0, 0, g_tab, g_idx, g_n, g_stride, g_out); });
      printf("k_calib_gather<128> rows=%zu row_bytes=128 line_stride=512 %.3f ms %.1f Grows/s\n", nrows, ms, nrows / ms / 1e6);
    }
  }
  g_bytes = table;
  float ms = timed([] { hipLaunchKernelGGL(k_calib_stream16, dim3(8192), dim3(256), 0, 0, (const uint4*)g_tab, g_bytes / 16, g_out); });
  printf("k_calib_stream16 bytes=%zu %.3f ms %.2f TB/s\n", table, ms, table / ms / 1e9);
  return 0;
}
