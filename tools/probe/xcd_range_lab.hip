// Would splitting the MaxSim kernel's tokens over the 8 XCDs BY CENTROID RANGE cut its fabric traffic?  (VERDICT round 3, item 2:
// "a document's code-sorted tokens split over the 8 XCDs by centroid-id range so each L2 holds an eighth of the table".)
// Access stream only, same data as gather_lab.hip: 64 x 1024 documents x 128 tokens (codes sorted inside a document, 8 Zipf
// topics + 20 % uniform), per token a 256-byte centroid row (table 131072 x 256 B = 32 MiB: 8 x an XCD's 4 MiB L2) and 64
// residual bytes, fetched in k_maxsim6's 16x16x32 operand order (lane (r, g): chunks g, g + 4, g + 8, g + 12 of token r's row,
// 16 residual bytes).
//   base      every workgroup takes whole documents (what k_maxsim6 does)
//   range/8   workgroup i works for centroid range i % 8 (= its XCD under the round-robin dispatch): of every document it takes
//             only the tokens whose code lies in that eighth of the table (a contiguous run: codes are sorted), 16 per step
//   control   the same split with the range NOT aligned to the XCD (range = (i % 8 + i / 8) % 8): same work, no L2 affinity
//   range/16  sixteen ranges, two passes over the documents per workgroup (2 MiB of rows per pass)
// Each with plain and non-temporal residual loads.
//   hipcc -O3 --offload-arch=gfx950 xcd_range_lab.hip -o xcd_range_lab.bin && ./xcd_range_lab.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Args {
  const uint4* tab;       // [C][16]
  const int32_t* codes;   // [T] sorted inside a document
  const uint4* resid;     // [T][4]
  const int32_t* docs;    // [ND] document ids of the batch
  const uint8_t* seg;     // [ND][nr + 1] first token of each range's run
  int nd, doclen, nr, mode /*0 base, 1 range aligned, 2 range misaligned*/, nt;
  uint4* out;
};

__device__ __forceinline__ void xacc(uint4& a, const uint4 v) { a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w; }
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
// issued BEFORE the step's row loads and consumed after them: the counter is in order, so once the (younger) row loads the
// compiler waits for have returned, this one has too
__device__ __forceinline__ void ld_nt_issue(u4v& t, const uint4* p) { asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(t) : "v"(p) : "memory"); }

__global__ __launch_bounds__(1024) void k(const Args a, const int nwaves) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const int x8 = blockIdx.x & 7, j = blockIdx.x >> 3, nj = gridDim.x >> 3;
  // the waves that share a range split the documents; base: all waves split them
  int d0, d1;
  if (a.mode == 0) {
    const int nw = gridDim.x * nwaves, gw = blockIdx.x * nwaves + wave;
    d0 = (int)((long long)a.nd * gw / nw); d1 = (int)((long long)a.nd * (gw + 1) / nw);
  } else {
    const int nw = nj * nwaves, gw = j * nwaves + wave;
    d0 = (int)((long long)a.nd * gw / nw); d1 = (int)((long long)a.nd * (gw + 1) / nw);
  }
  const int passes = a.mode == 0 ? 1 : a.nr / 8;
  for (int p = 0; p < passes; ++p) {
    const int rng = a.mode == 0 ? 0 : ((a.mode == 1 ? x8 : (x8 + j) & 7) + 8 * p);
    for (int d = d0; d < d1; ++d) {
      const long long off = (long long)a.docs[d] * a.doclen;
      int s = 0, e = a.doclen;
      if (a.mode) { s = a.seg[(long long)d * (a.nr + 1) + rng]; e = a.seg[(long long)d * (a.nr + 1) + rng + 1]; }
      for (int t0 = s; t0 < e; t0 += 16) {
        int tk = t0 + r;
        tk = tk < e ? tk : e - 1;
        const long long row = off + tk;
        const int code = a.codes[row];
        const uint4* p0 = &a.tab[(long long)code * 16 + g];
        uint4 v[4];
        u4v tq = {0, 0, 0, 0};
        if (a.nt) ld_nt_issue(tq, &a.resid[row * 4 + g]);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) v[s2] = p0[s2 * 4];
        uint4 rq = make_uint4(0, 0, 0, 0);
        if (!a.nt) rq = a.resid[row * 4 + g];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) xacc(acc, v[s2]);
        if (a.nt) {
          asm volatile("" : "+v"(tq) : "v"(acc.x), "v"(acc.y), "v"(acc.z), "v"(acc.w));   // (not before the row loads have been consumed)
          rq = make_uint4(tq.x, tq.y, tq.z, tq.w);
        }
        xacc(acc, rq);
      }
    }
  }
  a.out[(long long)blockIdx.x * blockDim.x + tid] = acc;
}

// Part 2 -- centroid rows only, from precomputed code lists (no dependent loads: the split's own bookkeeping above made the
// kernel latency-bound, which says nothing about the L2s).  M2's pattern: 16 lanes per row, 32 rows per step.
//   mode 0: one list (document order), every workgroup a slice;  1: a list per range, workgroup i reads range i % 8's;
//   2: a list per range, range (i % 8 + i / 8) % 8 (control)
struct LArgs { const uint4* tab; const int32_t* list; const long long* loff; int mode; uint4* out; int nr; int cmask; const int32_t* tlist; const uint4* resid; };
__global__ __launch_bounds__(1024) void krows(const LArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x8 = blockIdx.x & 7, j = blockIdx.x >> 3, nj = gridDim.x >> 3;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int p = 0; p < (a.mode == 0 ? 1 : a.nr / 8); ++p) {
  long long b, e;
  int gw, nw;
  if (a.mode == 0) { b = a.loff[0]; e = a.loff[a.nr]; gw = blockIdx.x * 16 + wave; nw = gridDim.x * 16; }
  else { const int rng = (a.mode == 1 ? x8 : (x8 + j) & 7) + 8 * p; b = a.loff[rng]; e = a.loff[rng + 1]; gw = j * 16 + wave; nw = nj * 16; }
  const long long n = e - b, per = ((n + nw - 1) / nw + 31) & ~31ll;
  const long long i0 = b + gw * per, i1 = i0 + per < e ? i0 + per : e;
  if (p) __syncthreads();   // (a workgroup's waves change range together)
  for (long long i = i0; i < i1; i += 32) {
    const long long mine = i + (lane & 31) < i1 ? i + (lane & 31) : i1 - 1;
    const int mycode = a.list[mine] & a.cmask;   // (cmask < C - 1: a table small enough for every L2 -- is the gather bound inside the CU or behind it?)
    uint4 v[8];
    uint4 rq[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    if (a.tlist) {   // the tokens' residuals too: 64 B per token, 4 lanes each
      const int mytok = a.tlist[mine];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int tok = __shfl(mytok, 16 * q + (lane >> 2), 64);
        rq[q] = a.resid[(long long)tok * 4 + (lane & 3)];
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int code = __shfl(mycode, 4 * q + (lane >> 4), 64);
      v[q] = a.tab[(long long)code * 16 + (lane & 15)];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) xacc(acc, v[q]);
    xacc(acc, rq[0]); xacc(acc, rq[1]);
  }
  }
  a.out[(long long)blockIdx.x * blockDim.x + tid] = acc;
}

// Part 3 -- every DISTINCT centroid row of a document into LDS once (global_load_lds_dwordx4, whole rows), the tokens read
// their row from LDS in the MFMA operand order; a workgroup of 4 waves per document (2 steps of 16 tokens per wave), the rows of
// document i + 1 are in flight while document i is read (two LDS buffers of 48 rows).  gather_lab's M7 did this one document
// per WAVE with nothing in flight across documents (180 us): is the pattern itself better than that?
struct DArgs { const uint4* tab; const int32_t* ucodes; const int32_t* uoff; const uint8_t* rank; const uint4* resid; const int32_t* docs; int nd, doclen; uint4* out; };
#define DROWS 32   // (timing lab: documents with more distinct rows are clamped; a kernel would read the overflow directly)
// D = documents in flight ahead of the one being read: the rows / residuals of document it + D and the codes of document
// it + 2 D are issued in iteration it (8 vector-memory instructions per wave and iteration, always), so that everything
// iteration it needs was issued D iterations ago: s_waitcnt vmcnt(8 (D - 1)).  D + 1 LDS buffers of 32 rows.
template <int D>
__global__ __launch_bounds__(256) void kdist(const DArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [D + 1][DROWS][256 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const int n_my = (a.nd - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  auto doc_of = [&](int it) { const int d = (int)blockIdx.x + it * (int)gridDim.x; return a.docs[d < a.nd ? d : a.nd - 1]; };
  // instruction k (0 .. 7) of a document stages rows 4 k .. 4 k + 3; wave w issues k = w, w + 4
  auto load_codes = [&](int doc, int (&c)[2]) {
    const int uo = a.uoff[doc], nu = a.uoff[doc + 1] - uo;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int u = 4 * (wave + 4 * i) + g;
      u = u < nu ? u : nu - 1;
      c[i] = a.ucodes[uo + u];
    }
  };
  auto stage = [&](const int (&c)[2], int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = wave + 4 * i;
      const int slot = 4 * k + g;
      const uint4* src = a.tab + (long long)c[i] * 16 + (r ^ (slot & 7));   // piece p of slot t holds global piece p ^ (t & 7)
      const uint32_t m0 = lds0 + (uint32_t)buf * (DROWS * 256u) + (uint32_t)k * 1024u;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0), "v"(src) : "memory", "m0");
    }
  };
  auto load_tokens = [&](int doc, uint4 (&q)[2], int (&k2)[2]) {
    const long long off = (long long)doc * a.doclen;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const long long t0 = off + (wave * 2 + s2) * 16;
      q[s2] = a.resid[t0 * 4 + lane];
      k2[s2] = a.rank[t0 + r];
    }
  };
  int cq[D][2];          // codes of documents it + D .. it + 2 D - 1 (cq[0] is consumed next)
  uint4 rq[D][2];        // residuals / ranks of documents it .. it + D - 1
  int rk[D][2];
  // prologue: documents 0 .. D - 1 staged, codes of D .. 2 D - 1 loaded (in the steady-state instruction order)
#pragma unroll
  for (int j = 0; j < D; ++j) {
    int c0[2];
    load_codes(doc_of(j), c0);
    stage(c0, j);
    load_tokens(doc_of(j), rq[j], rk[j]);
    load_codes(doc_of(D + j), cq[j]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int it = 0; it < n_my; ++it) {
    if (it) {
      if constexpr (D == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if constexpr (D == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (D == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    }
    __syncthreads();   // document it's rows have landed (every wave's part); nobody still reads the buffer staged next
    uint4 rqn[2];
    int rkn[2], c2[2];
    stage(cq[0], (it + D) % (D + 1));
    load_tokens(doc_of(it + D), rqn, rkn);
    load_codes(doc_of(it + 2 * D), c2);
    const uint4* wl = reinterpret_cast<const uint4*>(smem + (size_t)(it % (D + 1)) * (DROWS * 256));
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      int slot = rk[0][s2];
      slot = slot < DROWS ? slot : DROWS - 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) xacc(acc, wl[slot * 16 + ((4 * q + g) ^ (slot & 7))]);
      xacc(acc, rq[0][s2]);
    }
#pragma unroll
    for (int j = 0; j + 1 < D; ++j)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) { cq[j][s2] = cq[j + 1][s2]; rq[j][s2] = rq[j + 1][s2]; rk[j][s2] = rk[j + 1][s2]; }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) { cq[D - 1][s2] = c2[s2]; rq[D - 1][s2] = rqn[s2]; rk[D - 1][s2] = rkn[s2]; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  a.out[((long long)blockIdx.x * blockDim.x + tid) & 0x3FFFF] = acc;
}

static uint64_t rng_s = 88172645463325252ull;
static inline uint64_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }

int main() {
  const int C = 131072, NDOCS = 1000000, L = 128, ND = 65536;
  const long long T = (long long)NDOCS * L;
  std::vector<int32_t> codes((size_t)T);
  for (int d = 0; d < NDOCS; ++d) {
    int topic[8];
    for (int i = 0; i < 8; ++i) {
      const uint64_t r = rnd();
      const int e = (int)(r % 17);
      const uint64_t rank = ((1ull << e) - 1) + ((r >> 8) & ((1ull << e) - 1));
      topic[i] = (int)((rank * 0x9E3779B1ull + 12345) & (C - 1));
    }
    int32_t* c = &codes[(size_t)d * L];
    for (int t = 0; t < L; ++t) {
      const uint64_t r = rnd();
      c[t] = ((r & 0xFF) < 205) ? topic[(r >> 8) & 7] : (int)((r >> 16) & (C - 1));
    }
    std::sort(c, c + L);
  }
  std::vector<int32_t> docs(ND);
  for (int q = 0; q < 64; ++q) {
    std::vector<int32_t> l(1024);
    for (auto& x : l) x = (int32_t)(rnd() % NDOCS);
    std::sort(l.begin(), l.end());
    std::copy(l.begin(), l.end(), docs.begin() + q * 1024);
  }
  uint4* tab; int32_t* dcodes; uint4* resid; int32_t* ddocs; uint4* out; uint8_t* dseg;
  CHK(hipMalloc(&tab, (size_t)C * 256)); CHK(hipMemset(tab, 1, (size_t)C * 256));
  CHK(hipMalloc(&dcodes, (size_t)T * 4)); CHK(hipMemcpy(dcodes, codes.data(), (size_t)T * 4, hipMemcpyHostToDevice));
  CHK(hipMalloc(&resid, (size_t)T * 64)); CHK(hipMemset(resid, 2, (size_t)T * 64));
  CHK(hipMalloc(&ddocs, ND * 4)); CHK(hipMemcpy(ddocs, docs.data(), ND * 4, hipMemcpyHostToDevice));
  CHK(hipMalloc(&out, (size_t)256 * 1024 * 16));
  CHK(hipMalloc(&dseg, (size_t)ND * 17));
  Args a{};
  a.tab = tab; a.codes = dcodes; a.resid = resid; a.docs = ddocs; a.nd = ND; a.doclen = L; a.out = out; a.seg = dseg;
  for (int nr : {8, 16}) {
    std::vector<uint8_t> seg((size_t)ND * (nr + 1));
    long long steps = 0;
    for (int d = 0; d < ND; ++d) {
      const int32_t* c = &codes[(size_t)docs[d] * L];
      for (int x = 0; x <= nr; ++x) seg[(size_t)d * (nr + 1) + x] = (uint8_t)(std::lower_bound(c, c + L, (int)((long long)x * C / nr)) - c);
      for (int x = 0; x < nr; ++x) steps += (seg[(size_t)d * (nr + 1) + x + 1] - seg[(size_t)d * (nr + 1) + x] + 15) / 16;
    }
    CHK(hipMemcpy(dseg, seg.data(), seg.size(), hipMemcpyHostToDevice));
    printf("## %d ranges: %.2f 16-token steps per document (8 without the split): lane utilisation %.2f\n", nr, (double)steps / ND, 8.0 * ND / steps);
    a.nr = nr;
    for (int mode : {0, 1, 2})
      for (int nt : {0, 1}) {
        if (nr == 16 && mode == 0) continue;
        a.mode = mode; a.nt = nt;
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
          CHK(hipEventRecord(e0));
          hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, a, 16);
          CHK(hipGetLastError());
          CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
          float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
          if (rep) best = std::min(best, ms);
        }
        printf("%-28s residual loads %-5s: %7.1f us\n", mode == 0 ? "base (whole documents)" : (mode == 1 ? "range = XCD" : "range != XCD (control)"), nt ? "nt" : "plain", best * 1e3);
      }
  }
    // ---- part 3: distinct rows through LDS, one document per workgroup, next document's rows in flight
  {
    std::vector<int32_t> ucodes; std::vector<int32_t> uoff(NDOCS + 1, 0); std::vector<uint8_t> rank((size_t)T);
    for (int d = 0; d < NDOCS; ++d) {
      const int32_t* c = &codes[(size_t)d * L];
      int n = 0;
      for (int t = 0; t < L; ++t) {
        if (t == 0 || c[t] != c[t - 1]) { ucodes.push_back(c[t]); ++n; }
        rank[(size_t)d * L + t] = (uint8_t)(n - 1);
      }
      uoff[d + 1] = uoff[d] + n;
    }
    int32_t *du, *duo; uint8_t* drk;
    CHK(hipMalloc(&du, ucodes.size() * 4 + 64)); CHK(hipMemcpy(du, ucodes.data(), ucodes.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMalloc(&duo, uoff.size() * 4)); CHK(hipMemcpy(duo, uoff.data(), uoff.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMalloc(&drk, rank.size())); CHK(hipMemcpy(drk, rank.data(), rank.size(), hipMemcpyHostToDevice));
    DArgs da{tab, du, duo, drk, resid, ddocs, ND, L, out};
    printf("## distinct rows (%.1f per document) -> LDS once per document, 4 waves per document, D documents in flight\n", (double)ucodes.size() / NDOCS);
    CHK(hipFuncSetAttribute((const void*)kdist<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    for (int D : {1, 2, 3, 4})
    for (int per_cu : {3, 4, 5, 6, 8}) {
      const size_t lds = (size_t)(D + 1) * DROWS * 256;
      if (lds * per_cu > 160 * 1024) continue;
      const int grid = 256 * per_cu;
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0));
        if (D == 1) hipLaunchKernelGGL(kdist<1>, dim3(grid), dim3(256), lds, 0, da);
        else if (D == 2) hipLaunchKernelGGL(kdist<2>, dim3(grid), dim3(256), lds, 0, da);
        else if (D == 3) hipLaunchKernelGGL(kdist<3>, dim3(grid), dim3(256), lds, 0, da);
        else hipLaunchKernelGGL(kdist<4>, dim3(grid), dim3(256), lds, 0, da);
        CHK(hipGetLastError());
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) best = std::min(best, ms);
      }
      printf("D = %d, %d workgroups per CU (%2zu KiB of LDS each): %7.1f us\n", D, per_cu, lds / 1024, best * 1e3);
    }
  }
  // ---- part 2: rows only, precomputed lists
  {
   for (int NR : {8, 16, 32, 64}) {
    std::vector<int32_t> l0, t0v; l0.reserve((size_t)ND * L); t0v.reserve((size_t)ND * L);
    std::vector<std::vector<int32_t>> lr(NR), tr(NR);
    for (int d = 0; d < ND; ++d) {
      const int32_t* c = &codes[(size_t)docs[d] * L];
      for (int t = 0; t < L; ++t) {
        const int32_t tok = (int32_t)((long long)docs[d] * L + t);
        l0.push_back(c[t]); t0v.push_back(tok);
        lr[c[t] / (C / NR)].push_back(c[t]); tr[c[t] / (C / NR)].push_back(tok);
      }
    }
    std::vector<int32_t> lcat, tcat; std::vector<long long> loff(NR + 1, 0);
    for (int x = 0; x < NR; ++x) { lcat.insert(lcat.end(), lr[x].begin(), lr[x].end()); tcat.insert(tcat.end(), tr[x].begin(), tr[x].end()); loff[x + 1] = (long long)lcat.size(); }
    int32_t *dl0, *dlr, *dt0, *dtr; long long* dloff;
    CHK(hipMalloc(&dl0, l0.size() * 4)); CHK(hipMemcpy(dl0, l0.data(), l0.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMalloc(&dlr, lcat.size() * 4)); CHK(hipMemcpy(dlr, lcat.data(), lcat.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMalloc(&dt0, t0v.size() * 4)); CHK(hipMemcpy(dt0, t0v.data(), t0v.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMalloc(&dtr, tcat.size() * 4)); CHK(hipMemcpy(dtr, tcat.data(), tcat.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMalloc(&dloff, (NR + 1) * 8));
    printf("## centroid rows only, %zu rows from precomputed code lists (M2's pattern), %d ranges of %d KiB of rows\n", l0.size(), NR, C / NR / 4);
    for (int wr : {0, 1})
    for (int mode : {0, 1, 2}) {
      if (mode == 0 && NR != 8) continue;
      std::vector<long long> lo = loff;
      if (mode == 0) { for (auto& v : lo) v = 0; lo[NR] = (long long)l0.size(); }
      CHK(hipMemcpy(dloff, lo.data(), (NR + 1) * 8, hipMemcpyHostToDevice));
      LArgs la{tab, mode == 0 ? dl0 : dlr, dloff, mode, out, NR, C - 1, wr ? (mode == 0 ? dt0 : dtr) : nullptr, resid};
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(krows, dim3(256), dim3(1024), 0, 0, la);
        CHK(hipGetLastError());
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) best = std::min(best, ms);
      }
      printf("%s%-44s %7.1f us\n", wr ? "rows + residuals: " : "rows only:        ", mode == 0 ? "document order, all workgroups" : (mode == 1 ? "per-range lists, range = XCD" : "per-range lists, range != XCD (control)"), best * 1e3);
    }
    if (NR == 8)
      for (int cm : {C / 2 - 1, C / 8 - 1, C / 32 - 1, 1023, 63}) {
        std::vector<long long> lo(NR + 1, 0); lo[NR] = (long long)l0.size();
        CHK(hipMemcpy(dloff, lo.data(), (NR + 1) * 8, hipMemcpyHostToDevice));
       for (int wr : {0, 1}) {
        LArgs la{tab, dl0, dloff, 0, out, NR, cm, wr ? dt0 : nullptr, resid};
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
          hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
          CHK(hipEventRecord(e0));
          hipLaunchKernelGGL(krows, dim3(256), dim3(1024), 0, 0, la);
          CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
          float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
          if (rep) best = std::min(best, ms);
        }
        printf("%sdocument order, codes folded into a table of %6d rows (%5d KiB): %7.1f us\n", wr ? "rows + residuals: " : "rows only:        ", cm + 1, (cm + 1) / 4, best * 1e3);
       }
      }
    (void)hipFree(dl0); (void)hipFree(dlr); (void)hipFree(dloff); (void)hipFree(dt0); (void)hipFree(dtr);
   }
  }
  return 0;
}
