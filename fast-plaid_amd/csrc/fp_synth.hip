// Device-side synthetic corpus generator (benchmark / full-size property tests) and IVF
// builder.  Bit-identical to the numpy twin in fast-plaid_amd/synth.py: every integer is a
// pure function of (seed, stream, counter) through the splitmix64 finaliser.
#include <hipcub/hipcub.hpp>

#include "fp_internal.h"

#define S_DOCLEN 1ull
#define S_TOPIC 2ull
#define S_TOKEN 3ull
#define S_RESID 4ull
#define TOPIC_SIZE 8
#define P_TOPIC_256 205ull

__host__ __device__ static inline uint64_t mix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ static inline uint64_t stream_key(uint64_t seed, uint64_t stream) {
  return mix64(seed * 0xD1342543DE82EF95ull + stream);
}
__host__ __device__ static inline int synth_doc_len(uint64_t key_len, int doc_len, int variable, uint64_t pid) {
  if (!variable) return doc_len;
  int lo = doc_len / 4 > 1 ? doc_len / 4 : 1;
  uint64_t span = (uint64_t)(doc_len - lo + 1);
  return lo + (int)(mix64(key_len + pid) % span);
}

int64_t fps_doc_offsets_host(const FpSynthParams& p, int64_t* doc_off_host, int* max_len) {
  // doc_off_host: [n_local + 2]; the extra last slot returns tok_base (tokens before doc_begin)
  const uint64_t kl = stream_key(p.seed, S_DOCLEN);
  int64_t tok_base = 0;
  if (p.variable_len) {
    for (int64_t d = 0; d < p.doc_begin; ++d) tok_base += synth_doc_len(kl, p.doc_len, 1, (uint64_t)d);
  } else {
    tok_base = p.doc_begin * (int64_t)p.doc_len;
  }
  int64_t n = p.doc_end - p.doc_begin, a = 0;
  int mx = 0;
  for (int64_t i = 0; i < n; ++i) {
    doc_off_host[i] = a;
    int l = synth_doc_len(kl, p.doc_len, p.variable_len, (uint64_t)(p.doc_begin + i));
    if (l > mx) mx = l;
    a += l;
  }
  doc_off_host[n] = a;
  doc_off_host[n + 1] = tok_base;
  *max_len = mx;
  return a;
}

__device__ static inline uint32_t zipf_centroid(uint64_t r, int lgC, uint64_t Cmask) {
  uint64_t e = r % (uint64_t)lgC;
  uint64_t m = (1ull << e) - 1ull;
  uint64_t rank = m + ((r >> 8) & m);
  return (uint32_t)((rank * 0x9E3779B1ull + 12345ull) & Cmask);
}

// one wave per document
__global__ __launch_bounds__(256) void k_synth_generate(FpSynthParams p, const int64_t* __restrict__ doc_off, int64_t n_docs,
                                                        int64_t tok_base, int32_t* __restrict__ codes,
                                                        uint8_t* __restrict__ residuals) {
  const int lane = threadIdx.x & 63;
  const int64_t dstep = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t d = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; d < n_docs; d += dstep) {
  const uint64_t pid = (uint64_t)(p.doc_begin + d);
  const uint64_t k_topic = stream_key(p.seed, S_TOPIC), k_tok = stream_key(p.seed, S_TOKEN), k_res = stream_key(p.seed, S_RESID);
  const uint64_t Cmask = (uint64_t)p.C - 1ull;
  uint32_t topics[TOPIC_SIZE];
#pragma unroll
  for (int j = 0; j < TOPIC_SIZE; ++j) topics[j] = zipf_centroid(mix64(k_topic + pid * TOPIC_SIZE + j), p.lgC, Cmask);
  const int64_t off = doc_off[d];
  const int len = (int)(doc_off[d + 1] - off);
  const int words = (p.pr + 7) / 8;
  for (int t = lane; t < len; t += 64) {
    const uint64_t tg = (uint64_t)(tok_base + off + t);
    const uint64_t r = mix64(k_tok + tg);
    uint32_t code;
    if ((r & 0xFFull) < P_TOPIC_256) {
      const int slot = (int)((r >> 8) & (TOPIC_SIZE - 1));
      code = topics[0];
#pragma unroll
      for (int j = 1; j < TOPIC_SIZE; ++j) code = (slot == j) ? topics[j] : code;
    } else {
      code = (uint32_t)((r >> 16) & Cmask);
    }
    codes[off + t] = (int32_t)code;
    uint8_t* rp = residuals + (off + t) * (int64_t)p.pr;
    for (int w = 0; w < words; ++w) {
      const uint64_t v = mix64(k_res + tg * (uint64_t)words + (uint64_t)w);
      if ((w + 1) * 8 <= p.pr) {
        *reinterpret_cast<uint64_t*>(rp + 8 * w) = v;  // little-endian, pr % 8 == 0 on supported shapes
      } else {
        for (int k = 0; 8 * w + k < p.pr; ++k) rp[8 * w + k] = (uint8_t)(v >> (8 * k));
      }
    }
  }
  }
}

void fps_generate(const FpSynthParams& p, const int64_t* doc_off_dev, int64_t n_docs, int64_t T, int64_t tok_base, int32_t* codes,
                  uint8_t* residuals, hipStream_t st) {
  (void)T;
  if (n_docs <= 0) return;
  const int64_t blocks = (n_docs + 3) / 4;
  hipLaunchKernelGGL(k_synth_generate, dim3(fp_grid_cap(blocks, 256)), dim3(256), 0, st, p, doc_off_dev, n_docs, tok_base, codes, residuals);
}

// ---- in-document token sort by centroid code -------------------------------------------------------
#define SORT_MAX_LEN 4096
__global__ __launch_bounds__(64) void k_doc_sort(int32_t* __restrict__ codes, const int64_t* __restrict__ doc_off, int64_t n_docs,
                                                 uint16_t* __restrict__ perm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* v = reinterpret_cast<unsigned long long*>(smem);
  const int lane = threadIdx.x;
  for (int64_t d = blockIdx.x; d < n_docs; d += gridDim.x) {
  const int64_t off = doc_off[d];
  const int n = (int)(doc_off[d + 1] - off);
  if (n > SORT_MAX_LEN) {
    for (int i = lane; i < n; i += 64) perm[off + i] = (uint16_t)(i & 0xFFFF);  // not sorted: identity (only valid for n < 65536)
    continue;
  }
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (int i = lane; i < np2; i += 64)
    v[i] = (i < n) ? (((unsigned long long)(uint32_t)codes[off + i] << 32) | (unsigned long long)i) : ~0ull;
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = lane; i < np2; i += 64) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = v[i], c = v[ixj];
          const bool up = ((i & k) == 0);
          if ((a > c) == up) { v[i] = c; v[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = lane; i < n; i += 64) {
    codes[off + i] = (int32_t)(v[i] >> 32);
    perm[off + i] = (uint16_t)(v[i] & 0xFFFF);
  }
  __syncthreads();  // the next document reuses the LDS buffer
  }
}

// new_res[off + i] = old_res[off + perm[i]]; one piece per thread: 16 bytes when the rows are multiples of 16 bytes, 4 bytes when
// they are multiples of 4, else the whole row byte by byte
__host__ __device__ static inline int permute_piece_bytes(int pr) { return pr % 16 == 0 ? 16 : (pr % 4 == 0 ? 4 : pr); }
__global__ void k_permute_rows(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const uint16_t* __restrict__ perm,
                               const int64_t* __restrict__ doc_off, int64_t n_docs, int64_t T, int pr) {
  const int pb = permute_piece_bytes(pr);
  const int ppr = pr / pb;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < T * ppr; g += (int64_t)gridDim.x * blockDim.x) {
  const int64_t row = g / ppr;
  const int piece = (int)(g % ppr);
  // document of `row`: binary search over doc_off
  int64_t lo = 0, hi = n_docs;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (doc_off[mid] <= row) lo = mid; else hi = mid;
  }
  const int64_t off = doc_off[lo];
  const int64_t srow = off + perm[row];
  if (pb == 16) {
    *reinterpret_cast<uint4*>(dst + row * pr + piece * 16) = *reinterpret_cast<const uint4*>(src + srow * pr + piece * 16);
  } else if (pb == 4) {
    *reinterpret_cast<uint32_t*>(dst + row * pr + piece * 4) = *reinterpret_cast<const uint32_t*>(src + srow * pr + piece * 4);
  } else {
    for (int b = 0; b < pr; ++b) dst[row * pr + b] = src[srow * pr + b];
  }
  }
}

#define HCHK0(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc0 = (int)e_; goto fail0; } } while (0)
int fps_sort_docs(int32_t* codes, uint8_t* residuals, const int64_t* doc_off_dev, int64_t n_docs, int64_t T, int max_len, int pr,
                  uint16_t** perm, hipStream_t st) {
  int rc0 = 0;
  uint8_t* tmp = nullptr;
  int np2 = 1;
  *perm = nullptr;
  HCHK0(hipMalloc((void**)perm, (size_t)(T > 0 ? T : 1) * 2 + 64));
  if (n_docs <= 0 || T <= 0) return 0;
  if (max_len >= 65536) {  // u16 positions cannot describe such documents: keep original order everywhere
    (void)hipFree(*perm);
    *perm = nullptr;
    return 0;
  }
  while (np2 < max_len && np2 < SORT_MAX_LEN) np2 <<= 1;
  hipLaunchKernelGGL(k_doc_sort, dim3(fp_grid_cap(n_docs, 64)), dim3(64), (size_t)np2 * 8, st, codes, doc_off_dev, n_docs, *perm);
  HCHK0(hipMalloc((void**)&tmp, (size_t)T * pr + 64));
  {
    const int ppr = pr / permute_piece_bytes(pr);
    const int64_t threads = T * ppr;
    hipLaunchKernelGGL(k_permute_rows, dim3(fp_grid_cap((threads + 255) / 256, 256)), dim3(256), 0, st, residuals, tmp, *perm, doc_off_dev, n_docs,
                       T, pr);
  }
  HCHK0(hipMemcpyAsync(residuals, tmp, (size_t)T * pr, hipMemcpyDeviceToDevice, st));
  HCHK0(hipStreamSynchronize(st));
fail0:
  if (tmp) (void)hipFree(tmp);
  return rc0;
}

// ---- per-document sorted unique codes ------------------------------------------------------------
// The approximate score sum_q max_t S[code_t, q] only depends on the SET of codes of a document
// (max is idempotent), so the index keeps, next to the raw per-token codes the exact stage
// needs, a deduplicated ascending code list per document: fewer 64-byte score rows to gather
// in S4 (the stage that dominates the batch time) and exactly the (cell, doc) pairs of the IVF.
// One wave per document: LDS bitonic sort (padded to a power of two), then ordered unique.
// Documents longer than UNIQ_MAX_LEN are copied as they are (duplicates are harmless).
#define UNIQ_MAX_LEN 8192
template <bool WRITE>
__global__ __launch_bounds__(64) void k_doc_unique(const int32_t* __restrict__ codes, const int64_t* __restrict__ doc_off,
                                                   int64_t n_docs, int64_t* __restrict__ ulen, const int64_t* __restrict__ uoff,
                                                   int32_t* __restrict__ ucodes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int32_t* v = reinterpret_cast<int32_t*>(smem);
  const int lane = threadIdx.x;
  for (int64_t d = blockIdx.x; d < n_docs; d += gridDim.x) {
  const int64_t off = doc_off[d];
  const int n = (int)(doc_off[d + 1] - off);
  if (n > UNIQ_MAX_LEN) {
    if (WRITE) { for (int i = lane; i < n; i += 64) ucodes[uoff[d] + i] = codes[off + i]; }
    else if (lane == 0) ulen[d] = n;
    continue;
  }
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (int i = lane; i < np2; i += 64) v[i] = (i < n) ? codes[off + i] : 0x7FFFFFFF;
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = lane; i < np2; i += 64) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const int32_t a = v[i], c = v[ixj];
          const bool up = ((i & k) == 0);
          if ((a > c) == up) { v[i] = c; v[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
  int base = 0;
  const int64_t wbase = WRITE ? uoff[d] : 0;
  for (int start = 0; start < n; start += 64) {
    const int i = start + lane;
    const bool flag = (i < n) && (i == 0 || v[i] != v[i - 1]);
    const unsigned long long m = __ballot(flag);
    if (WRITE && flag) {
      const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      ucodes[wbase + base + __popcll(m & below)] = v[i];
    }
    base += __popcll(m);
  }
  if (!WRITE && lane == 0) ulen[d] = base;
  __syncthreads();  // the next document reuses the LDS buffer
  }
}

#define HCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = (int)e_; goto fail; } } while (0)

int fps_build_ucodes(const int32_t* codes, const int64_t* doc_off_dev, int64_t n_docs, int max_len, int32_t** ucodes, int64_t* U,
                     int64_t* uoff_dev, hipStream_t st) {
  int rc = 0;
  int64_t* ulen = nullptr;
  void* tmp = nullptr;
  size_t tb = 0;
  int np2 = 1;
  size_t lds = 0;
  *ucodes = nullptr;
  *U = 0;
  if (n_docs <= 0) {
    HCHK(hipMemsetAsync(uoff_dev, 0, 8, st));
    HCHK(hipMalloc((void**)ucodes, 16));
    return 0;
  }
  while (np2 < max_len && np2 < UNIQ_MAX_LEN) np2 <<= 1;
  lds = (size_t)np2 * 4;
  HCHK(hipMalloc((void**)&ulen, (size_t)(n_docs + 1) * 8));
  HCHK(hipMemsetAsync(ulen, 0, (size_t)(n_docs + 1) * 8, st));
  hipLaunchKernelGGL(k_doc_unique<false>, dim3(fp_grid_cap(n_docs, 64)), dim3(64), lds, st, codes, doc_off_dev, n_docs, ulen, nullptr, nullptr);
  HCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, ulen, uoff_dev, (int)(n_docs + 1), st));
  HCHK(hipMalloc(&tmp, tb ? tb : 16));
  HCHK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, ulen, uoff_dev, (int)(n_docs + 1), st));
  HCHK(hipMemcpyAsync(U, uoff_dev + n_docs, 8, hipMemcpyDeviceToHost, st));
  HCHK(hipStreamSynchronize(st));
  HCHK(hipMalloc((void**)ucodes, (size_t)(*U > 0 ? *U : 4) * 4 + 64));
  hipLaunchKernelGGL(k_doc_unique<true>, dim3(fp_grid_cap(n_docs, 64)), dim3(64), lds, st, codes, doc_off_dev, n_docs, nullptr, uoff_dev, *ucodes);
  HCHK(hipStreamSynchronize(st));
fail:
  if (ulen) (void)hipFree(ulen);
  if (tmp) (void)hipFree(tmp);
  return rc;
}

// ---- packed unique codes for S4's level-0 scan ---------------------------------------------------------
// The level-0 kernel reads every candidate's unique-code list once per query; as int32 lists at arbitrary offsets (133 B on
// the benchmark corpus) each candidate touched 2-3 128-byte lines, most of whose bytes belonged to non-candidates (6.7 GB
// requested, 5.3 GB over the fabric for 2.8 GB of codes).  Here every document owns whole 128-byte LINES: a line is 8 pieces
// of 16 bytes, a piece = 6 codes of 20 bits (bits 0..119) + the number of codes in the piece (bits 120..127).  48 codes per
// line (one line per document on the benchmark corpus: 33 codes on average, 0.15 % of the documents need a second line), any
// number of lines per document; a lane reads ONE aligned 16-byte piece and needs nothing from its neighbours.  Codes are
// relative to a range of 2^17 centroids (one set of lines per range), so 17 of the 20 bits are used.
#define PCODES_PER_PIECE 6
#define PCODES_PER_LINE 48
// sub-run of a document's ascending unique codes that falls in [lo, hi): rs[d] .. re[d] (positions in ucodes)
__global__ void k_pcode_range(const int32_t* __restrict__ ucodes, const int64_t* __restrict__ uoff, int64_t n_docs, int32_t lo, int32_t hi,
                              int64_t* __restrict__ rs, int64_t* __restrict__ re) {
  for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n_docs; d += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = uoff[d], e = uoff[d + 1];
    int64_t a0 = b, a1 = e;
    while (a0 < a1) { const int64_t m = (a0 + a1) >> 1; if (ucodes[m] < lo) a0 = m + 1; else a1 = m; }
    int64_t z0 = a0, z1 = e;
    while (z0 < z1) { const int64_t m = (z0 + z1) >> 1; if (ucodes[m] < hi) z0 = m + 1; else z1 = m; }
    rs[d] = a0;
    re[d] = z0;
  }
}
// lines beyond a document's first ("extra" lines: documents with more codes in the range than one line holds)
__global__ void k_pcode_lines(const int64_t* __restrict__ rs, const int64_t* __restrict__ re, int64_t n_docs, int64_t* __restrict__ nextra, int cpl /*codes per line*/) {
  for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d <= n_docs; d += (int64_t)gridDim.x * blockDim.x) {
    const int64_t nl = d < n_docs ? (re[d] - rs[d] + cpl - 1) / cpl : 0;
    nextra[d] = nl > 1 ? nl - 1 : 0;
  }
}
// Layout: line d (d < n_docs) is document d's FIRST line -- the scan addresses it by the document id, no lookup -- and the extra
// lines of the documents that need them follow behind, contiguous per document; poff[d] = {n_docs + first extra line, count}.
// A document with extra lines carries bit 7 in the count byte of its first line's last piece.
__global__ __launch_bounds__(256) void k_pcode_pack(const int32_t* __restrict__ ucodes, const int64_t* __restrict__ rs,
                                                    const int64_t* __restrict__ re, int64_t n_docs, int32_t code_base,
                                                    const int64_t* __restrict__ xoff /*[n_docs + 1] exclusive scan of the extra-line counts*/,
                                                    int32_t* __restrict__ poff, uint4* __restrict__ lines,
                                                    int ppl /*pieces per line: 8 or 4*/, uint32_t pad_code, uint4* __restrict__ shared_first, int nr, int r) {
  // one thread per piece.  shared_first != nullptr (tables of several ranges): the first lines go to shared_first, line
  // d * nr + r, `lines` holds the extra lines only and poff's first index counts from 0
  const int64_t nl = n_docs + xoff[n_docs];
  const int64_t xbase = shared_first ? 0 : n_docs;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < nl * ppl + n_docs + 1; g += (int64_t)gridDim.x * blockDim.x) {
    if (g >= nl * ppl) {   // per document {first extra line, extra line count}
      const int64_t d = g - nl * ppl;
      if (d < n_docs) {
        poff[2 * d] = (int32_t)(xbase + xoff[d]);
        poff[2 * d + 1] = (int32_t)(xoff[d + 1] - xoff[d]);
      }
      continue;
    }
    const int64_t line = g / ppl;
    const int piece = (int)(g % ppl);
    int64_t doc = line, lineno = 0;
    if (line >= n_docs) {   // an extra line: its document is the last d with xoff[d] <= x
      const int64_t x = line - n_docs;
      int64_t lo = 0, hi = n_docs;
      while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (xoff[mid] <= x) lo = mid; else hi = mid;
      }
      doc = lo;
      lineno = 1 + (x - xoff[lo]);
    }
    const int64_t u0 = rs[doc], n = re[doc] - u0;
    const int64_t first = lineno * (PCODES_PER_PIECE * ppl) + (int64_t)piece * PCODES_PER_PIECE;
    // piece = 6 codes of 20 bits (bits 0..119) + the count in bits 120..127.  Slots past the piece's last code hold pad_code:
    // the index of a byte behind the range's table slice that the scan keeps zero, so that all six lookups can be summed
    // without a predicate or a correction
    unsigned __int128 bits = 0;
    uint32_t cnt = 0;
    for (int j = 0; j < PCODES_PER_PIECE; ++j) {
      uint32_t c = pad_code;
      if (first + j < n) {
        c = (uint32_t)(ucodes[u0 + first + j] - code_base);
        ++cnt;
      }
      bits |= (unsigned __int128)(c & 0xFFFFFu) << (20 * j);
    }
    if (lineno == 0 && piece == ppl - 1 && xoff[doc + 1] > xoff[doc]) cnt |= 0x80u;   // "this document has extra lines"
    bits |= (unsigned __int128)cnt << 120;
    uint32_t w[4];
    for (int k = 0; k < 4; ++k) w[k] = (uint32_t)(bits >> (32 * k));
    const uint4 out = make_uint4(w[0], w[1], w[2], w[3]);
    if (!shared_first) lines[g] = out;
    else if (line < n_docs) shared_first[(line * nr + r) * ppl + piece] = out;
    else lines[(line - n_docs) * ppl + piece] = out;
  }
}

// The unique codes in [code_lo, code_hi) (at most 2^17 values) of every document, packed into 128-byte (ppl 8) or 64-byte (ppl 4)
// lines.  Allocates *lines ([n_docs first lines + the extra lines]) and *poff ([N] x {first extra line, extra line count} i32
// pairs); *n_lines = lines in all; returns 0 or a hipError
int fps_build_pcodes(const int32_t* ucodes, const int64_t* uoff_dev, int64_t n_docs, int64_t code_lo, int64_t code_hi, void** lines,
                     int32_t** poff, int64_t* n_lines, hipStream_t st, int ppl, int64_t n_centroids, void* shared_first, int nr, int r) {
  int rc = 0;
  int64_t* nlines = nullptr;
  int64_t* loff = nullptr;
  int64_t* rs = nullptr;
  int64_t* re = nullptr;
  void* tmp = nullptr;
  size_t tb = 0;
  *lines = nullptr;
  *poff = nullptr;
  *n_lines = 0;
  // the scan keeps 16 zero bytes behind its table slice [code_lo, min(code_lo + 2^17, Cpad)): the pad slots point there
  const int64_t Cpad = (n_centroids + 15) & ~(int64_t)15;
  const uint32_t pad_code = (uint32_t)std::min<int64_t>((int64_t)1 << 17, Cpad - code_lo);
  HCHK(hipMalloc((void**)poff, (size_t)(n_docs + 1) * 8 + 64));
  HCHK(hipMalloc((void**)&nlines, (size_t)(n_docs + 1) * 8));
  HCHK(hipMalloc((void**)&loff, (size_t)(n_docs + 1) * 8));
  HCHK(hipMalloc((void**)&rs, (size_t)(n_docs + 1) * 8));
  HCHK(hipMalloc((void**)&re, (size_t)(n_docs + 1) * 8));
  hipLaunchKernelGGL(k_pcode_range, dim3(fp_grid_cap((n_docs + 256) / 256, 256)), dim3(256), 0, st, ucodes, uoff_dev, n_docs, (int32_t)code_lo,
                     (int32_t)code_hi, rs, re);
  hipLaunchKernelGGL(k_pcode_lines, dim3(fp_grid_cap((n_docs + 256) / 256, 256)), dim3(256), 0, st, rs, re, n_docs, nlines, PCODES_PER_PIECE * ppl);
  HCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, nlines, loff, (int)(n_docs + 1), st));
  HCHK(hipMalloc(&tmp, tb ? tb : 16));
  HCHK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, nlines, loff, (int)(n_docs + 1), st));
  HCHK(hipMemcpyAsync(n_lines, loff + n_docs, 8, hipMemcpyDeviceToHost, st));
  HCHK(hipStreamSynchronize(st));
  if (*n_lines >= 0x7FFFFFFFll) { rc = (int)hipErrorInvalidValue; goto fail; }
  {
    const int64_t n_extra = *n_lines;
    const int64_t all = n_extra + n_docs;   // with the first lines
    if (all >= 0x7FFFFFFFll || n_docs * (int64_t)nr >= 0x7FFFFFFFll) { rc = (int)hipErrorInvalidValue; goto fail; }
    *n_lines = shared_first ? n_extra : all;   // lines in *lines
    HCHK(hipMalloc(lines, (size_t)(*n_lines > 0 ? *n_lines : 1) * 16 * ppl + 256));
    const int64_t work = all * ppl + n_docs + 1;
    hipLaunchKernelGGL(k_pcode_pack, dim3(fp_grid_cap((work + 255) / 256, 256)), dim3(256), 0, st, ucodes, rs, re, n_docs, (int32_t)code_lo, loff,
                       *poff, static_cast<uint4*>(*lines), ppl, pad_code, static_cast<uint4*>(shared_first), nr, r);
  }
  HCHK(hipStreamSynchronize(st));
fail:
  if (nlines) (void)hipFree(nlines);
  if (loff) (void)hipFree(loff);
  if (rs) (void)hipFree(rs);
  if (re) (void)hipFree(re);
  if (tmp) (void)hipFree(tmp);
  return rc;
}

// ---- IVF: per-cell ascending unique local doc ids, from the per-document unique codes -------------
__global__ __launch_bounds__(256) void k_make_keys(const int32_t* __restrict__ ucodes, const int64_t* __restrict__ uoff,
                                                   int64_t n_docs, unsigned long long* __restrict__ keys) {
  const int lane = threadIdx.x & 63;
  const int64_t dstep = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t d = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; d < n_docs; d += dstep) {
    const int64_t off = uoff[d], end = uoff[d + 1];
    for (int64_t t = off + lane; t < end; t += 64)
      keys[t] = ((unsigned long long)(uint32_t)ucodes[t] << 32) | (unsigned long long)(uint32_t)d;
  }
}

__global__ void k_ivf_pids(const unsigned long long* __restrict__ uniq, int64_t U, int32_t* __restrict__ pids) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < U; i += (int64_t)gridDim.x * blockDim.x)
    pids[i] = (int32_t)(uint32_t)uniq[i];
}

__global__ void k_ivf_offsets(const unsigned long long* __restrict__ uniq, int64_t U, int64_t P, int64_t* __restrict__ ivf_off) {
  int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c > P) return;
  const unsigned long long target = (unsigned long long)c << 32;  // first key of cell c
  int64_t lo = 0, hi = U;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (uniq[mid] < target) lo = mid + 1; else hi = mid;
  }
  ivf_off[c] = lo;
}

int fps_build_ivf(const int32_t* ucodes, const int64_t* uoff_dev, int64_t n_docs, int64_t U, int64_t P, int32_t** ivf_pids,
                  int64_t* ivf_total, int64_t* ivf_off_dev, hipStream_t st) {
  int rc = 0;
  unsigned long long *ka = nullptr, *kb = nullptr;
  void* tmp = nullptr;
  size_t tb = 0;
  int end_bit = 32;
  *ivf_pids = nullptr;
  *ivf_total = 0;
  if (U <= 0 || U > 0x7FFFFFFFll) {
    if (U == 0) {
      HCHK(hipMemsetAsync(ivf_off_dev, 0, (size_t)(P + 1) * sizeof(int64_t), st));
      HCHK(hipMalloc((void**)ivf_pids, 16));
      return 0;
    }
    return -1;
  }
  while ((1ll << (end_bit - 32)) < P) ++end_bit;
  HCHK(hipMalloc((void**)&ka, (size_t)U * 8));
  HCHK(hipMalloc((void**)&kb, (size_t)U * 8));
  hipLaunchKernelGGL(k_make_keys, dim3(fp_grid_cap((n_docs + 3) / 4, 256)), dim3(256), 0, st, ucodes, uoff_dev, n_docs, ka);
  HCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, ka, kb, (int)U, 0, end_bit, st));
  HCHK(hipMalloc(&tmp, tb ? tb : 16));
  HCHK(hipcub::DeviceRadixSort::SortKeys(tmp, tb, ka, kb, (int)U, 0, end_bit, st));
  HCHK(hipMalloc((void**)ivf_pids, (size_t)U * sizeof(int32_t) + 64));
  hipLaunchKernelGGL(k_ivf_pids, dim3(fp_grid_cap((U + 255) / 256, 256)), dim3(256), 0, st, kb, U, *ivf_pids);
  hipLaunchKernelGGL(k_ivf_offsets, dim3((unsigned)((P + 1 + 255) / 256)), dim3(256), 0, st, kb, U, P, ivf_off_dev);
  HCHK(hipStreamSynchronize(st));
  *ivf_total = U;
fail:
  if (ka) (void)hipFree(ka);
  if (kb) (void)hipFree(kb);
  if (tmp) (void)hipFree(tmp);
  return rc;
}


// ---- final ranking of rerank lists beyond the LDS sort (R > 16384, i.e. n_full_scores > 65536) --------------------------
// Same keys and order as k_final_topk (score descending, id ascending; padding entries of a sharded buffer skipped), sorted by a
// segmented device radix sort (one segment per query).  The temporaries are allocated and freed inside the call: a rare path
// (the reference accepts any n_full_scores, rust/search/search.rs:605-619), kept simple rather than fast.
static __device__ __forceinline__ uint32_t srt_mono32(float f) {
  uint32_t b = __float_as_uint(f + 0.0f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__global__ void k_final_keys(const float* __restrict__ score, const int32_t* __restrict__ pid_local, const int64_t* __restrict__ pid_global,
                             const int32_t* __restrict__ cnt, int64_t stride, int B, int64_t pid_offset, unsigned long long* __restrict__ keys,
                             int64_t* __restrict__ seg) {
  const int64_t total = (int64_t)B * stride;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / stride);
    const int64_t r = i - (int64_t)b * stride;
    unsigned long long key = 0ull;
    if (r < (cnt ? (int64_t)cnt[b] : stride)) {
      const int64_t id = pid_local ? (int64_t)pid_local[i] + pid_offset : pid_global[i];
      if (id >= 0) key = ((unsigned long long)srt_mono32(score[i]) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)id);
    }
    keys[i] = key;
    if (r == 0) seg[b] = i;
    if (i == total - 1) seg[B] = total;
  }
}
__global__ void k_final_emit(const unsigned long long* __restrict__ keys, int64_t stride, int B, int64_t top_k, int64_t* __restrict__ out_pid,
                             float* __restrict__ out_score, int32_t* __restrict__ out_cnt) {
  const int b = blockIdx.x;
  const unsigned long long* v = keys + (int64_t)b * stride;
  __shared__ int s_m;
  if (threadIdx.x == 0) {   // valid keys are > 0 and sorted to the front: first zero by bisection
    int64_t lo = 0, hi = stride;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (v[mid] != 0ull) lo = mid + 1; else hi = mid; }
    s_m = (int)(lo < top_k ? lo : top_k);
    out_cnt[b] = s_m;
  }
  __syncthreads();
  const int m = s_m;
  for (int64_t i = threadIdx.x; i < top_k; i += blockDim.x) {
    if (i < m) {
      const unsigned long long key = v[i];
      const uint32_t kb = (uint32_t)(key >> 32);
      out_pid[(int64_t)b * top_k + i] = (int64_t)(0xFFFFFFFFu - (uint32_t)key);
      out_score[(int64_t)b * top_k + i] = __uint_as_float((kb & 0x80000000u) ? (kb & 0x7FFFFFFFu) : ~kb);
    } else {
      out_pid[(int64_t)b * top_k + i] = -1;
      out_score[(int64_t)b * top_k + i] = 0.f;
    }
  }
}
int fps_final_topk_big(const float* score, const int32_t* pid_local, const int64_t* pid_global, const int32_t* cnt, int64_t stride, int B,
                       int64_t top_k, int64_t pid_offset, int64_t* out_pid, float* out_score, int32_t* out_cnt, hipStream_t st) {
  int rc = 0;
  const int64_t total = (int64_t)B * stride;
  unsigned long long *ka = nullptr, *kb = nullptr;
  int64_t* seg = nullptr;
  void* tmp = nullptr;
  size_t tb = 0;
  if (total >= 0x7FFFFFFFll) return -1;
  HCHK(hipMalloc((void**)&ka, (size_t)total * 8));
  HCHK(hipMalloc((void**)&kb, (size_t)total * 8));
  HCHK(hipMalloc((void**)&seg, (size_t)(B + 1) * 8));
  hipLaunchKernelGGL(k_final_keys, dim3(fp_grid_cap((total + 255) / 256, 256)), dim3(256), 0, st, score, pid_local, pid_global, cnt, stride, B,
                     pid_offset, ka, seg);
  HCHK(hipcub::DeviceSegmentedRadixSort::SortKeysDescending(nullptr, tb, ka, kb, (int)total, B, seg, seg + 1, 0, 64, st));
  HCHK(hipMalloc(&tmp, tb ? tb : 16));
  HCHK(hipcub::DeviceSegmentedRadixSort::SortKeysDescending(tmp, tb, ka, kb, (int)total, B, seg, seg + 1, 0, 64, st));
  hipLaunchKernelGGL(k_final_emit, dim3((unsigned)B), dim3(256), 0, st, kb, stride, B, top_k, out_pid, out_score, out_cnt);
  HCHK(hipStreamSynchronize(st));
fail:
  if (ka) (void)hipFree(ka);
  if (kb) (void)hipFree(kb);
  if (seg) (void)hipFree(seg);
  if (tmp) (void)hipFree(tmp);
  return rc;
}
