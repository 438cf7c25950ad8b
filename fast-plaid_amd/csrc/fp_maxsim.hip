// S6+S7 of the PLAID search path for gfx950: residual decompression (rust/search/search.rs:53-107) fused with the exact
// MaxSim (search.rs:626-656), the per-token norms it reads (computed once at index creation), and the exact-order repair of
// near-tied final scores.
//
// Numerical contract (oracle/plaid_oracle.c): e = h(cent + w); n = h(sqrt(sum_fp32 e_k^2)) with the sum in ascending k;
// e^ = h(fl32(e / n)); sim = h(sum_fp32 e^_k q_k); score = sum_fp32 over query tokens of max over document tokens.
// Everything up to e^ is reproduced bit for bit.  The contraction runs on MFMA, whose fp32 accumulation ORDER differs from
// the CPU's ascending-k chain, so a sim can differ by one fp16 ulp when its fp32 value lies within a few fp32 ulps of a
// rounding boundary.  The kernel therefore keeps the column maxima in fp32 (h is monotone: max_t h(a_t) = h(max_t a_t)),
// rounds once per (document, column), and flags the columns whose fp32 maximum lies within eps of a boundary; k_final_mark
// picks the flagged documents whose score is near-tied with a neighbour of the final ranking, and k_maxsim_repair recomputes
// exactly those columns with the ascending-k chain (bit-identical to the reference) before the final sort.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "fp_device.h"
#include "fp_internal.h"

#define MS_WAVES 12
#define MS_THREADS (MS_WAVES * 64)

template <int D, int NBITS>
struct MsCfg {
  static constexpr int PR = D * NBITS / 8;   // packed residual bytes per token
  static constexpr int RW = PR / 4;          // 32-bit residual words per token
  static constexpr int PB = 8 / NBITS;       // dims per byte
  static constexpr int KS = D / 16;          // MFMA k-steps
  static constexpr int NE = D / 2;           // half2 registers per token
  static constexpr int EW = PB >= 2 ? PB / 2 : 1;   // 32-bit words per LUT entry
  static constexpr int COPIES = 64 / EW;     // LUT copies: every byte value owns 256 bytes of LDS, one entry per lane (group)
  static constexpr int RAL = PR % 16 == 0 ? 16 : (PR % 8 == 0 ? 8 : 4);   // alignment of a residual row
  static_assert(D % 16 == 0 && PR % 4 == 0, "fast path: dim a multiple of 16, packed rows a multiple of 4 bytes");
};
#define MS_LUT_BYTES (256 * 256)   // byte -> weights table, 64 KiB of LDS whatever nbits is

// ---- byte -> bucket weights table in LDS -----------------------------------------------------------------------------
// entry (byte v, copy c) at byte offset v * 256 + c * EW * 4: the LDS address of a lane's entry is {0, 0, v, laneoff} as
// bytes, i.e. ONE v_perm_b32 of the residual word and a per-lane constant (no shift / mask / add per byte).
template <int D, int NBITS>
__device__ __forceinline__ void ms_fill_lut(unsigned char* lds, const uint16_t* __restrict__ lut_g, int tid, int nthreads) {
  using Cf = MsCfg<D, NBITS>;
  uint32_t* l32 = reinterpret_cast<uint32_t*>(lds);
  for (int i = tid; i < 256 * Cf::COPIES; i += nthreads) {
    const int entry = i / Cf::COPIES, copy = i % Cf::COPIES;
#pragma unroll
    for (int w = 0; w < Cf::EW; ++w) {
      uint32_t word;
      if constexpr (NBITS == 8) word = (uint32_t)lut_g[entry];
      else word = reinterpret_cast<const uint32_t*>(lut_g)[entry * Cf::EW + w];
      l32[(entry * 256 + copy * Cf::EW * 4) / 4 + w] = word;
    }
  }
}

// loads of one token: packed residual row -> rw, centroid row -> e
template <int D, int NBITS>
__device__ __forceinline__ void ms_load_token(const uint8_t* __restrict__ resid, const uint16_t* __restrict__ cent, long long row, int32_t code,
                                              uint32_t (&rw)[MsCfg<D, NBITS>::RW], uint32_t (&e)[MsCfg<D, NBITS>::NE]) {
  using Cf = MsCfg<D, NBITS>;
  const uint8_t* rp = resid + row * (long long)Cf::PR;
  if constexpr (Cf::RAL == 16) {
#pragma unroll
    for (int i = 0; i < Cf::RW / 4; ++i) {
      const uint4 v = *reinterpret_cast<const uint4*>(rp + 16 * i);
      rw[4 * i] = v.x; rw[4 * i + 1] = v.y; rw[4 * i + 2] = v.z; rw[4 * i + 3] = v.w;
    }
  } else if constexpr (Cf::RAL == 8) {
#pragma unroll
    for (int i = 0; i < Cf::RW / 2; ++i) {
      const uint2 v = *reinterpret_cast<const uint2*>(rp + 8 * i);
      rw[2 * i] = v.x; rw[2 * i + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < Cf::RW; ++i) rw[i] = *reinterpret_cast<const uint32_t*>(rp + 4 * i);
  }
  const uint16_t* cp = cent + (long long)code * D;
#pragma unroll
  for (int i = 0; i < Cf::NE / 4; ++i) {
    const uint4 v = *reinterpret_cast<const uint4*>(cp + 8 * i);
    e[4 * i] = v.x; e[4 * i + 1] = v.y; e[4 * i + 2] = v.z; e[4 * i + 3] = v.w;
  }
}

// The table sits at LDS address 0 (the kernels that use it declare no static __shared__ and put it first in their dynamic LDS;
// ms_lds_base_is_zero() traps otherwise), so the v_perm result IS the ds_read address: no base add per byte.
typedef uint32_t ms_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t ms_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const uint32_t ms_lds_u32;
typedef __attribute__((address_space(3))) const ms_u32x2 ms_lds_u64;
typedef __attribute__((address_space(3))) const ms_u32x4 ms_lds_u128;
__device__ __forceinline__ void ms_lds_base_is_zero(const unsigned char* smem) {
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)smem != 0u) __builtin_trap();
}

// e = h(cent + w): packed fp16 adds (== fp32 add + one rounding, fp_selftest_arith out[1])
template <int D, int NBITS>
__device__ __forceinline__ void ms_decode(uint32_t laneoff, const uint32_t (&rw)[MsCfg<D, NBITS>::RW], uint32_t (&e)[MsCfg<D, NBITS>::NE]) {
  using Cf = MsCfg<D, NBITS>;
#pragma unroll
  for (int w = 0; w < Cf::RW; ++w) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int bi = w * 4 + kk;   // byte index -> dims [bi*PB, bi*PB + PB)
      const uint32_t addr = __builtin_amdgcn_perm(rw[w], laneoff, 0x0C0C0400u | ((uint32_t)kk << 8));
      if constexpr (NBITS == 4) {
        e[bi] = h2_as_u32(u32_as_h2(e[bi]) + u32_as_h2(*(ms_lds_u32*)(uintptr_t)addr));
      } else if constexpr (NBITS == 2) {
        const ms_u32x2 wv = *(ms_lds_u64*)(uintptr_t)addr;
        e[2 * bi] = h2_as_u32(u32_as_h2(e[2 * bi]) + u32_as_h2(wv.x));
        e[2 * bi + 1] = h2_as_u32(u32_as_h2(e[2 * bi + 1]) + u32_as_h2(wv.y));
      } else if constexpr (NBITS == 1) {
        const ms_u32x4 wv = *(ms_lds_u128*)(uintptr_t)addr;
        e[4 * bi] = h2_as_u32(u32_as_h2(e[4 * bi]) + u32_as_h2(wv.x));
        e[4 * bi + 1] = h2_as_u32(u32_as_h2(e[4 * bi + 1]) + u32_as_h2(wv.y));
        e[4 * bi + 2] = h2_as_u32(u32_as_h2(e[4 * bi + 2]) + u32_as_h2(wv.z));
        e[4 * bi + 3] = h2_as_u32(u32_as_h2(e[4 * bi + 3]) + u32_as_h2(wv.w));
      } else {   // NBITS == 8: one dim per byte, two bytes make one packed register
        if ((kk & 1) == 0) {
          const uint32_t addr1 = __builtin_amdgcn_perm(rw[w], laneoff, 0x0C0C0400u | ((uint32_t)(kk + 1) << 8));
          const uint32_t lo = *(ms_lds_u32*)(uintptr_t)addr;
          const uint32_t hi = *(ms_lds_u32*)(uintptr_t)addr1;
          const uint32_t wv = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
          e[bi / 2] = h2_as_u32(u32_as_h2(e[bi / 2]) + u32_as_h2(wv));
        }
      }
    }
  }
}

// ---- per-token norms, computed once at index creation ------------------------------------------------------------------
// n = h(sqrt(sum_k e_k^2)), fp32, ascending k: exactly the CPU reference's order (a blocked or dot2 order flips n by an ulp on
// 0.008 % of tokens).  Stored as fp16 bits, 2 B per token; the search kernels read it instead of re-deriving the 128-step chain
// for every (query, document) pair.  Any dim / nbits (runtime loops; this runs once per index).
__global__ __launch_bounds__(256) void k_token_norms(const uint16_t* __restrict__ cent, const uint16_t* __restrict__ lut,
                                                     const int32_t* __restrict__ codes, const uint8_t* __restrict__ resid, int D, int nbits,
                                                     int64_t T, uint16_t* __restrict__ norms) {
  const int pb = 8 / nbits, pr = D * nbits / 8;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (int64_t)gridDim.x * blockDim.x) {
    const uint16_t* cp = cent + (int64_t)codes[t] * D;
    const uint8_t* rp = resid + t * pr;
    float ss = 0.f;
    for (int by = 0; by < pr; ++by) {
      const int byte = rp[by];
      for (int j = 0; j < pb; ++j) {
        const half_t w = __builtin_bit_cast(half_t, lut[byte * pb + j]);
        const half_t c = __builtin_bit_cast(half_t, cp[by * pb + j]);
        const float e = (float)(half_t)((float)w + (float)c);
        ss = __builtin_fmaf(e, e, ss);
      }
    }
    norms[t] = __builtin_bit_cast(uint16_t, (half_t)__builtin_sqrtf(ss));   // clamp_min(1e-12) is a no-op in fp16
  }
}

void fpk_token_norms(const FpIndexDev& ix, uint16_t* norms, hipStream_t st) {
  if (ix.T <= 0) return;
  hipLaunchKernelGGL(k_token_norms, dim3(fp_grid_cap((ix.T + 255) / 256, 256)), dim3(256), 0, st, ix.centroids, ix.lut, ix.codes, ix.residuals,
                     ix.dim, ix.nbits, ix.T, norms);
}

// inclusive prefix of the per-query rerank counts: pref[0] = 0, pref[b+1] = sum_{i<=b} cnt[i]
__global__ __launch_bounds__(256) void k_cnt_prefix(const int32_t* __restrict__ cnt, int B, int64_t* __restrict__ pref) {
  __shared__ long long s[256];
  long long base = 0;
  if (threadIdx.x == 0) pref[0] = 0;
  for (int start = 0; start < B; start += 256) {
    const int i = start + threadIdx.x;
    const long long x = (i < B) ? (long long)cnt[i] : 0;
    s[threadIdx.x] = x;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const long long t = ((int)threadIdx.x >= off) ? s[threadIdx.x - off] : 0;
      __syncthreads();
      s[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < B) pref[i + 1] = base + s[threadIdx.x];
    const long long tot = s[255];
    __syncthreads();
    base += tot;
  }
}

struct MsArgs {
  const uint16_t* cent;
  const uint16_t* lut_g;
  const int32_t* codes;
  const uint16_t* norms;
  const uint8_t* resid;
  const int64_t* doc_off;
  const uint16_t* qpad;
  const int32_t* sel_pid;
  const int64_t* pref;     // [B+1] prefix of the rerank counts
  float* exact;            // [B][Rcap]
  uint16_t* cm16;          // [B][Rcap][Qp] per-column maxima as fp16 bits (nullable)
  float* unc;              // [B][Rcap] sum of the fp16 ulps of the flagged columns (0 = certainly the reference's score; nullable)
  float* uncm;             // [B][Rcap] the part of unc by which the reference's score may be LOWER (flagged columns whose fp32 maximum
                           // sits just above a rounding boundary); the score may be higher by unc - uncm.  Nullable with unc.
  uint32_t* flags;         // [B][Rcap][Qp/32] flagged columns (nullable)
  int64_t Rcap;
  int B, Q, Qp, ch_begin, accumulate;
  float eps_rel;           // a column is flagged when its fp32 maximum lies within eps_rel * |q_col| of an fp16 rounding boundary
};

// Layout (v5).  ONE TOKEN PER LANE: a wave carries two independent 32-token streams (lanes 0-31 and 32-63), each walking its
// own sequence of documents chunk by chunk.  A lane decompresses all D dims of its token, then v_permlane32_swap turns the
// per-token registers into the two A operands of v_mfma_f32_32x32x16_f16:
//   regs G0 = dims [16m, 16m+8), G1 = dims [16m+8, 16m+16) of the lane's own token;
//   swap(vdst=G0, src=G1): lanes 32-63 of G0 <-> lanes 0-31 of G1
//   => G0 = A operand for the LOWER stream's 32 tokens (rows), G1 = A operand for the UPPER one.
// The kernel is VALU-issue bound (rocprof time == issue slots x 4 cycles): what v5 removes against v3 is the 128-step norm
// chain (stored norms), two of the three VALU ops per residual byte (v_perm address), the per-value fp16 conversions of the
// epilogue (fp32 v_max3), and the grid tail (every workgroup takes an equal share of the batch's (query, document) pairs).
template <int D, int NBITS, int NCH>
__global__ __launch_bounds__(MS_THREADS) void k_maxsim5(const MsArgs a) {
  using Cf = MsCfg<D, NBITS>;
  constexpr int KS = Cf::KS, NE = Cf::NE, RW = Cf::RW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* lut = smem;                                                        // 64 KiB
  uint4* qs = reinterpret_cast<uint4*>(smem + MS_LUT_BYTES);                       // B fragments, 1 KiB per (chunk, k-step)
  float* qn = reinterpret_cast<float*>(smem + MS_LUT_BYTES + NCH * KS * 64 * 16);  // [NCH*32] eps of the column
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, hi = lane >> 5;
  ms_lds_base_is_zero(smem);
  ms_fill_lut<D, NBITS>(lut, a.lut_g, tid, MS_THREADS);
  const uint32_t laneoff = (uint32_t)((lane & (Cf::COPIES - 1)) * Cf::EW * 4);
  const half_t negm = (half_t)NEG_MASK_F;
  int nq = a.Q - a.ch_begin * 32;   // valid query columns handled by this launch
  nq = nq < 0 ? 0 : (nq > NCH * 32 ? NCH * 32 : nq);
  const int nflag = a.Qp / 32;

  // this workgroup's share of the flattened (query, rerank slot) space
  const long long tot = a.pref[a.B];
  const long long lo = tot * (long long)blockIdx.x / (long long)gridDim.x;
  const long long hi_end = tot * (long long)(blockIdx.x + 1) / (long long)gridDim.x;
  int b = 0;
  {  // last b with pref[b] <= lo
    int l = 0, h = a.B;
    while (h - l > 1) { const int m = (l + h) >> 1; if (a.pref[m] <= lo) l = m; else h = m; }
    b = l;
  }
  for (; b < a.B && a.pref[b] < hi_end; ++b) {
    const long long pb0 = a.pref[b], pb1 = a.pref[b + 1];
    const int ra = (int)((lo > pb0 ? lo : pb0) - pb0);
    const int rb = (int)((hi_end < pb1 ? hi_end : pb1) - pb0);
    if (rb <= ra) continue;
    __syncthreads();   // the previous query's fragments are no longer read (and the LUT is complete)
    for (int i = tid; i < NCH * KS * 64; i += MS_THREADS) {
      const int ln = i & 63, m = (i >> 6) % KS, c = (i >> 6) / KS;
      const int ch = a.ch_begin + c;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ch * 32 < a.Qp)  // B[k = 8*(ln>>5) + j][n = ln&31] = Q[q = ch*32 + (ln&31)][16m + 8*(ln>>5) + j]
        v = *reinterpret_cast<const uint4*>(a.qpad + ((int64_t)b * a.Qp + ch * 32 + (ln & 31)) * D + 16 * m + 8 * (ln >> 5));
      qs[i] = v;
    }
    if (tid < NCH * 32) {   // eps of the column = eps_rel * |q|_2  (|e^| <= ~1: the two fp32 sums differ by at most ~2^-24 sum|e^_k q_k| D)
      const int ch = a.ch_begin + tid / 32;
      float ss = 0.f;
      if (ch * 32 < a.Qp) {
        const uint16_t* qp = a.qpad + ((int64_t)b * a.Qp + ch * 32 + (tid & 31)) * D;
        for (int k = 0; k < D; ++k) { const float x = (float)__builtin_bit_cast(half_t, qp[k]); ss = __builtin_fmaf(x, x, ss); }
      }
      qn[tid] = a.eps_rel * __builtin_sqrtf(ss);
    }
    __syncthreads();
    const int per = (rb - ra + MS_WAVES - 1) / MS_WAVES;
    const int r0 = ra + wave * per;
    if (r0 >= rb) continue;
    const int rend = (r0 + per < rb) ? (r0 + per) : rb;
    const int32_t* selp = a.sel_pid + (int64_t)b * a.Rcap;
    float* outp = a.exact + (int64_t)b * a.Rcap;

    // ---- per-stream state (identical in the 32 lanes of a stream) ----
    int r = r0 + hi;          // document slot of this stream: r0+hi, r0+hi+2, ...
    int t0 = 0, len = 0;
    long long off = 0;
    bool valid = r < rend;
    long long n_off = 0;      // prefetched metadata of the stream's next document
    int n_len = 0;
    auto meta = [&](int rr, long long& o, int& l) {
      const int32_t pid = selp[rr];
      o = a.doc_off[pid];
      l = (int)(a.doc_off[pid + 1] - o);
    };
    if (valid) meta(r, off, len);
    if (r + 2 < rend) meta(r + 2, n_off, n_len);
    auto next_doc = [&]() {  // move the stream to its next document (may be empty or absent)
      r += 2;
      valid = r < rend;
      off = n_off;
      len = n_len;
      t0 = 0;
      if (r + 2 < rend) meta(r + 2, n_off, n_len);
    };
    auto emit_empty = [&]() {   // every column keeps the masked value; nothing to flag
      if (l31 == 0) {
        const float v = (float)nq * NEG_MASK_F;
        outp[r] = a.accumulate ? (outp[r] + v) : v;
        if (a.unc && !a.accumulate) { a.unc[(int64_t)b * a.Rcap + r] = 0.f; a.uncm[(int64_t)b * a.Rcap + r] = 0.f; }
      }
      if (a.cm16) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
          if ((a.ch_begin + c) * 32 < a.Qp) a.cm16[((int64_t)b * a.Rcap + r) * a.Qp + (a.ch_begin + c) * 32 + l31] = __builtin_bit_cast(uint16_t, negm);
      }
      if (a.flags && l31 == 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
          if (a.ch_begin + c < nflag) a.flags[((int64_t)b * a.Rcap + r) * nflag + a.ch_begin + c] = 0u;
      }
    };
    while (valid && len == 0) { emit_empty(); next_doc(); }

    auto tok_row = [&]() -> long long {
      int tok = t0 + l31;
      tok = tok < len ? tok : len - 1;  // clamp: loads stay in bounds, rows masked at the max
      return off + tok;
    };
    int32_t code = 0;
    uint16_t nrm = 0;
    if (valid) { const long long rr = tok_row(); code = a.codes[rr]; nrm = a.norms[rr]; }

    float mx[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) mx[c] = NEG_MASK_F;

    // (issuing the next chunk's loads before the epilogue -- software pipelining -- measured slower: 251 -> 266 us, 8 dwords of
    // scratch; the kernel is VALU-issue bound, not latency bound)
    while (__any(valid)) {
      uint32_t e[NE];
      uint32_t rw[RW];
      // ---- loads of this chunk (exhausted streams read row 0 of the arrays: harmless) ----
#if defined(FP_MS_ABL) && (FP_MS_ABL == 1 || FP_MS_ABL == 5)
      ms_load_token<D, NBITS>(a.resid, a.cent, valid ? tok_row() : 0, 0, rw, e);   // timing-only ablation: no scattered centroid gather
#else
      ms_load_token<D, NBITS>(a.resid, a.cent, valid ? tok_row() : 0, valid ? code : 0, rw, e);
#endif
      const float nf = (float)__builtin_bit_cast(half_t, nrm);
      // ---- this chunk's bookkeeping, then step the stream and prefetch the next code / norm ----
      const int c_t0 = t0, c_len = len, c_r = r;
      const bool c_valid = valid;
      const bool c_last = valid && (t0 + 32 >= len);
      if (valid) {
        t0 += 32;
        if (t0 >= len) {
          next_doc();
          while (valid && len == 0) { emit_empty(); next_doc(); }
        }
        if (valid) { const long long rr = tok_row(); code = a.codes[rr]; nrm = a.norms[rr]; }
      }
      // ---- e = h(cent + w);  e^ = h(fl32(e / n)) through the compensated reciprocal product (fp_selftest_arith out[0]) ----
#if !defined(FP_MS_ABL) || FP_MS_ABL != 3
      ms_decode<D, NBITS>(laneoff, rw, e);
#else
#pragma unroll
      for (int i = 0; i < RW; ++i) e[i] ^= rw[i];
#endif
      float r_hi, r_lo;
      recip2(nf, r_hi, r_lo);
#if !defined(FP_MS_ABL) || FP_MS_ABL == 1 || FP_MS_ABL == 3
#pragma unroll
      for (int i = 0; i < NE; i += 2) norm_pair2(e[i], e[i + 1], r_hi, r_lo);
#elif FP_MS_ABL == 4 || FP_MS_ABL == 5   // timing-only: one v_fma_mixlo/hi per element
#pragma unroll
      for (int i = 0; i < NE; ++i)
        asm volatile("v_fma_mixlo_f16 %0, %0, %1, 0 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %0, %1, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(e[i]) : "v"(r_hi + r_lo));
#else
      e[0] ^= __float_as_uint(r_hi + r_lo);
#endif
      // ---- MFMA: acc0 rows = lower stream's tokens, acc1 rows = upper stream's tokens ----
      f16v acc0[NCH], acc1[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc0[c][i] = 0.f; acc1[c][i] = 0.f; }
#pragma unroll
      for (int m = 0; m < KS; ++m) {
        uint32_t g0[4], g1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          auto sw = __builtin_amdgcn_permlane32_swap(e[8 * m + j], e[8 * m + 4 + j], false, false);
          g0[j] = sw[0];
          g1[j] = sw[1];
        }
        const h8 a0 = __builtin_bit_cast(h8, make_uint4(g0[0], g0[1], g0[2], g0[3]));
        const h8 a1 = __builtin_bit_cast(h8, make_uint4(g1[0], g1[1], g1[2], g1[3]));
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const h8 bq = __builtin_bit_cast(h8, qs[(c * KS + m) * 64 + lane]);
          acc0[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bq, acc0[c], 0, 0, 0);
          acc1[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bq, acc1[c], 0, 0, 0);
        }
      }
      // ---- epilogue.  acc0 belongs to the lower stream (state in lane 0), acc1 to the upper (lane 32);
      // D[row = token][col = q = lane&31], this lane's rows (i&3) + 8*(i>>2) + 4*hi.  The maximum over tokens is taken on
      // the fp32 accumulators (rounding is monotone, so it commutes with the maximum) ----
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int s_t0 = __builtin_amdgcn_readlane(c_t0, 32 * s), s_len = __builtin_amdgcn_readlane(c_len, 32 * s);
        const int s_r = __builtin_amdgcn_readlane(c_r, 32 * s);
        const bool s_valid = __builtin_amdgcn_readlane((int)c_valid, 32 * s) != 0;
        const bool s_last = __builtin_amdgcn_readlane((int)c_last, 32 * s) != 0;
        if (!s_valid) continue;
        const bool partial = (s_t0 + 32 > s_len);
        float total = 0.f, ubud = 0.f, ubm = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = (s == 0) ? acc0[c][i] : acc1[c][i];
          if (partial) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int row = (i & 3) + 8 * (i >> 2) + 4 * hi;
              if (s_t0 + row >= s_len) v[i] = NEG_MASK_F;
            }
          }
          float m = __builtin_fmaxf(v[0], v[1]);
#pragma unroll
          for (int i = 2; i < 16; i += 2) m = __builtin_fmaxf(__builtin_fmaxf(m, v[i]), v[i + 1]);   // v_max3_f32
          // the other lane half holds the other 16 tokens of the same column
          {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            m = __builtin_fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
          }
          // running maximum lives in the lanes of stream s
          if (hi == s) mx[c] = __builtin_fmaxf(mx[c], m);
          if (s_last) {
            const int q = (a.ch_begin + c) * 32 + l31;
            const bool mine = (hi == s) && (q < a.Q);
            // the reference's value: h(max_t fp32 sum); masked / empty columns keep -10000 (exactly representable)
            const float am = mx[c];
            const half_t hm = (half_t)am;
            float sv = mine ? (float)hm : 0.f;
            // distance of the fp32 maximum to the nearest fp16 rounding boundary
            uint32_t ef = (__float_as_uint(am) >> 23) & 0xFFu;
            ef = ef < 113u ? 113u : ef;                                   // fp16 subnormal range: fixed spacing 2^-24
            const float halfulp = __uint_as_float((ef - 11u) << 23);
            const float dist = halfulp - __builtin_fabsf(am - (float)hm);
            const bool flag = mine && !(dist > qn[c * 32 + l31]);         // NaN -> flagged (the repair reproduces it)
            float fu = flag ? 2.0f * halfulp : 0.f;
            // the other candidate lies on ONE side: a maximum that was rounded up (am < hm) can only have been one ulp lower
            float fm = (flag && !(am > (float)hm)) ? 2.0f * halfulp : 0.f;
            if (flag && !(am < (float)hm) && !(am > (float)hm)) fu += 2.0f * halfulp;   // exactly on the value (or NaN): both sides
            const unsigned long long bal = __ballot(flag);
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) {
              sv += __shfl_xor(sv, sft, 64);
              fu += __shfl_xor(fu, sft, 64);
              fm += __shfl_xor(fm, sft, 64);
            }
            total += sv;
            ubud += fu;
            ubm += fm;
            if (a.cm16 && hi == s && (a.ch_begin + c) * 32 < a.Qp)
              a.cm16[((int64_t)b * a.Rcap + s_r) * a.Qp + (a.ch_begin + c) * 32 + l31] = __builtin_bit_cast(uint16_t, hm);
            if (a.flags && lane == 0 && a.ch_begin + c < nflag)
              a.flags[((int64_t)b * a.Rcap + s_r) * nflag + a.ch_begin + c] = (uint32_t)(s == 0 ? bal : (bal >> 32));
            if (hi == s) mx[c] = NEG_MASK_F;
          }
        }
        if (s_last && lane == 0) {
          outp[s_r] = a.accumulate ? (outp[s_r] + total) : total;
          if (a.unc) {
            float* up = a.unc + (int64_t)b * a.Rcap + s_r;
            *up = a.accumulate ? (*up + ubud) : ubud;
            float* um = a.uncm + (int64_t)b * a.Rcap + s_r;
            *um = a.accumulate ? (*um + ubm) : ubm;
          }
        }
      }
    }
  }
}

// ---- any dim / nbits: exact ascending-k chains (the arithmetic of k_token_scores; bit-identical to the reference, so nothing
// is ever flagged).  One 64-thread workgroup per (query, rerank slot); lane j owns query columns j, j+64, ...  Slower than the
// MFMA kernel by an order of magnitude; used for the shapes the fast path is not instantiated for.
__global__ __launch_bounds__(64) void k_maxsim_generic(const uint16_t* __restrict__ cent, const uint16_t* __restrict__ lut,
                                                       const int32_t* __restrict__ codes, const uint16_t* __restrict__ norms,
                                                       const uint8_t* __restrict__ resid, const int64_t* __restrict__ doc_off, int D, int nbits,
                                                       const uint16_t* __restrict__ qpad, int Q, int Qp, const int32_t* __restrict__ sel_pid,
                                                       const int32_t* __restrict__ sel_cnt, int64_t Rcap, float* __restrict__ exact,
                                                       float* __restrict__ unc, int qcap /*query columns per lane held in registers*/) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* e = reinterpret_cast<float*>(smem);   // [D]
  const int b = blockIdx.y, r = blockIdx.x, lane = threadIdx.x;
  if (r >= sel_cnt[b]) return;
  const int32_t pid = sel_pid[(int64_t)b * Rcap + r];
  const int64_t t0 = doc_off[pid];
  const int len = (int)(doc_off[pid + 1] - t0);
  const int pb = 8 / nbits, pr = D * nbits / 8;
  float total = 0.f;
  for (int q0 = 0; q0 < Q; q0 += 64 * qcap) {   // passes over the document when the query has more than 64*qcap columns
    float mx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) mx[i] = NEG_MASK_F;
    for (int s = 0; s < len; ++s) {
      const int64_t t = t0 + s;
      const int32_t code = codes[t];
      const float nf = (float)__builtin_bit_cast(half_t, norms[t]);
      __syncthreads();
      for (int d = lane; d < D; d += 64) {
        const int byte = resid[t * pr + d / pb];
        const half_t w = __builtin_bit_cast(half_t, lut[byte * pb + d % pb]);
        const half_t c = __builtin_bit_cast(half_t, cent[(int64_t)code * D + d]);
        const float ev = (float)(half_t)((float)w + (float)c);
        e[d] = (float)(half_t)(ev / nf);
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = q0 + i * 64 + lane;
        if (i < qcap && q < Q) {
          const uint16_t* qq = qpad + ((int64_t)b * Qp + q) * D;
          float acc = 0.f;
          for (int d = 0; d < D; ++d) acc += e[d] * (float)__builtin_bit_cast(half_t, qq[d]);   // products of fp16 pairs are exact in fp32
          const float sim = (float)(half_t)acc;
          mx[i] = sim > mx[i] ? sim : mx[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = q0 + i * 64 + lane;
      float sv = (i < qcap && q < Q) ? mx[i] : 0.f;
#pragma unroll
      for (int sft = 32; sft > 0; sft >>= 1) sv += __shfl_xor(sv, sft, 64);
      total += sv;
    }
  }
  if (lane == 0) {
    exact[(int64_t)b * Rcap + r] = total;
    if (unc) unc[(int64_t)b * Rcap + r] = 0.f;
  }
}

// ---- exact-order repair --------------------------------------------------------------------------------------------------
// k_final_mark: sort the exact scores of a query (score desc, slot asc == doc id asc: the rerank list is in ascending id
// order).  A flagged document's true (reference) score lies in [s - u, s + u], u = its uncertainty budget; the order of two
// documents is only in doubt when their intervals overlap.  With the list sorted, document i overlaps some higher-ranked one
// iff s_i + u_i >= min_{j<i} (s_j - u_j) and some lower-ranked one iff s_i - u_i <= max_{j>i} (s_j + u_j): one prefix-min and one
// suffix-max scan.  Flagged documents in such a conflict (with at least one of the two inside the emitted top_k) are marked;
// every unmarked document's interval is disjoint from all others, so after the marked ones are re-scored exactly the order is
// the reference's.  marks[b][0..nmark[b]) = rerank slots.
__global__ __launch_bounds__(1024) void k_final_mark(const float* __restrict__ score, const float* __restrict__ unc,
                                                     const float* __restrict__ uncm /*nullable: symmetric intervals*/, const int32_t* __restrict__ cnt,
                                                     int64_t stride, int npow2, int64_t top_k, int32_t* __restrict__ marks,
                                                     int32_t* __restrict__ nmark) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* v = reinterpret_cast<unsigned long long*>(smem);          // [npow2] keys
  float* lo = reinterpret_cast<float*>(smem + (size_t)npow2 * 8);              // [npow2] s - u, then its exclusive prefix-min
  float* hi = lo + npow2;                                                        // [npow2] s + u, then its exclusive suffix-max
  __shared__ int s_any, s_n;
  const int b = blockIdx.x;
  const int n = cnt[b];
  if (threadIdx.x == 0) { s_any = 0; s_n = 0; }
  __syncthreads();
  int any = 0;
  for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
    unsigned long long key = 0ull;
    if (i < n) {
      key = ((unsigned long long)mono32(score[(int64_t)b * stride + i]) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
      any |= unc[(int64_t)b * stride + i] > 0.f ? 1 : 0;
    }
    v[i] = key;
  }
  if (any) s_any = 1;
  __syncthreads();
  if (s_any) {   // (uniform) nothing flagged -> nothing to mark, no sort needed
    if (npow2 == 1024 && blockDim.x == 1024) {   // one key per thread: shuffle-based network
      const unsigned long long sorted = fp_sort1024_desc(v[threadIdx.x], v);
      v[threadIdx.x] = sorted;
      __syncthreads();
    } else {
      for (int k = 2; k <= npow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
            const int ixj = i ^ j;
            if (ixj > i) {
              const unsigned long long x = v[i], y = v[ixj];
              const bool desc = ((i & k) == 0);
              if ((x < y) == desc) { v[i] = y; v[ixj] = x; }
            }
          }
          __syncthreads();
        }
      }
    }
    const float inf = __builtin_inff();
    for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
      float l = inf, h = -inf;
      if (i < n) {
        const unsigned long long key = v[i];
        const float si = unmono32((uint32_t)(key >> 32));
        const int64_t o = (int64_t)b * stride + (int)(0xFFFFFFFFu - (uint32_t)key);
        const float u = unc[o];
        const float um = uncm ? uncm[o] : u;
        l = si - um;
        h = si + (uncm ? u - um : u);
      }
      lo[i] = l;
      hi[i] = h;
    }
    __syncthreads();
    // inclusive scans (Hillis-Steele): lo <- prefix-min, hi <- suffix-max; the exclusive values are read from the neighbours below
    for (int off = 1; off < npow2; off <<= 1) {
      float nl[16], nh[16];   // npow2 / blockDim.x <= 16 elements per thread (npow2 <= 8192 here)
      int c = 0;
      for (int i = threadIdx.x; i < npow2; i += blockDim.x, ++c) {
        nl[c] = (i >= off) ? __builtin_fminf(lo[i], lo[i - off]) : lo[i];
        nh[c] = (i + off < npow2) ? __builtin_fmaxf(hi[i], hi[i + off]) : hi[i];
      }
      __syncthreads();
      c = 0;
      for (int i = threadIdx.x; i < npow2; i += blockDim.x, ++c) { lo[i] = nl[c]; hi[i] = nh[c]; }
      __syncthreads();
    }
    const int kk = (int)(top_k < n ? top_k : n);   // emitted positions [0, kk)
    const float lmin_top = kk > 0 ? lo[kk - 1] : inf;   // min over the emitted documents of s - u
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const unsigned long long key = v[i];
      const int slot = (int)(0xFFFFFFFFu - (uint32_t)key);
      const float u = unc[(int64_t)b * stride + slot];
      if (!(u > 0.f)) continue;
      const float um = uncm ? uncm[(int64_t)b * stride + slot] : u;
      const float up = uncm ? u - um : u;
      const float si = unmono32((uint32_t)(key >> 32));
      bool conflict;
      if (i < kk) {
        const float lmin = i > 0 ? lo[i - 1] : inf;               // higher-ranked documents
        const float hmax = i + 1 < npow2 ? hi[i + 1] : -inf;      // lower-ranked documents (also those below the cut)
        conflict = !(si + up < lmin) || !(si - um > hmax);
      } else {
        conflict = !(si + up < lmin_top);                         // outside the emitted range: only a jump into it matters
      }
      if (conflict) marks[(int64_t)b * stride + atomicAdd(&s_n, 1)] = slot;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) nmark[b] = s_n;
}

// k_maxsim_repair: ONE WAVE (= one 64-thread workgroup) per marked (query, slot) -- or, with marks == nullptr, per slot whose
// budget is > 0.  Tokens are decompressed to the same bits as in k_maxsim5 (once per 64-token step; the byte -> weights table
// is a plain 1 KiB LDS copy here: filling the 64 KiB conflict-free table per workgroup cost more than the few documents a
// workgroup repairs); every flagged column is re-evaluated with the ascending-k fp32 chain, the stored column maxima are
// patched and the score re-summed (same reduction tree as k_maxsim5).
template <int D, int NBITS>
__global__ __launch_bounds__(64) void k_maxsim_repair(const uint16_t* __restrict__ cent, const uint16_t* __restrict__ lut_g,
                                                      const int32_t* __restrict__ codes, const uint16_t* __restrict__ norms,
                                                      const uint8_t* __restrict__ resid, const int64_t* __restrict__ doc_off,
                                                      const uint16_t* __restrict__ qpad, int Q, int Qp, const int32_t* __restrict__ sel_pid,
                                                      const int32_t* __restrict__ sel_cnt, int64_t Rcap, const int32_t* __restrict__ marks,
                                                      const int32_t* __restrict__ nmark, float* __restrict__ exact, float* __restrict__ unc,
                                                      uint16_t* __restrict__ cm16, uint32_t* __restrict__ flags) {
  using Cf = MsCfg<D, NBITS>;
  constexpr int NE = Cf::NE, RW = Cf::RW, PB = Cf::PB;
  __shared__ uint16_t slut[256 * PB];
  const int b = blockIdx.y;
  const int nwork = marks ? nmark[b] : sel_cnt[b];
  if ((int)blockIdx.x >= nwork) return;
  const int lane = threadIdx.x;
  for (int i = lane; i < 256 * PB; i += 64) slut[i] = lut_g[i];
  __syncthreads();
  const int nflag = Qp / 32;
  for (int wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
    const int r = marks ? marks[(int64_t)b * Rcap + wi] : wi;
    const int64_t slot = (int64_t)b * Rcap + r;
    if (!(unc[slot] > 0.f)) continue;
    const int32_t pid = sel_pid[slot];
    const long long off = doc_off[pid];
    const int len = (int)(doc_off[pid + 1] - off);
    // chunks in launch order (two per MaxSim launch); each chunk's 32 column maxima are patched in registers (lane = column) and
    // summed through the 64-lane xor tree of k_maxsim5
    float total = 0.f, part = 0.f;
    for (int ch = 0; ch < nflag; ++ch) {
      const int q = ch * 32 + (lane & 31);
      float cv = (lane < 32 && q < Q) ? (float)__builtin_bit_cast(half_t, cm16[slot * Qp + q]) : 0.f;
      const uint32_t fm = flags[slot * nflag + ch];
      if (fm) {
        float colmax = NEG_MASK_F;   // lane c: running exact maximum of column c
        for (int t0 = 0; t0 < len; t0 += 64) {
          int tok = t0 + lane;
          const bool live = tok < len;
          tok = live ? tok : len - 1;
          uint32_t e[NE];
          uint32_t rw[RW];
          ms_load_token<D, NBITS>(resid, cent, off + tok, codes[off + tok], rw, e);
          const float nf = (float)__builtin_bit_cast(half_t, norms[off + tok]);
          // e = h(cent + w): same packed fp16 adds as ms_decode, weights from the small table
#pragma unroll
          for (int w = 0; w < RW; ++w) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              const int bi = w * 4 + kk;
              const uint32_t byte = (rw[w] >> (8 * kk)) & 0xFFu;
              if constexpr (PB >= 2) {
#pragma unroll
                for (int j = 0; j < PB / 2; ++j) {
                  const uint32_t wv = *reinterpret_cast<const uint32_t*>(&slut[byte * PB + 2 * j]);
                  e[bi * (PB / 2) + j] = h2_as_u32(u32_as_h2(e[bi * (PB / 2) + j]) + u32_as_h2(wv));
                }
              } else if ((kk & 1) == 0) {   // nbits 8: two bytes make one packed register
                const uint32_t byte1 = (rw[w] >> (8 * (kk + 1))) & 0xFFu;
                const uint32_t wv = (uint32_t)slut[byte] | ((uint32_t)slut[byte1] << 16);
                e[bi / 2] = h2_as_u32(u32_as_h2(e[bi / 2]) + u32_as_h2(wv));
              }
            }
          }
          float r_hi, r_lo;
          recip2(nf, r_hi, r_lo);
#pragma unroll
          for (int i = 0; i < NE; i += 2) norm_pair2(e[i], e[i + 1], r_hi, r_lo);
          uint32_t f2 = fm;
          while (f2) {
            const int col = __builtin_ctz(f2);
            f2 &= f2 - 1;
            const uint32_t* qq = reinterpret_cast<const uint32_t*>(qpad + ((int64_t)b * Qp + ch * 32 + col) * D);   // uniform address
            // the reference's chain  acc = fma(e_k, q_k, acc), k ascending, one fp32 rounding per step.  Written out as the
            // instruction sequence: left to the compiler, fmaf((float)e.x, (float)q.x, fmaf(...)) on packed halves is open to
            // its dot-product combines (v_dot2c_f32_f16 rounds differently: tools/probe/chain_probe.hip counts 79 % of random
            // 128-term chains off in the last fp32 bit), and the forms tried here that kept plain fp32 FMAs still missed the
            // oracle on ~0.1 % of the columns, while this one matched on all of them (tests/repair_worker.py).
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < NE; ++i) {
              const uint32_t qv = qq[i];
              asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]\n\ts_nop 0\n\t"
                           "v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]\n\ts_nop 0"
                           : "+v"(acc) : "v"(e[i]), "v"(qv));
            }
            float sim = live ? (float)(half_t)acc : NEG_MASK_F;
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) { const float o = __shfl_xor(sim, sft, 64); sim = o > sim ? o : sim; }
            // (a NaN sim is dropped by `o > sim`: a NaN column keeps what the MFMA pass produced)
            if (lane == col) colmax = sim > colmax ? sim : colmax;
          }
        }
        if (lane < 32 && ((fm >> lane) & 1u)) {
          cv = colmax;
          cm16[slot * Qp + q] = __builtin_bit_cast(uint16_t, (half_t)colmax);
        }
        if (lane == 0) flags[slot * nflag + ch] = 0u;
      }
      float sv = cv;
#pragma unroll
      for (int sft = 32; sft > 0; sft >>= 1) sv += __shfl_xor(sv, sft, 64);
      if ((ch & 1) == 0) {
        if (ch > 0) total = (ch == 2) ? part : total + part;   // previous launch's sum
        part = sv;
      } else {
        part += sv;
      }
    }
    total = (nflag <= 2) ? part : total + part;
    if (lane == 0) {
      exact[slot] = total;
      unc[slot] = 0.f;
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
static int ms_num_cus() {
  static int cus[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 63;
  if (!cus[dev]) {
    hipDeviceProp_t p;
    cus[dev] = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return cus[dev];
}

bool fpk_maxsim_fast_shape(int dim, int nbits) {
  return (dim == 128 || dim == 96 || dim == 64 || dim == 48) && (nbits == 4 || nbits == 2);
}

template <int D, int NBITS>
static void launch_maxsim5(const FpIndexDev& ix, const uint16_t* qpad, const FpSearchShape& sh, const int32_t* sel_pid, const int64_t* pref,
                           int64_t Rcap, float* exact, const FpMaxsimAux& aux, hipStream_t st) {
  using Cf = MsCfg<D, NBITS>;
  const int nch = sh.Qp / 32;
  const int64_t tot_max = (int64_t)sh.B * Rcap;
  int grid = ms_num_cus();
  if ((int64_t)grid * MS_WAVES > tot_max) grid = (int)std::max<int64_t>(1, (tot_max + MS_WAVES - 1) / MS_WAVES);
  static const float eps_rel = [] { const char* e = getenv("FP_MAXSIM_EPS"); const float v = e ? (float)atof(e) : 0.f; return v > 0.f ? v : 1.9073486e-06f; }();   // 2^-19
  MsArgs a{ix.centroids, ix.lut, ix.codes, ix.norms, ix.residuals, ix.doc_off, qpad, sel_pid, pref, exact, aux.cm16, aux.unc, aux.uncm, aux.flags,
           Rcap, sh.B, sh.Q, sh.Qp, 0, 0, eps_rel};
  static std::atomic<uint64_t> ok1{0}, ok2{0};
  fp_allow_big_lds((const void*)k_maxsim5<D, NBITS, 1>, ok1, 96 * 1024);
  fp_allow_big_lds((const void*)k_maxsim5<D, NBITS, 2>, ok2, 96 * 1024);
  for (int ch = 0; ch < nch;) {  // 32-column query chunks: two per launch where possible (the tokens are decompressed once per launch)
    a.ch_begin = ch;
    if (nch - ch >= 2) {
      const size_t lds = MS_LUT_BYTES + (size_t)2 * Cf::KS * 64 * 16 + 2 * 32 * 4;
      hipLaunchKernelGGL((k_maxsim5<D, NBITS, 2>), dim3((unsigned)grid), dim3(MS_THREADS), lds, st, a);
      ch += 2;
    } else {
      const size_t lds = MS_LUT_BYTES + (size_t)1 * Cf::KS * 64 * 16 + 1 * 32 * 4;
      hipLaunchKernelGGL((k_maxsim5<D, NBITS, 1>), dim3((unsigned)grid), dim3(MS_THREADS), lds, st, a);
      ch += 1;
    }
    a.accumulate = 1;
  }
}

// exact scores of the rerank lists.  pref: [B+1] int64 scratch.  aux (all nullable together): per-column maxima, uncertainty
// budgets and flag masks for the exact-order repair.  Returns 0, or -1 when q_len is too large for the generic kernel.
int fpk_maxsim(const FpIndexDev& ix, const uint16_t* qpad, const FpSearchShape& sh, const int32_t* sel_pid, const int32_t* sel_cnt,
               int64_t Rcap, float* exact, int64_t* pref, const FpMaxsimAux& aux, hipStream_t st) {
#define MS_CASE(D_, NB_) \
  if (ix.dim == D_ && ix.nbits == NB_) { \
    hipLaunchKernelGGL(k_cnt_prefix, dim3(1), dim3(256), 0, st, sel_cnt, sh.B, pref); \
    launch_maxsim5<D_, NB_>(ix, qpad, sh, sel_pid, pref, Rcap, exact, aux, st); \
    return 0; \
  }
  MS_CASE(128, 4) MS_CASE(128, 2) MS_CASE(96, 4) MS_CASE(96, 2) MS_CASE(64, 4) MS_CASE(64, 2) MS_CASE(48, 4) MS_CASE(48, 2)
#undef MS_CASE
  const int qcap = (sh.Q + 63) / 64 > 8 ? 8 : (sh.Q + 63) / 64;
  hipLaunchKernelGGL(k_maxsim_generic, dim3((unsigned)Rcap, (unsigned)sh.B), dim3(64), (size_t)ix.dim * 4, st, ix.centroids, ix.lut, ix.codes,
                     ix.norms, ix.residuals, ix.doc_off, ix.dim, ix.nbits, qpad, sh.Q, sh.Qp, sel_pid, sel_cnt, Rcap, exact, aux.unc, qcap);
  // the generic kernel is exact: nothing flagged (aux.unc zeroed above; cm16 / flags are not read when unc == 0)
  return 0;
}

// marks the flagged documents that are near-tied in the final ranking (marks / nmark), or nothing when npow2 keys do not fit LDS
int fpk_final_mark(const float* score, const float* unc, const float* uncm, const int32_t* cnt, int64_t stride, int B, int64_t top_k, int32_t* marks,
                   int32_t* nmark,
                   hipStream_t st) {
  int np2 = fp_next_pow2((int)stride);
  if (np2 < 2) np2 = 2;
  if (np2 > 8192) return -1;   // 16 B of LDS per entry, 16 scan elements per thread
  static std::atomic<uint64_t> lds_ok{0};
  fp_allow_big_lds((const void*)k_final_mark, lds_ok, 136 * 1024);
  hipLaunchKernelGGL(k_final_mark, dim3((unsigned)B), dim3(1024), (size_t)np2 * 16, st, score, unc, uncm, cnt, stride, np2, top_k, marks, nmark);
  return 0;
}

// marks == nullptr: every slot with a budget > 0 is repaired
void fpk_maxsim_repair(const FpIndexDev& ix, const uint16_t* qpad, const FpSearchShape& sh, const int32_t* sel_pid, const int32_t* sel_cnt,
                       int64_t Rcap, const int32_t* marks, const int32_t* nmark, float* exact, const FpMaxsimAux& aux, hipStream_t st) {
  if (!fpk_maxsim_fast_shape(ix.dim, ix.nbits) || !aux.unc) return;   // the generic kernel never flags
  // one wave per document; workgroups beyond a query's marked count exit at once
  const dim3 grid((unsigned)std::min<int64_t>(marks ? 512 : 4096, Rcap), (unsigned)sh.B);
#define MS_CASE(D_, NB_) \
  if (ix.dim == D_ && ix.nbits == NB_) { \
    hipLaunchKernelGGL((k_maxsim_repair<D_, NB_>), grid, dim3(64), 0, st, ix.centroids, ix.lut, ix.codes, ix.norms, ix.residuals, \
                       ix.doc_off, qpad, sh.Q, sh.Qp, sel_pid, sel_cnt, Rcap, marks, nmark, exact, aux.unc, aux.cm16, aux.flags); \
    return; \
  }
  MS_CASE(128, 4) MS_CASE(128, 2) MS_CASE(96, 4) MS_CASE(96, 2) MS_CASE(64, 4) MS_CASE(64, 2) MS_CASE(48, 4) MS_CASE(48, 2)
#undef MS_CASE
}
