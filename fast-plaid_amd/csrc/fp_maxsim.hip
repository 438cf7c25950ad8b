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

// shapes k_maxsim6 is instantiated for (MS6_SHAPES below); an index of such a shape keeps its residual rows in native unit order
static constexpr bool ms6_shape(int dim, int nbits) {
  return (dim == 128 && (nbits == 4 || nbits == 2 || nbits == 8 || nbits == 1)) || (dim == 64 && (nbits == 4 || nbits == 2)) ||
         (dim == 96 && nbits == 4) || (dim == 256 && (nbits == 4 || nbits == 2));
}

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

// ---- per-token reciprocals (k_maxsim6) -----------------------------------------------------------------------------------
// e^ = h(fl32(e / n)) costs the search two fp32 FMAs per value through the compensated reciprocal (norm_pair2).  h(fl32(e * r))
// with ONE fp32 multiplier r gives the same bits for all dims of a token for nearly every r in a window around 1/n about
// 2^-18 wide, i.e. for dozens of fp32 values: this kernel tries r = fl32(1/n) and its neighbours (+-1 .. +-4 ulp) against
// norm_pair2's result on every dim -- the very instruction sequences the search runs -- and stores the first that matches
// everywhere.  A token with no such r (about 1 in 10^4: two dims pinch the window from both sides; also n = 0 / inf / nan) is
// stored with bit 31 set and the search takes the compensated path for the 16-token step that contains it.  +-0 count as
// equal (a zero operand contributes nothing to the contraction); NaNs likewise.
__global__ __launch_bounds__(256) void k_token_rinv(const uint16_t* __restrict__ cent, const uint16_t* __restrict__ lut,
                                                    const int32_t* __restrict__ codes, const uint8_t* __restrict__ resid,
                                                    const uint16_t* __restrict__ norms, int D, int nbits, int64_t T, uint32_t* __restrict__ rinv,
                                                    unsigned long long* __restrict__ n_hard, int hard_every /*testing: every n-th token takes the exact path*/) {
  const int pb = 8 / nbits, pr = D * nbits / 8;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (int64_t)gridDim.x * blockDim.x) {
    const uint16_t* cp = cent + (int64_t)codes[t] * D;
    const uint8_t* rp = resid + t * pr;
    const float nf = (float)__builtin_bit_cast(half_t, norms[t]);
    float r_hi, r_lo;
    recip2(nf, r_hi, r_lo);
    const uint32_t r0 = __float_as_uint(r_hi);
    const bool searchable = (r0 & 0x80000000u) == 0u && ((r0 >> 23) & 0xFFu) != 0u && ((r0 >> 23) & 0xFFu) != 0xFFu;   // positive, normal, finite
    uint32_t found = r0 | 0x80000000u;
    bool ok_any = false;
    if (searchable) {
      for (int ci = 0; ci < 9 && !ok_any; ++ci) {
        const int j = (ci + 1) / 2 * ((ci & 1) ? 1 : -1);   // 0, +1, -1, +2, -2, ...
        const uint32_t rb = r0 + (uint32_t)j;
        if (((rb >> 23) & 0xFFu) == 0u || ((rb >> 23) & 0xFFu) == 0xFFu) continue;
        const float r = __uint_as_float(rb);
        bool ok = true;
        for (int d = 0; d < D && ok; d += 4) {   // D is a multiple of 8: two packed pairs per round
          uint32_t pk[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            uint32_t two = 0;
#pragma unroll
            for (int x = 0; x < 2; ++x) {
              const int dd = d + 2 * h + x;
              const int byte = rp[dd / pb];
              const half_t w = __builtin_bit_cast(half_t, lut[byte * pb + dd % pb]);
              const half_t c = __builtin_bit_cast(half_t, cp[dd]);
              const half_t e = (half_t)((float)w + (float)c);
              two |= (uint32_t)__builtin_bit_cast(uint16_t, e) << (16 * x);
            }
            pk[h] = two;
          }
          uint32_t a0 = pk[0], a1 = pk[1], b0 = pk[0], b1 = pk[1];
          norm_pair2(a0, a1, r_hi, r_lo);
          norm_mul2(b0, b1, r);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t x = h ? a1 : a0, y = h ? b1 : b0;
#pragma unroll
            for (int z = 0; z < 2; ++z) {
              const uint32_t u = (x >> (16 * z)) & 0xFFFFu, v = (y >> (16 * z)) & 0xFFFFu;
              const bool both_zero = ((u | v) & 0x7FFFu) == 0u;
              const bool both_nan = (u & 0x7FFFu) > 0x7C00u && (v & 0x7FFFu) > 0x7C00u;
              if (u != v && !both_zero && !both_nan) ok = false;
            }
          }
        }
        if (ok) { ok_any = true; found = rb; }
      }
    }
    if (hard_every > 0 && t % hard_every == 0) { ok_any = false; found |= 0x80000000u; }
    rinv[t] = found;
    if (!ok_any && n_hard) atomicAdd(n_hard, 1ull);
  }
}

void fpk_token_rinv(const FpIndexDev& ix, uint32_t* rinv, unsigned long long* n_hard_dev, hipStream_t st) {
  if (ix.T <= 0) return;
  static const int hard_every = (int)fp_test_opt("ms_rinv_hard_every", 0);
  hipLaunchKernelGGL(k_token_rinv, dim3(fp_grid_cap((ix.T + 255) / 256, 256)), dim3(256), 0, st, ix.centroids, ix.lut, ix.codes, ix.residuals,
                     ix.norms, ix.dim, ix.nbits, ix.T, rinv, n_hard_dev, hard_every);
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
  const uint32_t* rinv;       // k_maxsim6: per-token reciprocal bits for the one-multiply normalisation (bit 31: take the exact path); nullable
  int xcd;                    // k_maxsim6: 1 = workgroup -> share mapping that keeps a query's documents on one XCD
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
      ms_load_token<D, NBITS>(a.resid, a.cent, valid ? tok_row() : 0, valid ? code : 0, rw, e);
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
      ms_decode<D, NBITS>(laneoff, rw, e);
      float r_hi, r_lo;
      recip2(nf, r_hi, r_lo);
#pragma unroll
      for (int i = 0; i < NE; i += 2) norm_pair2(e[i], e[i + 1], r_hi, r_lo);
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

// ======================================================================================================================
// k_maxsim6 -- the MFMA-native layout (round 3).
//
// What bounded k_maxsim5 (profiles/r03_gather_lab.txt, r03_maxsim5_ablations.txt): its token-per-lane gather -- 16 dwordx4
// loads per 64 tokens, every one of them touching 64 different cache lines -- takes 246 us for the cfg2 rerank set WITHOUT any
// arithmetic (the texture path pays per instruction and line), and the kernel ran at exactly that; dropping the normalisation
// or the decode from it changed nothing.  The same bytes fetched as 64-byte chunks of a row per instruction take ~150 us.
//
// Layout: v_mfma_f32_16x16x32_f16 wants from lane (r = lane % 16, g = lane / 16) eight consecutive k values of row r.  A step
// is 16 tokens (rows); k-step s of the lane covers dims 32 s + 8 g .. + 8, so that
//   * one dwordx4 per k-step fetches the lane's 16 bytes of its token's centroid row, the four lanes of a row reading 64
//     contiguous bytes (4 instructions x 16 lines per 16 tokens instead of 16 x 64 per 64);
//   * the decompressed, normalised values ARE the A operand: no lane swaps, no transposition through LDS;
//   * the residual bytes of the lane's 8-dim units sit next to each other in the index ("native unit order",
//     fp_resid_native_pos: unit 4 s + g is stored at g * KS4 + s), so the lane's residuals are ONE load and the 16 tokens of
//     a step one contiguous kilobyte;
//   * the norm / reciprocal of the token is the same for every value the lane holds.
// A lane holds 20 registers of loads per step, so three steps are kept in flight (codes one step further ahead): 16 waves x
// 2 x 5 KiB of loads outstanding per CU while a step is being computed.
// Everything downstream of the fp32 accumulators (fp32 column maximum, rounding, certification window, flags, budgets, the
// per-document sum) is k_maxsim5's.
template <int KS4, int NBITS>
struct Ms6Cfg {
  static constexpr int D = KS4 * 32;                  // dims (multiple of 32)
  static constexpr int PR = D * NBITS / 8;            // residual bytes per token
  static constexpr int LB = KS4 * NBITS;              // residual bytes per lane: its KS4 units of 8 dims
  static constexpr int RW = (LB + 3) / 4;             // ... as 32-bit words
  static constexpr int PB = 8 / NBITS;                // dims per byte
  static constexpr int NE = KS4 * 4;                  // half2 registers per lane (8 dims per k-step)
  static constexpr int EW = PB >= 2 ? PB / 2 : 1;     // 32-bit words per LUT entry
  static constexpr int COPIES = 64 / EW;
  static_assert(LB % 4 == 0, "lane residual chunk must be whole words");
};

template <int NBITS>
__device__ __forceinline__ void ms6_fill_lut(unsigned char* lds, const uint16_t* __restrict__ lut_g, int tid, int nthreads) {
  constexpr int PB = 8 / NBITS, EW = PB >= 2 ? PB / 2 : 1, COPIES = 64 / EW;
  uint32_t* l32 = reinterpret_cast<uint32_t*>(lds);
  for (int i = tid; i < 256 * COPIES; i += nthreads) {
    const int entry = i / COPIES, copy = i % COPIES;
#pragma unroll
    for (int w = 0; w < EW; ++w) {
      uint32_t word;
      if constexpr (NBITS == 8) word = (uint32_t)lut_g[entry];
      else word = reinterpret_cast<const uint32_t*>(lut_g)[entry * EW + w];
      l32[(entry * 256 + copy * EW * 4) / 4 + w] = word;
    }
  }
}

// e += weights of the lane's residual bytes (same table addressing as ms_decode: one v_perm per byte)
template <int NBITS, int RW, int NE>
__device__ __forceinline__ void ms6_decode(uint32_t laneoff, const uint32_t (&rw)[RW], uint32_t (&e)[NE]) {
#pragma unroll
  for (int w = 0; w < RW; ++w) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int bi = w * 4 + kk;
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
      } else {   // NBITS == 8
        if ((kk & 1) == 0) {
          const uint32_t addr1 = __builtin_amdgcn_perm(rw[w], laneoff, 0x0C0C0400u | ((uint32_t)(kk + 1) << 8));
          const uint32_t lo = *(ms_lds_u32*)(uintptr_t)addr;
          const uint32_t hi = *(ms_lds_u32*)(uintptr_t)addr1;
          e[bi / 2] = h2_as_u32(u32_as_h2(e[bi / 2]) + u32_as_h2(__builtin_amdgcn_perm(hi, lo, 0x05040100u)));
        }
      }
    }
  }
}

typedef float f4v __attribute__((ext_vector_type(4)));

template <int KS4, int NBITS>
struct Ms6Buf {
  uint32_t e[Ms6Cfg<KS4, NBITS>::NE];    // centroid half2 pairs of the lane's dims, k-step major (decoded in place)
  uint32_t rw[Ms6Cfg<KS4, NBITS>::RW];   // the lane's residual bytes
};

struct Ms6Step {   // one 16-token step of one document; wave-uniform
  long long off;   // first token row of the document
  int t0, len, r;  // first token of the step, document length, rerank slot
  int valid;
};

// waves per workgroup (one workgroup per CU): 16 where three steps of loads fit 128 registers, fewer for wide rows / two column chunks
static constexpr int ms6_waves(int ks4, int nch, int nbits = 4) { return ks4 >= 8 ? 8 : ((ks4 >= 4 && (nch == 2 || nbits == 8)) ? 12 : 16); }

template <int KS4, int NBITS, int NCH>
__global__ __launch_bounds__(ms6_waves(KS4, NCH, NBITS) * 64) __attribute__((amdgpu_waves_per_eu(ms6_waves(KS4, NCH, NBITS) / 4, ms6_waves(KS4, NCH, NBITS) / 4))) void k_maxsim6(const MsArgs a) {
  using Cf = Ms6Cfg<KS4, NBITS>;
  constexpr int MS6_WAVES = ms6_waves(KS4, NCH, NBITS), MS6_THREADS = MS6_WAVES * 64;
  constexpr int D = Cf::D, NE = Cf::NE, RW = Cf::RW, PR = Cf::PR, LB = Cf::LB;
  constexpr int NC16 = NCH * 2;   // 16-column groups per launch
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* lut = smem;                                                          // 64 KiB
  uint4* qs = reinterpret_cast<uint4*>(smem + MS_LUT_BYTES);                         // B fragments: [NC16][KS4][64] x 16 B
  float* qn = reinterpret_cast<float*>(smem + MS_LUT_BYTES + NC16 * KS4 * 64 * 16);  // [NCH * 32] certification window per column
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r16 = lane & 15, g = lane >> 4;
  ms_lds_base_is_zero(smem);
  ms6_fill_lut<NBITS>(lut, a.lut_g, tid, MS6_THREADS);
  const uint32_t laneoff = (uint32_t)((lane & (Cf::COPIES - 1)) * Cf::EW * 4);
  const half_t negm = (half_t)NEG_MASK_F;
  int nq = a.Q - a.ch_begin * 32;
  nq = nq < 0 ? 0 : (nq > NCH * 32 ? NCH * 32 : nq);
  const int nflag = a.Qp / 32;

  // this workgroup's share of the flattened (query, rerank slot) space.  xcd: workgroup i runs on XCD i % 8; giving XCD x the
  // x-th eighth of the space keeps a query's documents (and the centroid rows they share) in ONE L2
  int bid = blockIdx.x;
  if (a.xcd && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const long long tot = a.pref[a.B];
  const long long lo = tot * (long long)bid / (long long)gridDim.x;
  const long long hi_end = tot * (long long)(bid + 1) / (long long)gridDim.x;
  int b = 0;
  {
    int l = 0, h = a.B;
    while (h - l > 1) { const int m = (l + h) >> 1; if (a.pref[m] <= lo) l = m; else h = m; }
    b = l;
  }
  for (; b < a.B && a.pref[b] < hi_end; ++b) {
    const long long pb0 = a.pref[b], pb1 = a.pref[b + 1];
    const int ra = (int)((lo > pb0 ? lo : pb0) - pb0);
    const int rb = (int)((hi_end < pb1 ? hi_end : pb1) - pb0);
    if (rb <= ra) continue;
    __syncthreads();
    // B fragment of (16-column group c, k-step s), lane (n, gq): q[col = 16 c + n][dims 32 s + 8 gq .. + 8]
    for (int i = tid; i < NC16 * KS4 * 64; i += MS6_THREADS) {
      const int ln = i & 63, s = (i >> 6) % KS4, c = (i >> 6) / KS4;
      const int col = a.ch_begin * 32 + c * 16 + (ln & 15);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (col < a.Qp) v = *reinterpret_cast<const uint4*>(a.qpad + ((int64_t)b * a.Qp + col) * D + 32 * s + 8 * (ln >> 4));
      qs[i] = v;
    }
    if (tid < NCH * 32) {
      const int col = a.ch_begin * 32 + tid;
      float ss = 0.f;
      if (col < a.Qp) {
        const uint16_t* qp = a.qpad + ((int64_t)b * a.Qp + col) * D;
        for (int k = 0; k < D; ++k) { const float x = (float)__builtin_bit_cast(half_t, qp[k]); ss = __builtin_fmaf(x, x, ss); }
      }
      qn[tid] = a.eps_rel * __builtin_sqrtf(ss);
    }
    __syncthreads();
    const int per = (rb - ra + MS6_WAVES - 1) / MS6_WAVES;
    const int r0 = ra + wave * per;
    if (r0 >= rb) continue;
    const int rend = (r0 + per < rb) ? (r0 + per) : rb;
    const int32_t* selp = a.sel_pid + (int64_t)b * a.Rcap;
    float* outp = a.exact + (int64_t)b * a.Rcap;

    // ---- the stream of 16-token steps of documents r0 .. rend-1 (all wave-uniform).  The document metadata comes through the
    // scalar cache (constant address space: the lists were written by earlier kernels), so that no vector load -- and with it no
    // vmcnt(0) -- sits between the pipelined row loads ----
    typedef const __attribute__((address_space(4))) int32_t* ms_cptr_i32;
    typedef const __attribute__((address_space(4))) int64_t* ms_cptr_i64;
    const ms_cptr_i32 selc = (ms_cptr_i32)(uintptr_t)selp;
    const ms_cptr_i64 doffc = (ms_cptr_i64)(uintptr_t)a.doc_off;
    auto meta = [&](int rr, long long& o, int& l) {
      const int32_t pid = selc[rr];
      o = doffc[pid];
      l = (int)(doffc[pid + 1] - o);
    };
    // empty documents first (every column keeps the masked value, nothing to flag); the stream below skips them
    for (int rr = r0; rr < rend; ++rr) {
      long long o; int l;
      meta(rr, o, l);
      if (l != 0) continue;
      if (lane == 0) {
        const float v = (float)nq * NEG_MASK_F;
        outp[rr] = a.accumulate ? (outp[rr] + v) : v;
        if (a.unc && !a.accumulate) { a.unc[(int64_t)b * a.Rcap + rr] = 0.f; a.uncm[(int64_t)b * a.Rcap + rr] = 0.f; }
      }
      if (a.cm16 && lane < 32) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
          if ((a.ch_begin + c) * 32 < a.Qp) a.cm16[((int64_t)b * a.Rcap + rr) * a.Qp + (a.ch_begin + c) * 32 + lane] = __builtin_bit_cast(uint16_t, negm);
      }
      if (a.flags && lane == 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
          if (a.ch_begin + c < nflag) a.flags[((int64_t)b * a.Rcap + rr) * nflag + a.ch_begin + c] = 0u;
      }
    }
    int it_r = r0, it_t0 = 0, it_len = 0, n_len = 0;
    long long it_off = 0, n_off = 0;
    if (it_r < rend) meta(it_r, it_off, it_len);
    if (it_r + 1 < rend) meta(it_r + 1, n_off, n_len);
    auto next_doc = [&]() {
      ++it_r;
      it_off = n_off;
      it_len = n_len;
      it_t0 = 0;
      if (it_r + 1 < rend) meta(it_r + 1, n_off, n_len);
    };
    while (it_r < rend && it_len == 0) next_doc();
    auto next_step = [&]() -> Ms6Step {   // the iterator's current step, then advance
      Ms6Step st{it_off, it_t0, it_len, it_r, it_r < rend ? 1 : 0};
      if (st.valid) {
        it_t0 += 16;
        if (it_t0 >= it_len) {
          next_doc();
          while (it_r < rend && it_len == 0) next_doc();
        }
      }
      return st;
    };
    // the lane's token inside the step's document (clamped: loads stay in bounds, rows beyond the document are masked at the maximum)
    auto tok_of = [&](const Ms6Step& st) -> int {
      const int tok = st.t0 + r16;
      return st.valid ? (tok < st.len ? tok : st.len - 1) : 0;
    };
    auto load_tok = [&](Ms6Buf<KS4, NBITS>& bf, const Ms6Step& st, int32_t code) {
      const long long row = (st.valid ? st.off : 0) + tok_of(st);
      const uint8_t* rp = a.resid + row * (long long)PR + g * LB;
      if constexpr (LB % 16 == 0) {
#pragma unroll
        for (int i = 0; i < LB / 16; ++i) {
          const uint4 v = *reinterpret_cast<const uint4*>(rp + 16 * i);
          bf.rw[4 * i] = v.x; bf.rw[4 * i + 1] = v.y; bf.rw[4 * i + 2] = v.z; bf.rw[4 * i + 3] = v.w;
        }
      } else if constexpr (LB % 8 == 0) {
#pragma unroll
        for (int i = 0; i < LB / 8; ++i) {
          const uint2 v = *reinterpret_cast<const uint2*>(rp + 8 * i);
          bf.rw[2 * i] = v.x; bf.rw[2 * i + 1] = v.y;
        }
      } else {
#pragma unroll
        for (int i = 0; i < LB / 4; ++i) bf.rw[i] = *reinterpret_cast<const uint32_t*>(rp + 4 * i);
      }
      const uint16_t* cp = a.cent + (long long)code * D + 8 * g;
#pragma unroll
      for (int s = 0; s < KS4; ++s) {
        const uint4 v = *reinterpret_cast<const uint4*>(cp + 32 * s);
        bf.e[4 * s] = v.x; bf.e[4 * s + 1] = v.y; bf.e[4 * s + 2] = v.z; bf.e[4 * s + 3] = v.w;
      }
    };
    auto load_cn = [&](const Ms6Step& st, int32_t& code, uint32_t& nrm) {   // nrm: reciprocal bits (a.rinv) or the fp16 norm
      const long long row = (st.valid ? st.off : 0) + tok_of(st);
      code = a.codes[row];
      nrm = a.rinv ? a.rinv[row] : (uint32_t)a.norms[row];
    };

    float mx[NC16];
#pragma unroll
    for (int c = 0; c < NC16; ++c) mx[c] = NEG_MASK_F;

    auto compute = [&](const Ms6Step& st, Ms6Buf<KS4, NBITS>& bf, uint32_t nrm) {
      // e = h(cent + w);  e^ = h(fl32(e / n))
      ms6_decode<NBITS, RW, NE>(laneoff, bf.rw, bf.e);
      if (a.rinv && !__any((int)(nrm >> 31))) {   // one multiply per value by the token's stored reciprocal
        const float r = __uint_as_float(nrm);
#pragma unroll
        for (int i = 0; i < NE; i += 2) norm_mul2(bf.e[i], bf.e[i + 1], r);
      } else {   // no reciprocals in the index, or a token of the step has none: the compensated quotient
        uint16_t n16 = (uint16_t)nrm;
        if (a.rinv) n16 = a.norms[(st.valid ? st.off : 0) + tok_of(st)];
        float r_hi, r_lo;
        recip2((float)__builtin_bit_cast(half_t, n16), r_hi, r_lo);
#pragma unroll
        for (int i = 0; i < NE; i += 2) norm_pair2(bf.e[i], bf.e[i + 1], r_hi, r_lo);
      }
      // MFMA: D[row = token][col], this lane's rows 4 g + i
      f4v acc[NC16];
#pragma unroll
      for (int c = 0; c < NC16; ++c) {
#pragma unroll
        for (int s = 0; s < KS4; ++s) {
          const h8 av = __builtin_bit_cast(h8, make_uint4(bf.e[4 * s], bf.e[4 * s + 1], bf.e[4 * s + 2], bf.e[4 * s + 3]));
          const h8 bq = __builtin_bit_cast(h8, qs[(c * KS4 + s) * 64 + lane]);
          if (s == 0) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bq, f4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          else acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bq, acc[c], 0, 0, 0);
        }
      }
      const bool partial = st.t0 + 16 > st.len;
#pragma unroll
      for (int c = 0; c < NC16; ++c) {
        float v0 = acc[c][0], v1 = acc[c][1], v2 = acc[c][2], v3 = acc[c][3];
        if (partial) {
          const int rowb = st.t0 + 4 * g;
          if (rowb + 0 >= st.len) v0 = NEG_MASK_F;
          if (rowb + 1 >= st.len) v1 = NEG_MASK_F;
          if (rowb + 2 >= st.len) v2 = NEG_MASK_F;
          if (rowb + 3 >= st.len) v3 = NEG_MASK_F;
        }
        mx[c] = max3_raw(v2, v3, max3_raw(v0, v1, mx[c]));
      }
      if (st.t0 + 16 >= st.len) {   // the document's last step: combine the four row groups, round, certify, sum
        float total = 0.f, ubud = 0.f, ubm = 0.f;
        const unsigned long long upper16 = 0xFFFF0000FFFF0000ull;   // lanes with (lane & 16)
#pragma unroll
        for (int c = 0; c < NC16; ++c) {
          {  // maximum over lanes l, l^16, l^32, l^48 (the same column, other row groups)
            auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx[c]), __float_as_uint(mx[c]), false, false);
            float m = __builtin_fmaxf(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
            auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            mx[c] = __builtin_fmaxf(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
          }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          // lane l < 32 takes column 32 c + l: lanes 0-15 from group 2c, lanes 16-31 from group 2c + 1 (every row group holds both)
          float am;   // (a plain ?: over two array elements becomes an indexed scratch access; spell the select)
          asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(am) : "v"(mx[2 * c]), "v"(mx[2 * c + 1]), "s"(upper16));
          const int q = (a.ch_begin + c) * 32 + (lane & 31);
          const bool mine = (lane < 32) && (q < a.Q);
          const half_t hm = (half_t)am;
          float sv = mine ? (float)hm : 0.f;
          uint32_t ef = (__float_as_uint(am) >> 23) & 0xFFu;
          ef = ef < 113u ? 113u : ef;
          const float halfulp = __uint_as_float((ef - 11u) << 23);
          const float dist = halfulp - __builtin_fabsf(am - (float)hm);
          const bool flag = mine && !(dist > qn[c * 32 + (lane & 31)]);
          float fu = flag ? 2.0f * halfulp : 0.f;
          float fm = (flag && !(am > (float)hm)) ? 2.0f * halfulp : 0.f;
          if (flag && !(am < (float)hm) && !(am > (float)hm)) fu += 2.0f * halfulp;
          const unsigned long long bal = __ballot(flag);
#pragma unroll
          for (int sft = 32; sft > 0; sft >>= 1) sv += __shfl_xor(sv, sft, 64);
          if (bal) {
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) {
              fu += __shfl_xor(fu, sft, 64);
              fm += __shfl_xor(fm, sft, 64);
            }
          }
          total += sv;
          ubud += fu;
          ubm += fm;
          if (a.cm16 && lane < 32 && (a.ch_begin + c) * 32 < a.Qp)
            a.cm16[((int64_t)b * a.Rcap + st.r) * a.Qp + (a.ch_begin + c) * 32 + lane] = __builtin_bit_cast(uint16_t, hm);
          if (a.flags && lane == 0 && a.ch_begin + c < nflag) a.flags[((int64_t)b * a.Rcap + st.r) * nflag + a.ch_begin + c] = (uint32_t)bal;
        }
        if (lane == 0) {
          outp[st.r] = a.accumulate ? (outp[st.r] + total) : total;
          if (a.unc) {
            float* up = a.unc + (int64_t)b * a.Rcap + st.r;
            *up = a.accumulate ? (*up + ubud) : ubud;
            float* um = a.uncm + (int64_t)b * a.Rcap + st.r;
            *um = a.accumulate ? (*um + ubm) : ubm;
          }
        }
#pragma unroll
        for (int c = 0; c < NC16; ++c) mx[c] = NEG_MASK_F;
      }
    };

    // ---- software pipeline: step k is computed while the loads of steps k+1, k+2 are in flight and the code / norm of
    // step k+3 are being fetched.  vmcnt retires in order, so a step's code load must be OLDER than the previous step's
    // row loads (or waiting for it would drain them): order per iteration = codes(k+3), rows(k+2), compute(k). ----
    Ms6Buf<KS4, NBITS> B0, B1, B2;
    Ms6Step d0 = next_step(), d1 = next_step(), d2 = next_step(), d3{};
    int32_t code0, code1, code2, code3 = 0;
    uint32_t nrm0, nrm1, nrm2, nrm3 = 0;
    load_cn(d0, code0, nrm0);
    load_cn(d1, code1, nrm1);
    load_cn(d2, code2, nrm2);
    asm volatile("" ::: "memory");
    load_tok(B0, d0, code0);   // (unconditional: an exhausted stream reads row 0 -- every path then has the same number of
    load_tok(B1, d1, code1);   //  loads in flight, which is what lets the compiler wait with vmcnt(n) instead of vmcnt(0))
    asm volatile("" ::: "memory");
    // (the empty asm statements with a memory clobber pin the ORDER in which the loads are issued: without them the compiler
    // sinks a step's code load next to its first use, one iteration later, and then has to drain every load in front of it)
    auto body = [&](Ms6Buf<KS4, NBITS>& cur, Ms6Buf<KS4, NBITS>& ld) {
      d3 = next_step();
      load_cn(d3, code3, nrm3);
      asm volatile("" ::: "memory");
      load_tok(ld, d2, code2);
      asm volatile("" ::: "memory");
      compute(d0, cur, nrm0);
      asm volatile("" ::: "memory");
      d0 = d1; d1 = d2; d2 = d3;
      nrm0 = nrm1; nrm1 = nrm2; nrm2 = nrm3;
      code2 = code3;
    };
    while (true) {
      if (!d0.valid) break;
      body(B0, B2);
      if (!d0.valid) break;
      body(B1, B0);
      if (!d0.valid) break;
      body(B2, B1);
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
                                                     int32_t* __restrict__ nmark, uint32_t* __restrict__ flat_n /*nullable: zeroed counter*/,
                                                     uint2* __restrict__ flat /*... and the batch-wide work list {query, slot} it indexes*/) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* v = reinterpret_cast<unsigned long long*>(smem);          // [npow2] keys
  float* lo = reinterpret_cast<float*>(smem + (size_t)npow2 * 8);              // [npow2] s - u, then its exclusive prefix-min
  float* hi = lo + npow2;                                                        // [npow2] s + u, then its exclusive suffix-max
  __shared__ int s_any, s_n;
  const int b = blockIdx.x;
  const int n = cnt[b];
  if (threadIdx.x == 0) { s_any = 0; s_n = 0; }
  if (npow2 == 1024 && blockDim.x == 1024) {
    // One document per thread (the usual rerank list).  Everything the kernel reads from global memory is fetched up front in slot
    // order (the budgets used to be gathered through the sorted slots twice: two more dependent round trips), and the two scans
    // are wave scans plus one combine over the sixteen wave totals (Hillis-Steele over 1024 elements was twenty barriers).
    __shared__ float s_u[1024], s_um[1024], s_wl[16], s_wh[16];
    const int i = (int)threadIdx.x, lane = i & 63, wave = i >> 6;
    const int64_t o = (int64_t)b * stride + i;
    const float sc = i < n ? score[o] : 0.f;
    const float u0 = i < n ? unc[o] : 0.f;
    const float um0 = (uncm && i < n) ? uncm[o] : u0;
    s_u[i] = u0;
    s_um[i] = um0;
    __syncthreads();   // (also orders the s_any / s_n reset)
    if (i < n && u0 > 0.f) s_any = 1;
    __syncthreads();
    if (s_any) {   // (uniform)
      const unsigned long long key0 = i < n ? (((unsigned long long)mono32(sc) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i)) : 0ull;
      const unsigned long long key = fp_sort1024_desc(key0, v);
      const float inf = __builtin_inff();
      const int slot = (int)(0xFFFFFFFFu - (uint32_t)key);
      const float si = unmono32((uint32_t)(key >> 32));
      float u = 0.f, um = 0.f, l = inf, h = -inf;
      if (i < n) {
        u = s_u[slot];
        um = s_um[slot];
        l = si - um;
        h = si + (uncm ? u - um : u);
      }
      // inclusive prefix-min of l / suffix-max of h inside the wave
      float pl = l, ph = h;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const float a = __shfl_up(pl, off, 64), c = __shfl_down(ph, off, 64);
        if (lane >= off) pl = __builtin_fminf(pl, a);
        if (lane + off < 64) ph = __builtin_fmaxf(ph, c);
      }
      if (lane == 63) s_wl[wave] = pl;
      if (lane == 0) s_wh[wave] = ph;
      __syncthreads();
      float bl = inf, bh = -inf;
      for (int w = 0; w < wave; ++w) bl = __builtin_fminf(bl, s_wl[w]);
      for (int w = wave + 1; w < 16; ++w) bh = __builtin_fmaxf(bh, s_wh[w]);
      lo[i] = __builtin_fminf(pl, bl);
      hi[i] = __builtin_fmaxf(ph, bh);
      __syncthreads();
      const int kk = (int)(top_k < n ? top_k : n);   // emitted positions [0, kk)
      const float lmin_top = kk > 0 ? lo[kk - 1] : inf;
      if (i < n && u > 0.f) {
        const float up = uncm ? u - um : u;
        bool conflict;
        if (i < kk) {
          const float lmin = i > 0 ? lo[i - 1] : inf;
          const float hmax = i + 1 < 1024 ? hi[i + 1] : -inf;
          conflict = !(si + up < lmin) || !(si - um > hmax);
          conflict = conflict || um > 0.00095f || up > 0.00095f;   // (see the general path below)
        } else {
          conflict = !(si + up < lmin_top);
        }
        if (conflict) marks[(int64_t)b * stride + atomicAdd(&s_n, 1)] = slot;
      }
    }
  } else {
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
        // ... and an emitted document whose flagged columns could add up to more than the score tolerance (three or more of them
        // on one side in the top binade: rare at 32 query tokens, less so at 128) is re-evaluated too, so that EVERY returned score
        // is within 1e-3 of the reference's by construction, not by the odds
        conflict = conflict || um > 0.00095f || up > 0.00095f;
      } else {
        conflict = !(si + up < lmin_top);                         // outside the emitted range: only a jump into it matters
      }
      if (conflict) marks[(int64_t)b * stride + atomicAdd(&s_n, 1)] = slot;
    }
  }
  }
  __syncthreads();
  if (threadIdx.x == 0) nmark[b] = s_n;
  if (flat_n) {
    // The marked documents of ALL queries on one list (any order: documents are repaired independently), so that the repair
    // kernel's waves share them evenly whatever their spread over the queries (135 per query on average at cfg2).
    // (The repair itself is bound by re-reading the marked documents' tokens -- 41 KB of centroid rows and residuals per
    // 128-token document, 5-6.6 TB/s measured with 8.7 k and 32 k documents per batch -- not by the 1.4 flagged columns per
    // document it re-evaluates: tools/flag_stats.py.)
    __shared__ uint32_t s_base;
    const int cntm = s_n;
    if (threadIdx.x == 0) s_base = cntm > 0 ? atomicAdd(flat_n, (uint32_t)cntm) : 0u;
    __syncthreads();
    for (int i = threadIdx.x; i < cntm; i += blockDim.x) flat[s_base + i] = make_uint2((uint32_t)b, (uint32_t)marks[(int64_t)b * stride + i]);
  }
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
                                                      uint16_t* __restrict__ cm16, uint32_t* __restrict__ flags,
                                                      const uint32_t* __restrict__ flat_n, const uint2* __restrict__ flat) {
  using Cf = MsCfg<D, NBITS>;
  constexpr int native = ms6_shape(D, NBITS) ? 1 : 0;   // the stored unit order of this shape's residual rows
  constexpr int NE = Cf::NE, RW = Cf::RW, PB = Cf::PB;
  __shared__ uint16_t slut[256 * PB];
  // work items: the batch-wide list (1-D grid), or per query (blockIdx.y) the marked slots / every slot
  int b = flat ? 0 : (int)blockIdx.y;
  const int nwork = flat ? (int)*flat_n : (marks ? nmark[b] : sel_cnt[b]);
  if ((int)blockIdx.x >= nwork) return;
  const int lane = threadIdx.x;
  for (int i = lane; i < 256 * PB; i += 64) slut[i] = lut_g[i];
  __syncthreads();
  const int nflag = Qp / 32;
  for (int wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
    int r;
    if (flat) {
      const uint2 it = flat[wi];
      b = (int)it.x;
      r = (int)it.y;
    } else {
      r = marks ? marks[(int64_t)b * Rcap + wi] : wi;
    }
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
              const int bi = fp_resid_logical(w * 4 + kk, NBITS, D / 8, native);   // (a constant after unrolling)
              const uint32_t byte = (rw[w] >> (8 * kk)) & 0xFFu;
              if constexpr (PB >= 2) {
#pragma unroll
                for (int j = 0; j < PB / 2; ++j) {
                  const uint32_t wv = *reinterpret_cast<const uint32_t*>(&slut[byte * PB + 2 * j]);
                  e[bi * (PB / 2) + j] = h2_as_u32(u32_as_h2(e[bi * (PB / 2) + j]) + u32_as_h2(wv));
                }
              } else if ((kk & 1) == 0) {   // nbits 8: two bytes make one packed register (a unit is 8 bytes: pairs stay together)
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
  return ((dim == 96 || dim == 48) && (nbits == 4 || nbits == 2)) || ms6_shape(dim, nbits);
}
bool fpk_maxsim6_shape(int dim, int nbits) { return ms6_shape(dim, nbits); }

__global__ __launch_bounds__(256) void k_resid_native(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int64_t n, int pr, int nbits, int nu) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * nu; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / nu;
    const int u = (int)(i - t * nu);
    const int pos = fp_resid_native_unit(u, nu);
    for (int w = 0; w < nbits; ++w) dst[t * pr + pos * nbits + w] = src[t * pr + u * nbits + w];
  }
}
void fpk_resid_native(const FpIndexDev& ix, int64_t t0, int64_t n, uint8_t* tmp, hipStream_t st) {
  if (n <= 0) return;
  const int nu = ix.dim / 8;
  uint8_t* rows = const_cast<uint8_t*>(ix.residuals) + t0 * ix.pr;
  hipLaunchKernelGGL(k_resid_native, dim3(fp_grid_cap((n * nu + 255) / 256, 256)), dim3(256), 0, st, rows, tmp, n, ix.pr, ix.nbits, nu);
  (void)hipMemcpyAsync(rows, tmp, (size_t)n * ix.pr, hipMemcpyDeviceToDevice, st);
}

template <int KS4, int NBITS>
static void launch_maxsim6(const FpIndexDev& ix, const uint16_t* qpad, const FpSearchShape& sh, const int32_t* sel_pid, const int64_t* pref,
                           int64_t Rcap, float* exact, const FpMaxsimAux& aux, hipStream_t st) {
  const int nch = sh.Qp / 32;
  const int64_t tot_max = (int64_t)sh.B * Rcap;
  auto grid_for = [&](int waves) {
    int g = ms_num_cus();
    if ((int64_t)g * waves > tot_max) g = (int)std::max<int64_t>(1, (tot_max + waves - 1) / waves);
    return g;
  };
  const float eps_env = 1.9073486e-06f;   // 2^-19
  const float eps_rel = eps_env * (ix.dim > 128 ? (float)ix.dim / 128.f : 1.f);   // (the window was measured at dim 128; the reorder noise grows with the number of terms)
  const int xcd = 0;   // (a query's documents on one XCD: measured -1 us, round 3)
  MsArgs a{ix.centroids, ix.lut, ix.codes, ix.norms, ix.residuals, ix.doc_off, qpad, sel_pid, pref, exact, aux.cm16, aux.unc, aux.uncm, aux.flags,
           Rcap, sh.B, sh.Q, sh.Qp, 0, 0, eps_rel, ix.rinv, xcd};
  static std::atomic<uint64_t> ok1{0}, ok2{0};
  fp_allow_big_lds((const void*)k_maxsim6<KS4, NBITS, 1>, ok1, 96 * 1024);
  fp_allow_big_lds((const void*)k_maxsim6<KS4, NBITS, 2>, ok2, 112 * 1024);
  for (int ch = 0; ch < nch;) {
    a.ch_begin = ch;
    if (nch - ch >= 2) {
      const size_t lds = MS_LUT_BYTES + (size_t)4 * KS4 * 64 * 16 + 2 * 32 * 4;
      hipLaunchKernelGGL((k_maxsim6<KS4, NBITS, 2>), dim3((unsigned)grid_for(ms6_waves(KS4, 2, NBITS))), dim3(ms6_waves(KS4, 2, NBITS) * 64), lds, st, a);
      ch += 2;
    } else {
      const size_t lds = MS_LUT_BYTES + (size_t)2 * KS4 * 64 * 16 + 1 * 32 * 4;
      hipLaunchKernelGGL((k_maxsim6<KS4, NBITS, 1>), dim3((unsigned)grid_for(ms6_waves(KS4, 1, NBITS))), dim3(ms6_waves(KS4, 1, NBITS) * 64), lds, st, a);
      ch += 1;
    }
    a.accumulate = 1;
  }
}

template <int D, int NBITS>
static void launch_maxsim5(const FpIndexDev& ix, const uint16_t* qpad, const FpSearchShape& sh, const int32_t* sel_pid, const int64_t* pref,
                           int64_t Rcap, float* exact, const FpMaxsimAux& aux, hipStream_t st) {
  using Cf = MsCfg<D, NBITS>;
  const int nch = sh.Qp / 32;
  const int64_t tot_max = (int64_t)sh.B * Rcap;
  int grid = ms_num_cus();
  if ((int64_t)grid * MS_WAVES > tot_max) grid = (int)std::max<int64_t>(1, (tot_max + MS_WAVES - 1) / MS_WAVES);
  const float eps_env = 1.9073486e-06f;   // 2^-19
  const float eps_rel = eps_env * (ix.dim > 128 ? (float)ix.dim / 128.f : 1.f);   // (the window was measured at dim 128; the reorder noise grows with the number of terms)
  MsArgs a{ix.centroids, ix.lut, ix.codes, ix.norms, ix.residuals, ix.doc_off, qpad, sel_pid, pref, exact, aux.cm16, aux.unc, aux.uncm, aux.flags,
           Rcap, sh.B, sh.Q, sh.Qp, 0, 0, eps_rel, nullptr, 0};
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
               int64_t Rcap, float* exact, int64_t* pref, const FpMaxsimAux& aux, hipStream_t st, bool pref_ready) {
#define MS6_CASE(KS4_, NB_) \
  if (ix.dim == KS4_ * 32 && ix.nbits == NB_) { \
    if (!pref_ready) hipLaunchKernelGGL(k_cnt_prefix, dim3(1), dim3(256), 0, st, sel_cnt, sh.B, pref); \
    launch_maxsim6<KS4_, NB_>(ix, qpad, sh, sel_pid, pref, Rcap, exact, aux, st); \
    return 0; \
  }
#ifdef FP_MAXSIM_LAB_HOOK   // tools/probe/maxsim7_lab.hip (a variant library for experiments; never defined in the product build)
  if (fpk_maxsim_lab(ix, qpad, sh, sel_pid, sel_cnt, Rcap, exact, pref, aux, st, pref_ready) == 0) return 0;
#endif
  if (ix.resid_native) {
    MS6_CASE(4, 4) MS6_CASE(4, 2) MS6_CASE(4, 8) MS6_CASE(4, 1) MS6_CASE(2, 4) MS6_CASE(2, 2) MS6_CASE(3, 4) MS6_CASE(8, 4) MS6_CASE(8, 2)
  }
#undef MS6_CASE
#define MS_CASE(D_, NB_) \
  if (ix.dim == D_ && ix.nbits == NB_) { \
    if (!pref_ready) hipLaunchKernelGGL(k_cnt_prefix, dim3(1), dim3(256), 0, st, sel_cnt, sh.B, pref); \
    launch_maxsim5<D_, NB_>(ix, qpad, sh, sel_pid, pref, Rcap, exact, aux, st); \
    return 0; \
  }
  MS_CASE(96, 2) MS_CASE(48, 4) MS_CASE(48, 2)   // k_maxsim5: the fast shapes k_maxsim6 has no instantiation for (dim not a multiple of 32, 6-byte lane chunks)
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
                   hipStream_t st, uint32_t* flat_n, void* flat) {
  int np2 = fp_next_pow2((int)stride);
  if (np2 < 2) np2 = 2;
  if (np2 > 8192) return -1;   // 16 B of LDS per entry, 16 scan elements per thread
  static std::atomic<uint64_t> lds_ok{0};
  fp_allow_big_lds((const void*)k_final_mark, lds_ok, 136 * 1024);
  hipLaunchKernelGGL(k_final_mark, dim3((unsigned)B), dim3(1024), (size_t)np2 * 16, st, score, unc, uncm, cnt, stride, np2, top_k, marks, nmark,
                     flat_n, static_cast<uint2*>(flat));
  return 0;
}

// marks == nullptr: every slot with a budget > 0 is repaired
void fpk_maxsim_repair(const FpIndexDev& ix, const uint16_t* qpad, const FpSearchShape& sh, const int32_t* sel_pid, const int32_t* sel_cnt,
                       int64_t Rcap, const int32_t* marks, const int32_t* nmark, float* exact, const FpMaxsimAux& aux, hipStream_t st,
                       const uint32_t* flat_n, const void* flat) {
  if (!fpk_maxsim_fast_shape(ix.dim, ix.nbits) || !aux.unc) return;   // the generic kernel never flags
  // one wave per document; workgroups beyond the work list exit at once.  Batch-wide list: 2 x the average marked count of the
  // benchmark (~140 per query) in waves, the kernel strides if a batch has more.
  const dim3 grid = flat ? dim3((unsigned)std::min<int64_t>(320 * (int64_t)sh.B, std::min<int64_t>(Rcap * sh.B, 1 << 20)))
                         : dim3((unsigned)std::min<int64_t>(marks ? 192 : 4096, Rcap), (unsigned)sh.B);
#define MS_CASE(D_, NB_) \
  if (ix.dim == D_ && ix.nbits == NB_) { \
    hipLaunchKernelGGL((k_maxsim_repair<D_, NB_>), grid, dim3(64), 0, st, ix.centroids, ix.lut, ix.codes, ix.norms, ix.residuals, \
                       ix.doc_off, qpad, sh.Q, sh.Qp, sel_pid, sel_cnt, Rcap, marks, nmark, exact, aux.unc, aux.cm16, aux.flags, flat_n, \
                       static_cast<const uint2*>(flat)); \
    return; \
  }
  MS_CASE(128, 4) MS_CASE(128, 2) MS_CASE(128, 8) MS_CASE(128, 1) MS_CASE(96, 4) MS_CASE(96, 2) MS_CASE(64, 4) MS_CASE(64, 2) MS_CASE(48, 4) MS_CASE(48, 2)
  MS_CASE(256, 4) MS_CASE(256, 2)
#undef MS_CASE
}
