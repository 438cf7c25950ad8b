// HIP kernels of the PLAID search hot path for gfx950 (MI355X, CDNA4).  wave = 64.
//
// Stage map (reference rust/search/search.rs):
//   S1 k_centroid_scores*  :491        fp16 MFMA GEMM  S[b][c][q] = h(sum_k cent[c,k]*Q[b,q,k]); the epilogue also emits the column
//                                      maxima per 128 centroids (S2) and either level 0's excess byte per centroid or the 8-bit bins
//   S2 k_probe_*           :518-532    per query token top-n_probe centroids -> sorted unique cells
//   S3 k_ivf_mark/k_cand_* :535-547    IVF gather + sort + unique  == per-query doc bitmap + ordered compaction
//   S4                     :553-600    sum_q max_t S[code_t, q]   (fp16 max, fp32 sum), three forms with identical selections:
//        level 0  k_l0_scan (an upper bound of every candidate from its packed code list and a per-centroid byte table in LDS)
//                 -> k_l0_topcut / k_l0_pilot / k_l0_thr / k_l0_count / k_l0_compact -> k_approx on ~1.5 % of the candidates;
//        k_approx_q8 (8-bit bounds of every candidate; documents with many distinct codes) -> k_approx on the survivors;
//        k_approx (exact fp16 for every candidate: small candidate sets, traces)
//   S5 k_sel_*             :602-623    top-R by approx (radix select, ties -> lower doc id)
//   S6+S7 (fp_maxsim.hip)  :626-656    decompress (search.rs:53-107) fused with the exact MaxSim: k_maxsim6 / k_maxsim5 / k_maxsim_generic,
//                                      k_final_mark + k_maxsim_repair (exact-order repair of near-tied scores)
//   S8 k_final_topk        :658-692    sort by (score desc, doc id asc), truncate to top_k
// Beside the search path: k_token_scores (search.rs:668-686), k_reconstruct (embeddings.rs), k_assign_* / k_quantize_pack
// (index/create.rs:148-184, :404-428), shard helpers, the exhaustive arithmetic self-test.
//
// Numerical contract (measured on ATen 2.10 CPU, see oracle/plaid_oracle.c): every fp16
// tensor op = fp32 arithmetic + one round-to-nearest-even to fp16; matmuls accumulate in
// fp32 and round once.  The MFMA accumulation ORDER differs from the CPU's ascending-k
// chain, which is the only source of (<= 1 fp16 ulp, rare) differences.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fp_internal.h"

#include "fp_device.h"

// ============================================================================================
// query packing: [B,Q,D] -> [B*Qp, D] with zero rows for q >= Q
// ============================================================================================
__global__ void k_pack_queries(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int B, int Q, int Qp, int D, FpZeroList z,
                               float* __restrict__ wcol, float w0, uint16_t* __restrict__ out2, int D2) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per 8 halfs
  // the batch's small fills ride along (each hipMemsetAsync is a graph node of its own, ~4.5 us of launch tail)
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
#pragma unroll
  for (int r = 0; r < FP_ZERO_REGIONS; ++r)
    for (int64_t j = i; j < (int64_t)z.n16[r]; j += nthreads) reinterpret_cast<uint4*>(z.p[r])[j] = make_uint4(0, 0, 0, 0);
  // S1's certification window per packed row: w0 |q_n| (thread n < B * Qp walks row n; 0 for the zero rows appended to a query)
  if (wcol && i < (int64_t)B * Qp) {
    const int b = (int)(i / Qp), q = (int)(i % Qp);
    float ss = 0.f;
    if (q < Q) {
      const uint16_t* row = in + ((int64_t)b * Q + q) * D;
      for (int k = 0; k < D; k += 8) {
        const uint4 v = *reinterpret_cast<const uint4*>(row + k);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const h2 pr = u32_as_h2(w[t]);
          ss = __builtin_fmaf((float)pr.x, (float)pr.x, ss);
          ss = __builtin_fmaf((float)pr.y, (float)pr.y, ss);
        }
      }
    }
    wcol[i] = w0 * __builtin_sqrtf(ss);
  }
  // (one thread per 16-byte piece of the WIDER of the two output rows)
  const int Dw = out2 ? D2 : D;
  int64_t per_row = Dw / 8;
  int64_t total = (int64_t)B * Qp * per_row;
  if (i >= total) return;
  int64_t row = i / per_row;
  int c8 = (int)(i % per_row);
  int b = (int)(row / Qp), q = (int)(row % Qp);
  uint4 v = make_uint4(0, 0, 0, 0);
  if (q < Q && c8 * 8 < D) v = *reinterpret_cast<const uint4*>(in + ((int64_t)b * Q + q) * D + c8 * 8);
  if (c8 * 8 < D) *reinterpret_cast<uint4*>(out + row * D + c8 * 8) = v;
  if (out2) *reinterpret_cast<uint4*>(out2 + row * D2 + c8 * 8) = v;
}

void fpk_pack_queries(const uint16_t* in, uint16_t* out, int B, int Q, int Qp, int D, hipStream_t st, const FpZeroList* zero, float* wcol, float w0,
                      uint16_t* q_pad2, int D2) {
  int64_t total = (int64_t)B * Qp * ((q_pad2 ? D2 : D) / 8);
  FpZeroList z{};
  if (zero) z = *zero;
  hipLaunchKernelGGL(k_pack_queries, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, out, B, Q, Qp, D, z, wcol, w0, q_pad2, D2);
}

// ============================================================================================
// S1  centroid scores: 128(n = b*Qp+q) x 128(c) tile per 256-thread block, K = D in one go.
// MFMA rows = query columns n, MFMA cols = centroids c.  LDS operand tiles are XOR-swizzled at
// 16-byte granularity (row-major D-halfs rows would put a whole ds_read_b128 lane group on one
// bank slot).
// Epilogue: the fp16 tile is staged through LDS (the operand tiles are dead by then) so that the
// [b][c][q] layout is written with 16 B per lane, 1 KiB contiguous per wave instruction (the
// direct 8-byte stores reached 2 TB/s on the 537 MB of S), and two by-products leave with it:
//   * S8, the 8-bit bins of S4's bound stage (fp_kernels.hip, "bound-and-refine"), Qp <= 64 only;
//   * cmax, the per-column maximum over the tile's 128 centroids: the threshold probe's chunk
//     maxima at 128-centroid granularity for free.
// ============================================================================================
// byte code of an excess e for S4's level-0 table (see "S4 level 0" below):  0..159 = e;  160..239 = 160 + ceil((e - 160) / 8);
// 240..254 = escape slot 0..14 whose u32 value is kept in esc[slot] (esc[63] counts the slots handed out);  255 = infinite
#define L0_LIN 160
#define L0_ESC_BASE 240
#define L0_ESC_SLOTS 15
#define L0_EINF 0x100000u   // S1's epilogue: added to an excess sum for every clamped bin (the sums themselves stay below it)
__device__ __forceinline__ uint8_t l0_encode(uint32_t e, bool inf, uint32_t* __restrict__ esc) {
  if (inf) return 255u;
  if (e < L0_LIN) return (uint8_t)e;
  if (e <= L0_LIN + 8u * (L0_ESC_BASE - 1u - L0_LIN)) return (uint8_t)(L0_LIN + (e - L0_LIN + 7u) / 8u);
  const uint32_t slot = atomicAdd(&esc[63], 1u);
  if (slot >= L0_ESC_SLOTS) return 255u;
  esc[slot] = e;
  return (uint8_t)(L0_ESC_BASE + slot);
}
#define S1_TILE 128
// Epilogue of one 128 x 128 output tile (shared by the two S1 kernels): fp16 tile staged through LDS so that S leaves with
// 16 B per lane, plus the by-products (column maxima, 8-bit bins or the level-0 excess byte).
// query index and first column of the four 32-column groups of a 128-column tile (wave-uniform; Qp is a multiple of 32, so a
// group never straddles two queries): one division per tile instead of one per 16-byte piece
struct S1Groups { uint32_t bq[4]; uint32_t q0[4]; };
__device__ __forceinline__ S1Groups s1_groups(int64_t n0, int Qp) {
  S1Groups G;
  uint32_t bq = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)n0 / (uint32_t)Qp));   // (B * Qp < 2^31; the divide runs on the vector unit)
  uint32_t q = (uint32_t)n0 - bq * (uint32_t)Qp;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    G.bq[g] = (uint32_t)__builtin_amdgcn_readfirstlane((int)bq);   // (uniform already; this pins the values to scalar registers)
    G.q0[g] = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
    q += 32u;
    if (q >= (uint32_t)Qp) { q = 0u; ++bq; }
  }
  return G;
}
// ---- exact centroid scores: certification of the MFMA result and ascending-chain re-evaluation of the flagged entries ----------
// (FpS1Exact, fp_internal.h.)  A lane's 32 accumulators are 16 PAIRS of adjacent query columns; pair pj = (a * NB + b) * 8 + j holds
// acc[a][b][2 j], acc[a][b][2 j + 1].  Flag word: bit pj = the pair's even element, bit 16 + pj = its odd element.
#define S1X_CAP 256          // listed entries per wave and tile (12.5 % of its 2048 outputs; ~5 % are flagged on unit vectors)
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t s1_cvt_pk(float a, float b) {
  uint32_t r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// hw = the UPPER candidates h(x + u) of every entry (== h(x) wherever the entry is not flagged); returns the flag word.
// mode 2: *all_mask = every entry of a non-zero query column (tests: everything is re-evaluated)
template <int NA, int NB, typename WF>
__device__ __forceinline__ uint32_t s1_certify(const f16v (&acc)[NA][NB], uint32_t (&hw)[NA][NB][8], const int mode, const float kappa, WF wpair,
                                               uint32_t* all_mask) {
  static_assert(NA * NB * 8 <= 16, "one 32-bit flag word per lane");
  if (!mode) {   // (wave-uniform) plain rounding
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 8; ++j) hw[a][b][j] = s1_cvt_pk(acc[a][b][2 * j], acc[a][b][2 * j + 1]);
    *all_mask = 0u;
    return 0u;
  }
  const f2v kap = f2v{kappa, kappa};
  if (mode == 3) {   // (wave-uniform) lazy form: the upper candidates h(x + u) only -- no flags, no list, no chains (FpS1Exact)
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int j0 = 0; j0 < 8; j0 += 4) {
        f2v w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = wpair(a, j0 + j);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const f2v x = f2v{acc[a][b][2 * (j0 + jj)], acc[a][b][2 * (j0 + jj) + 1]};
            const f2v up = x + (__builtin_elementwise_abs(x) * kap + w[jj]);
            hw[a][b][j0 + jj] = s1_cvt_pk(up.x, up.y);
          }
      }
    *all_mask = 0u;
    return 0u;
  }
  uint32_t A = 0u, M = 0u;
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int j0 = 0; j0 < 8; j0 += 4) {   // four window pairs at a time (all sixteen up front cost registers the kernel does not have)
      f2v w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = wpair(a, j0 + j);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = j0 + jj;
        if (mode == 2) {   // (wave-uniform; tests only)
          const uint32_t nz = (w[jj].x > 0.f ? 1u : 0u) | (w[jj].y > 0.f ? 0x10000u : 0u);
#pragma unroll
          for (int b = 0; b < NB; ++b) M |= nz << ((a * NB + b) * 8 + j);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int pj = (a * NB + b) * 8 + j;
          const f2v x = f2v{acc[a][b][2 * j], acc[a][b][2 * j + 1]};
          const f2v u = __builtin_elementwise_abs(x) * kap + w[jj];
          const f2v up = x + u, lo = x - u;
          const uint32_t hu = s1_cvt_pk(up.x, up.y), hl = s1_cvt_pk(lo.x, lo.y);
          hw[a][b][j] = hu;
          uint32_t t;
          asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(hu ^ hl), "v"(0x00010001u));   // {even differs, odd differs} in bits 0 and 16
          A |= t << pj;
        }
      }
    }
  *all_mask = M;
  return A;
}
// accumulator index -> query column inside the wave's 32-column group (the 32x32 MFMA's D layout), element rr of lane half hi
__device__ __forceinline__ int s1_col_of(int rr, int hi) { return (rr & 3) + 8 * (rr >> 2) + 4 * hi; }
// eight steps of the reference's chain acc = fma(c_k, q_k, acc), k ascending, on packed fp16 pairs (k_maxsim_repair's spelling:
// v_fma_mix_f32 takes the halves in place; left to the compiler the pair becomes v_dot2, which rounds differently)
__device__ __forceinline__ void s1_chain8(float& acc, const uint4 c, const uint4 q) {
  const uint32_t cw[4] = {c.x, c.y, c.z, c.w}, qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int t = 0; t < 4; ++t)
    asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]\n\ts_nop 0\n\t"
                 "v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]\n\ts_nop 0"
                 : "+v"(acc)
                 : "v"(cw[t]), "v"(qw[t]));
}

// ---- the lazy form (FpS1Exact mode 3): what a STORED value s = h(x + u) says about the reference's value t = h(chain) ------------
// The chain result lies within [x - u, x + u] (the certification window), so t >= h(x - u); and s = h(x + u) puts x + u at or
// above the midpoint between s and its fp16 predecessor.  Hence t >= RNE16(mid - 2 u) =: s1_lower16(s) <= t <= s.  u is
// rebuilt from s (w + kappa |x|, |x| <= |s| (1 + 2^-10) + u: the factor 1.01 covers it), the fp32 subtraction is nudged down by
// more than its own rounding error.  Zero query columns (w == 0) hold exact zeros.  The map is monotone in s.
__device__ __forceinline__ uint16_t unmono16(uint32_t k) { return (k & 0x8000u) ? (uint16_t)(k ^ 0x8000u) : (uint16_t)~k; }
__device__ __forceinline__ float s1_u2(float s_abs, float w, float kappa) { return 2.f * (w + 1.01f * kappa * s_abs) + 1e-30f; }
__device__ __forceinline__ uint16_t s1_lower16(uint16_t stored, float w, float kappa) {
  if (w == 0.f) return stored;
  uint32_t k = mono16(stored);
  if (k == 0u) return stored;
  uint32_t kp = k - 1u;
  if (kp == 0x7FFFu) kp = 0x7FFEu;   // (key 0x7FFF is -0, which mono16 folds into +0)
  const float sf = (float)__builtin_bit_cast(half_t, stored), pf = (float)__builtin_bit_cast(half_t, unmono16(kp));
  const float mid = 0.5f * sf + 0.5f * pf;   // exact in fp32
  float lo = mid - s1_u2(__builtin_fabsf(sf), w, kappa);
  lo = lo - __builtin_fabsf(lo) * 2.4e-7f - 1e-37f;
  return __builtin_bit_cast(uint16_t, (half_t)lo);
}

// ---- epilogue of one 128 x 128 output tile, part 1: the fp16 tile (upper candidates) staged in LDS + the column maxima ----------
template <int NWC /*wave columns: 2 (256 threads, 64 x 64 per wave) or 4 (512 threads, 64 x 32 per wave)*/, int NWR = 2 /*wave rows: 2, or 4 (32 query columns per wave)*/>
__device__ __forceinline__ void s1_stage(const uint32_t (&hw)[4 / NWR][4 / NWC][8], unsigned char* smem, const int tid, const int64_t c0, const int64_t C,
                                         const bool want_cmax) {
  const int wave = tid >> 6, lane = tid & 63;
  constexpr int NB = 4 / NWC;          // 32-centroid MFMA tiles per wave
  constexpr int NA = 4 / NWR;          // 32-column query groups per wave
  const int wr = wave / NWC, wc = wave % NWC;
  const int l31 = lane & 31, hi = lane >> 5;
  __syncthreads();  // operand tiles are dead: the same LDS now stages the output
  // D[row = n (query col)][col = c]; lane: col = lane&31, rows (r&3)+8*(r>>2)+4*hi.
  // Output tile in LDS: [group g = n/32 (4)][c (128)][32 q] halves = 64 B rows, the 16-byte chunk index
  // XORed with (c>>2)&3 (lanes = consecutive c at a 64-byte stride would otherwise share banks).
  unsigned char* Os = smem;                                           // 32 KiB
  uint16_t* red = reinterpret_cast<uint16_t*>(smem + 4 * 128 * 64);   // [NWC][128 n] column maxima of each wave's centroids
  const uint32_t ninf2 = 0xFC00FC00u;  // packed fp16 -inf
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    const int g = wr * NA + a;
    uint32_t cm[8];  // packed column maxima: cm[2*q4 + h] = columns q = 8*q4 + 4*hi + 2*h + {0,1}
#pragma unroll
    for (int r = 0; r < 8; ++r) cm[r] = ninf2;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int cl = wc * (32 * NB) + b * 32 + l31;       // centroid within the tile
      const bool cok = (c0 + cl) < C;
      unsigned char* orow = Os + ((size_t)(g * 128 + cl)) * 64;
      const int f = (cl >> 2) & 3;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {             // q = 8*q4 + 4*hi + 0..3  -> chunk q4, half hi
        const uint32_t lo = hw[a][b][2 * q4], hi2 = hw[a][b][2 * q4 + 1];
        *reinterpret_cast<uint2*>(orow + ((q4 ^ f) * 16) + hi * 8) = make_uint2(lo, hi2);
        if (cok) {
          cm[2 * q4] = pk_max_raw(cm[2 * q4], lo);
          cm[2 * q4 + 1] = pk_max_raw(cm[2 * q4 + 1], hi2);
        }
      }
    }
    if (want_cmax) {  // max over this wave's 64 centroids: DPP inside the 16-lane rows, one cross-row exchange
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        uint32_t v = cm[r];
        v = pk_max_raw(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
        v = pk_max_raw(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
        v = pk_max_raw(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x124, 0xF, 0xF, false));   // row_ror:4
        v = pk_max_raw(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, false));   // row_ror:8
        v = pk_max_raw(v, shfl_xor_u32(v, 16));                                                   // the other row of this half
        cm[r] = v;
      }
      if (l31 == 0) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int q = 8 * (r >> 1) + 4 * hi + 2 * (r & 1);
          *reinterpret_cast<uint32_t*>(red + wc * 128 + g * 32 + q) = cm[r];
        }
      }
    }
  }
}
// LDS address of the staged fp16 value of (32-column group g, centroid cl of the tile, accumulator element rr of lane half hi)
__device__ __forceinline__ uint16_t* s1_staged_slot(unsigned char* Os, int g, int cl, int rr, int hi) {
  return reinterpret_cast<uint16_t*>(Os + ((size_t)(g * 128 + cl)) * 64 + (((rr >> 2) ^ ((cl >> 2) & 3)) * 16) + hi * 8 + (rr & 3) * 2);
}

// ---- part 2: write-out of the staged tile (16 B per lane) with the by-products (8-bit bins or the level-0 excess byte) ----------
template <int NWC, bool DMA_WAIT = false, int NWR = 2>
__device__ __forceinline__ void s1_writeout(unsigned char* smem, const int tid, const int64_t n0, const int64_t c0,
                                            const int tile_idx, uint16_t* __restrict__ S, const int64_t C, const int64_t Ntot, const int Qp,
                                            uint8_t* __restrict__ S8, uint16_t* __restrict__ cmax, const int nch, const FpS1Excess& ex, const S1Groups& G) {
  constexpr int NT = 64 * NWR * NWC;   // threads
  constexpr int NIT = 2048 / NT;       // 16-byte pieces of the output tile per thread
  unsigned char* Os = smem;                                           // 32 KiB
  uint16_t* red = reinterpret_cast<uint16_t*>(smem + 4 * 128 * 64);   // [NWC][128 n] column maxima of each wave's centroids
  __syncthreads();
  // k_centroid_scores_stream: the next tile's operand loads (straight into the other LDS buffer) were issued before this
  // tile's MFMAs; waiting for them here, in front of this tile's stores, keeps the wait from covering the stores' latency too
  if (DMA_WAIT) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) (a builtin, so that the compiler's own wait insertion sees it)
  // write-out: 2048 16-byte pieces (g, c, chunk), NIT per thread, consecutive threads -> consecutive bytes of S when Qp == 32.
  // Piece p = it * NT + tid: the group g and the first centroid row of the pass depend on `it` only (compile time), the
  // lane's row / chunk within the pass on tid only -- so every address below is a wave-uniform base (scalar registers) plus a
  // 32-bit per-lane offset that does not change with `it`.
  uint32_t eacc[NIT];   // ex.e8 != nullptr: excess of row (g, cl) over the column floors (+ L0_EINF per clamped bin), summed over the row's 4 lanes
  const int clt = tid >> 2, ch = tid & 3;
  const uint32_t lane_s = (uint32_t)(clt * Qp + ch * 8);                        // halves, within the pass's block of S
  const uint32_t lane_o = (uint32_t)(clt * 64 + ((ch ^ ((clt >> 2) & 3)) * 16));       // bytes, within the pass's block of the staged tile
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    eacc[it] = 0u;
    const int g = (it * NT) >> 9, cl0 = ((it * NT) >> 2) & 127;   // (cl0 is a multiple of 64: the staging swizzle (cl >> 2) & 3 sees clt only)
    const int64_t nb = n0 + g * 32;
    if (nb >= Ntot || c0 + cl0 + clt >= C) continue;
    const int64_t bq = G.bq[g];
    const int q0 = (int)G.q0[g];
    // (the floors are fetched before the store is issued: the memory counter is in order, so a load behind the store would make
    // its consumer wait for the store's acknowledgement)
    uint4 gw = make_uint4(0, 0, 0, 0);
    if (ex.e8) gw = *reinterpret_cast<const uint4*>(ex.gfl + (bq * Qp + q0) + (uint32_t)(ch * 8));
    const uint4 v = *reinterpret_cast<const uint4*>(Os + (g * 128 + cl0) * 64 + lane_o);
    uint16_t* Sblk = S + ((bq * C + c0 + cl0) * Qp + q0);   // (uniform)
    *reinterpret_cast<uint4*>(Sblk + lane_s) = v;
    if (ex.e8) {
      // S4 level 0's table entry, straight from the tile (instead of writing the 8-bit bins and reading them back):
      // e(c) = sum over the query's real columns of max(0, bin - floor_q); the floors come from a sampled pre-pass.
      // Packed fp16 throughout: 128 x, its floor, the clamp at bin 255 (floor = 155) and the difference to g = floor_q - 100
      // are small integers, exact in fp16; pad columns carry g = 2000 (never positive) and score 0 (never clamped).
      // (The floor has to be exact: rounding 128 x to nearest instead -- one packed fma with the constant 1280, still an upper
      // bound -- loosened the table enough to let 503 k instead of 391 k documents per batch through to the exact rescoring.)
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      const uint32_t gws[4] = {gw.x, gw.y, gw.z, gw.w};
      uint32_t part, inf;
      if (ex.rd) {
        // Round 6, the one-fma form: with the wave's fp16 rounding set toward -inf, fma(x, 128, 1024 - g) IS 1024 + floor(128 x - g)
        // whenever that is >= 1024 (fp16 steps by 1 in [1024, 2048); 128 x is exact, g an integer, one rounding), and anything
        // below 1024 is an excess of 0: y = max(fma, 1024) has the excess in its mantissa bits, 0x6400 + e.  Summed as 16-bit
        // integers from 0x7000 = -4 * 0x6400 (mod 2^16) the four biases cancel by themselves.  Three instructions per pair of
        // scores instead of eight (multiply, two floors, running maximum, clamp, subtract, max, add).  The clamp at bin 255 needs no
        // arithmetic: a clamped bin (128 x >= 155, i.e. x >= 1.2109375 = 0x3CD8, compared as integers -- positive halves order
        // like their bits, negative ones are negative) makes the centroid's entry infinite whatever the sum came to, and a sum
        // that left the [1024, 2048) window (x >= 7.2) is such a case.  Pad columns carry 1024 - 2000.
        uint32_t acc16 = 0x70007000u, tmax = 0x80008000u;
        f16_round_down();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint32_t y = pk_fma_mode_raw(w[t], 0x58005800u /*128, 128*/, gws[t]);
          acc16 = pk_add_u16(acc16, pk_max_raw(y, 0x64006400u /*1024, 1024*/));
          tmax = pk_max_i16(tmax, w[t]);
        }
        f16_round_nearest();
        part = (acc16 & 0xFFFFu) + (acc16 >> 16);
        inf = ((int)(short)(tmax & 0xFFFFu) >= 0x3CD8 || (int)(short)(tmax >> 16) >= 0x3CD8) ? L0_EINF : 0u;
      } else {
        const h2 k128 = {(half_t)128.f, (half_t)128.f}, k155 = {(half_t)155.f, (half_t)155.f}, kzero = {(half_t)0.f, (half_t)0.f};
        h2 dsum = kzero;
        uint32_t tmax = 0xE400E400u;   // packed -1024
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint32_t fl = pk_floor_raw(h2_as_u32(u32_as_h2(w[t]) * k128));
          tmax = pk_max_raw(tmax, fl);
          const h2 d = u32_as_h2(pk_min_raw(fl, h2_as_u32(k155))) - u32_as_h2(gws[t]);
          dsum += u32_as_h2(pk_max_raw(h2_as_u32(d), 0u));
        }
        // four integer terms of at most 255 per half: their total stays below 2048, exact in fp16
        part = (uint32_t)(int)(float)(half_t)(dsum.x + dsum.y);
        const h2 tm = u32_as_h2(tmax);
        inf = (tm.x >= (half_t)155.f || tm.y >= (half_t)155.f) ? L0_EINF : 0u;
      }
      uint32_t acc = part + inf;   // (the sums stay far below L0_EINF: at most 64 columns x 255; a garbage sum -- only next to a clamped bin -- below 2^17 per lane)
      acc += (uint32_t)__builtin_amdgcn_mov_dpp((int)acc, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
      acc += (uint32_t)__builtin_amdgcn_mov_dpp((int)acc, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
      eacc[it] = acc;
    }
    if (S8) {
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      uint32_t o[2] = {0u, 0u};
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const h2 pr = u32_as_h2(w[t >> 1]);
        const float x = (float)((t & 1) ? pr.y : pr.x);
        int bin = (int)floorf(x * 128.0f) + 100;   // == Q8_OFFSET, see the bound stage of S4
        bin = bin < 0 ? 0 : (bin > 255 ? 255 : bin);
        o[t >> 2] |= (uint32_t)bin << (8 * (t & 3));
      }
      // S8 is [b][Qp/32][c][32]: every 32-column chunk of a query is its own contiguous C x 32 B table
      uint8_t* S8blk = S8 + ((bq * (Qp / 32) + q0 / 32) * C + c0 + cl0) * 32;   // (uniform)
      *reinterpret_cast<uint2*>(S8blk + (uint32_t)(clt * 32 + ch * 8)) = make_uint2(o[0], o[1]);
    }
  }
  if (cmax && tid < 128) {
    const int64_t n = n0 + tid;
    if (n < Ntot) {
      uint16_t best = red[tid];
#pragma unroll
      for (int w = 1; w < NWC; ++w) {
        const uint16_t u = red[w * 128 + tid];
        best = (mono16(u) > mono16(best)) ? u : best;
      }
      cmax[n * nch + tile_idx] = best;
    }
  }
  if (ex.e8 && (tid & 3) == 0) {
    // one byte per (query, centroid): with Qp == 64 / 128 a query's two / four 32-column groups (pieces 512 apart) belong
    // together (Qp divides the tile's 128 columns, so a query's groups never straddle two tiles)
    const int per_q = Qp / 32;   // groups per query: 1, 2 or 4
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int g = (it * NT) >> 9, cl0 = ((it * NT) >> 2) & 127;
      if (g & (per_q - 1)) continue;
      const int64_t nb = n0 + g * 32;
      const int64_t c = c0 + cl0 + clt;
      if (nb >= Ntot || c >= C) continue;
      uint32_t acc = eacc[it];
#pragma unroll
      for (int j = 1; j < 4; ++j)
        if (j < per_q) acc += eacc[(it + j * (512 / NT)) % NIT];   // (index always < NIT here: g is a multiple of per_q)
      const int64_t bq = G.bq[g];
      ex.e8[bq * ex.Cpad + c] = l0_encode(acc & (L0_EINF - 1u), acc >= L0_EINF, ex.esc + bq * 64);
    }
  }
}

template <int KS, int NWC>
__global__ __launch_bounds__(128 * NWC) void k_centroid_scores(const uint16_t* __restrict__ cent, const uint16_t* __restrict__ qpad,
                                                         uint16_t* __restrict__ S, int64_t C, int64_t Ntot, int Qp,
                                                         uint8_t* __restrict__ S8 /*nullable*/, uint16_t* __restrict__ cmax /*nullable*/,
                                                         int nch, int D, int64_t crow_stride, FpS1Excess ex, int nrt_xcd, FpS1Exact xe) {
  // K is consumed in slices of KS dims: the operand tiles take 2 x 128 x KS x 2 B of LDS (32 KiB at KS = 64) instead of
  // 64 KiB for the whole K = 128, which lifts the kernel from 2 to 3 workgroups per CU (VGPR limit) so that one
  // workgroup's store phase overlaps another's MFMA phase.  Any dim that is a multiple of 8 (one 16-byte piece): KS = 64 /
  // 32 / 16 is the widest slice that divides dim, or 16 with the last slice's upper half zero-filled when dim % 16 == 8.
  constexpr int CH = KS / 8;             // 16-byte chunks per slice row
  constexpr int ROWB = KS * 2;           // bytes per slice row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Qs = smem;              // [128][ROWB]
  unsigned char* Cs = smem + 128 * ROWB; // [128][ROWB]
  const int tid = threadIdx.x;
  // Workgroups go round-robin over the 8 XCDs in launch order.  nrt_xcd > 0 (1-D grid): the nrt_xcd query tiles of one
  // centroid tile run back to back on ONE XCD, so the centroid tile crosses the fabric once (32 MB per batch at cfg2) instead
  // of once per query tile from the memory-side cache (512 MB); the 512 KB of queries stay in every XCD's L2 either way.
  int tile_c = (int)blockIdx.x, tile_n = (int)blockIdx.y;
  if (nrt_xcd > 0) {
    const uint32_t w = blockIdx.x, seq = w >> 3;
    tile_n = (int)(seq % (uint32_t)nrt_xcd);
    tile_c = (int)(seq / (uint32_t)nrt_xcd) * 8 + (int)(w & 7u);
    if (tile_c >= nch) return;   // (uniform)
  }
  const int64_t n0 = (int64_t)tile_n * 128;
  const int64_t c0 = (int64_t)tile_c * 128;
  const int wave = tid >> 6, lane = tid & 63;
  constexpr int NB = 4 / NWC;               // 32-centroid MFMA tiles per wave
  constexpr int NT = 128 * NWC;
  const int wr = wave / NWC, wc = wave % NWC;  // 2 x NWC waves, 64 (n) x 32 NB (c) each
  const int l31 = lane & 31, hi = lane >> 5;
  f16v acc[2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  for (int k0 = 0; k0 < D; k0 += KS) {
    if (k0 > 0) __syncthreads();   // the previous slice is no longer read
    // stage both tiles (coalesced 16 B per thread), swizzled
    for (int i = tid; i < 128 * CH; i += NT) {
      int row = i / CH, j = i % CH;
      int js = j ^ (row & (CH - 1));
      uint4 vq = make_uint4(0, 0, 0, 0), vc = make_uint4(0, 0, 0, 0);
      const bool kin = k0 + j * 8 < D;
      if (kin && n0 + row < Ntot) vq = *reinterpret_cast<const uint4*>(qpad + (n0 + row) * D + k0 + j * 8);
      if (kin && c0 + row < C) vc = *reinterpret_cast<const uint4*>(cent + (c0 + row) * crow_stride * D + k0 + j * 8);
      *reinterpret_cast<uint4*>(Qs + row * ROWB + js * 16) = vq;
      *reinterpret_cast<uint4*>(Cs + row * ROWB + js * 16) = vc;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < KS / 16; ++ks) {
      h8 af[2], bf[NB];
      const int j = ks * 2 + hi;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int rq = wr * 64 + t * 32 + l31;
        af[t] = *reinterpret_cast<const h8*>(Qs + rq * ROWB + ((j ^ (rq & (CH - 1))) * 16));
      }
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const int rc = wc * (32 * NB) + t * 32 + l31;
        bf[t] = *reinterpret_cast<const h8*>(Cs + rc * ROWB + ((j ^ (rc & (CH - 1))) * 16));
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
  }
  // certification of the accumulators (upper candidates staged), then the flagged entries through the ascending chain.  This
  // kernel holds only the last K slice in LDS, so both operand rows come from global memory (L2): every lane walks its OWN
  // flagged entries.  The fast path for the big tables is k_centroid_scores_stream.
  uint32_t hw[2][NB][8];
  uint32_t A = 0u, Aall = 0u;
  const int xmode = (NB == 1) ? xe.mode : 0;   // (the 64 x 64-per-wave layout has 64 accumulators per lane: exact mode runs the 4-column form)
  if constexpr (NB == 1) {
    A = s1_certify<2, NB>(acc, hw, xmode, xe.kappa, [&](int a, int j) -> f2v {
      const int64_t n = n0 + (wr * 2 + a) * 32 + s1_col_of(2 * j, hi);
      f2v w = f2v{0.f, 0.f};
      if (n + 1 < Ntot) { const float2 t = *reinterpret_cast<const float2*>(xe.wcol + n); w = f2v{t.x, t.y}; }
      else if (n < Ntot) w.x = xe.wcol[n];
      return w;
    }, &Aall);
  } else {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 8; ++j) hw[a][b][j] = s1_cvt_pk(acc[a][b][2 * j], acc[a][b][2 * j + 1]);
  }
  s1_stage<NWC>(hw, smem, tid, c0, C, cmax != nullptr);
  if (xmode == 1 || xmode == 2) {
    const uint32_t Aw = A;
    if (xmode == 2) A = Aall;
    if (c0 + wc * (32 * NB) + l31 >= C) A = 0u;   // rows beyond the table are never written
    uint32_t nflag = (uint32_t)__popc(A), nchg = 0u, nunf = 0u;
    uint32_t rem = A;
    while (rem) {
      const int p = __builtin_ctz(rem);
      rem &= rem - 1u;
      const int pj = p & 15, a = pj >> 3, rr = 2 * (pj & 7) + (p >> 4);
      const int cl = wc * (32 * NB) + l31;
      const int64_t n = n0 + (wr * 2 + a) * 32 + s1_col_of(rr, hi);
      const uint16_t* crow = cent + (c0 + cl) * crow_stride * D;
      const uint16_t* qrow = qpad + n * D;
      float ch = 0.f;
      for (int k = 0; k < D; k += 8) s1_chain8(ch, *reinterpret_cast<const uint4*>(crow + k), *reinterpret_cast<const uint4*>(qrow + k));
      const uint16_t val = __builtin_bit_cast(uint16_t, (half_t)ch);
      uint16_t* slot = s1_staged_slot(smem, wr * 2 + a, cl, rr, hi);
      if (*slot != val) { ++nchg; if (!((Aw >> p) & 1u)) ++nunf; }
      *slot = val;
    }
    if (xe.stats) {
      if (nflag) atomicAdd(&xe.stats[0], (unsigned long long)nflag);
      if (nchg) atomicAdd(&xe.stats[1], (unsigned long long)nchg);
      if (nunf) atomicAdd(&xe.stats[3], (unsigned long long)nunf);
    }
  }
  s1_writeout<NWC>(smem, tid, n0, c0, tile_c, S, C, Ntot, Qp, S8, cmax, nch, ex, s1_groups(n0, Qp));
}

// The streaming form of the same GEMM (dim 128 / 64, main pass): a workgroup of 8 waves (4 x 2: 32 query columns x 64 centroids
// each) keeps its query columns' MFMA A fragments in registers (32 VGPRs at dim 128) and walks `nct` consecutive 128-centroid
// tiles.  The centroid tile is the only operand that moves: `global_load_lds_dwordx4` writes it straight into one of two LDS
// buffers (no VGPR staging, so the loads of tile t+1 are in flight while tile t's MFMAs and epilogue run), swizzled on the
// global-address side (LDS piece p = row * CH + js receives the row's 16-byte piece js ^ (row % CH)).  The epilogue stages tile
// t's output in tile t's own operand buffer.  The loads are issued from inline assembly: the compiler orders every LDS access
// after an LDS-writing load it knows of with vmcnt(0), which would serialise the prefetch with the MFMA phase; here the one wait
// sits in the epilogue in front of the tile's stores and the barrier at the top of the loop publishes the buffer.
// (One-tile-per-workgroup form above: 64 KB of operands per tile in two serialised round trips, 201 us without its write-out.)
// one of the two LDS buffers: the operand tile (128 rows of DK halves; the staged output tile and, behind it at 32 KiB, the epilogue's
// column-maximum scratch reuse the space) -- 33 KiB up to dim 128, 65 KiB at dim 256 (one workgroup per CU then)
static constexpr int s1_buf_bytes(int dk) { return (128 * dk * 2 > 4 * 128 * 64 ? 128 * dk * 2 : 4 * 128 * 64) + 4 * 128 * 2; }
template <int DK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(DK > 128 ? 2 : 4, DK > 128 ? 2 : 4)))
void k_centroid_scores_stream(const uint16_t* __restrict__ cent, const uint16_t* __restrict__ qpad, uint16_t* __restrict__ S, int64_t C,
                              int64_t Ntot, int Qp, uint8_t* __restrict__ S8, uint16_t* __restrict__ cmax, int nch, int nct, int nrt,
                              FpS1Excess ex, FpS1Exact xe) {
  constexpr int CH = DK / 8;             // 16-byte chunks per row
  constexpr int ROWB = DK * 2;           // bytes per row
  constexpr int KSTEPS = DK / 16;
  constexpr int NLD = 128 * CH / 512;    // 16-byte pieces of a centroid tile per thread
  constexpr int S1_BUF = s1_buf_bytes(DK);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 1;  // 4 x 2 waves, 32 (n) x 64 (c) each (the tile loop derives its own copies from the laundered lane id)
  const int l31 = lane & 31, hi = lane >> 5;
  // (the divide runs on the vector unit: readfirstlane puts the wave-uniform results back into scalar registers)
  const int tile_n = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % (unsigned)nrt));
  const int t_first = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / (unsigned)nrt)) * nct;
  const int t_end = (t_first + nct < nch) ? t_first + nct : nch;
  if (t_first >= nch) return;   // (uniform)
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const uint32_t wave_s = (uint32_t)__builtin_amdgcn_readfirstlane(wave);
  // piece i of this lane: centroid row i * (512 / CH) + rbase of the tile, 16-byte chunk jcol of it, stored at the chunk
  // position jcol ^ (row % CH) of its LDS row (what the MFMA loop's reads undo)
  auto prefetch = [&](int tile, uint32_t buf_off) {
    int tidp = tid;
    asm volatile("" : "+v"(tidp));   // (per-piece offsets recomputed per tile instead of living in registers -- or scratch -- across the loop)
    const uint32_t rbase = (uint32_t)(tidp / CH);
    const uint32_t jcol = (uint32_t)(tidp % CH);
    const int64_t c0 = (int64_t)tile * 128;
    const uint16_t* tile_base = cent + c0 * DK;                        // (scalar)
    const uint32_t rows_valid = (uint32_t)(C - c0 < 128 ? C - c0 : 128);   // rows past the end read row 0 of the tile (never stored)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const uint32_t row = rbase + (uint32_t)(i * (512 / CH));
      const uint32_t j16 = (jcol ^ (row & (uint32_t)(CH - 1))) * 16u;   // (at dim <= 128 the row advances by a multiple of CH: the same for every i)
      const uint32_t voff = (row < rows_valid ? row : 0u) * (uint32_t)ROWB + j16;
      // lane l's 16 bytes land at LDS address M0 + 16 l
      const uint32_t m0v = lds0 + buf_off + ((uint32_t)i * 8u + wave_s) * 1024u;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // "m0 is reserved": nothing else in this kernel uses it
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(tile_base), "s"(m0v) : "memory", "m0");
#pragma clang diagnostic pop
    }
  };
  prefetch(t_first, 0u);
  h8 af[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const int64_t n = (int64_t)tile_n * 128 + wr * 32 + l31;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (n < Ntot) v = *reinterpret_cast<const uint4*>(qpad + n * DK + (ks * 2 + hi) * 8);
    af[ks] = __builtin_bit_cast(h8, v);
  }
  const S1Groups G0 = s1_groups((int64_t)tile_n * 128, Qp);
  // exact mode: the certification windows of the workgroup's 128 query columns (0 beyond the batch: those accumulators are exact
  // zeros) and a list of S1X_CAP flagged entries per wave, both behind the two tile buffers
  float* wtab = reinterpret_cast<float*>(smem + 2 * S1_BUF);
  if (xe.mode && tid < 128) {
    const int64_t n = (int64_t)tile_n * 128 + tid;
    wtab[tid] = n < Ntot ? xe.wcol[n] : 0.f;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the first tile and the A fragments (a builtin: the compiler's wait insertion then knows that nothing is pending at the loop head)
  for (int t = t_first; t < t_end; ++t) {
    const uint32_t cur_off = (uint32_t)((t - t_first) & 1) * (uint32_t)S1_BUF;
    unsigned char* cur = smem + cur_off;
    __syncthreads();   // every wave's pieces of tile t have landed; nobody still reads the other buffer (tile t-1's staging)
    if (t + 1 < t_end) prefetch(t + 1, (uint32_t)S1_BUF - cur_off);
    // (the lane id is laundered per tile: every per-lane LDS address below -- the sixteen swizzled operand chunks, the staging
    // rows, the write-out offsets -- is then recomputed from it inside the iteration instead of being hoisted out of the loop,
    // where the exact mode's chains leave no registers for them and they would be spilled: a scratch reload waits with vmcnt(0),
    // i.e. for the prefetch just issued)
    int tidv = tid;
    asm volatile("" : "+v"(tidv));
    const int lane = tidv & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tidv >> 6), wr = wave >> 1, wc = wave & 1;   // (scalar registers)
    uint16_t* lst = reinterpret_cast<uint16_t*>(smem + 2 * S1_BUF + 512) + wave * (2 * S1X_CAP);   // entries
    uint16_t* res = lst + S1X_CAP;                                                                                                       // ... and their values
    f16v acc[1][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][b][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int j = ks * 2 + hi;
      h8 bf[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int rc = wc * 64 + b * 32 + l31;
        bf[b] = *reinterpret_cast<const h8*>(cur + rc * ROWB + ((j ^ (rc & (CH - 1))) * 16));
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks], bf[b], acc[0][b], 0, 0, 0);
    }
    // (tile_n is laundered so that the epilogue's per-lane addresses are recomputed per tile instead of living in registers
    // across the loop: with them hoisted the kernel spills, and a scratch reload's vmcnt(0) would wait for the prefetch)
    int tn = tile_n;
    S1Groups G = G0;
    asm volatile("" : "+s"(tn));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      asm volatile("" : "+s"(G.bq[g]));
      asm volatile("" : "+s"(G.q0[g]));
    }
    // ---- certification + exact re-evaluation of the flagged entries (FpS1Exact).  The centroid tile is still in LDS and the
    // query rows are in the A fragments: lane (n, hi) holds the chunks 2 ks + hi of column n, so a lane that re-evaluates column
    // n' fetches them with ds_bpermute from lanes n' and n' + 32.  The wave first lists its flagged entries (round-robin over the
    // lanes' flag words), then lane i re-evaluates list entry 64 p + i: ~6.6 % of the 2048 outputs, two to three passes.
    uint32_t hw[1][2][8];
    uint32_t Aall = 0u;
    uint32_t A = s1_certify<1, 2>(acc, hw, xe.mode, xe.kappa, [&](int, int j) -> f2v {
      const float2 t2 = *reinterpret_cast<const float2*>(wtab + wr * 32 + s1_col_of(2 * j, hi));
      return f2v{t2.x, t2.y};
    }, &Aall);
    const uint32_t Aw = A;
    if (xe.mode == 2) A = Aall;
    uint32_t listed = 0u, flagged_total = 0u, lbase = 0u;
    const bool chains = xe.mode == 1 || xe.mode == 2;   // (mode 3, the lazy form: nothing is re-evaluated here)
    if (chains) {
      // list position of a lane's first entry = exclusive prefix of the lanes' counts (<= 32 each): one ballot per count bit
      const uint32_t cnt = (uint32_t)__popc(A);
      uint32_t off = 0u;
      lbase = 0u;
#pragma unroll
      for (int bit = 0; bit < 6; ++bit) {
        const unsigned long long bal = __ballot(((cnt >> bit) & 1u) != 0u);
        lbase += __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u)) << bit;
        off += (uint32_t)__popcll(bal) << bit;
      }
      for (uint32_t rem = A, slot = lbase; rem != 0u; rem &= rem - 1u, ++slot)
        if (slot < S1X_CAP) lst[slot] = (uint16_t)(((uint32_t)lane << 5) | (uint32_t)__builtin_ctz(rem));
      flagged_total = off;
      listed = off < S1X_CAP ? off : S1X_CAP;   // (wave-uniform)
      // One pass = 64 list entries, one per lane; while more than 64 remain a pass takes 128, TWO per lane, whose chains are
      // interleaved (the wait state a dependent v_fma_mix needs is then filled by the other chain instead of an s_nop, and the
      // loop's bookkeeping is shared): ~5 % flagged = ~107 entries per wave and tile = one double pass.
      struct Ent { const unsigned char* crow; int swz, ncol; };
      auto decode = [&](uint32_t e) -> Ent {
        const int ls = (int)(e >> 5), p = (int)(e & 31u), pj = p & 15, rr = 2 * (pj & 7) + (p >> 4);
        const int cl = wc * 64 + (pj >> 3) * 32 + (ls & 31);
        return Ent{cur + cl * ROWB, cl & (CH - 1), s1_col_of(rr, ls >> 5)};
      };
      auto fetch_q = [&](const Ent& en, int jj) -> uint4 {
        const uint4 av = __builtin_bit_cast(uint4, af[jj >> 1]);
        const int src = (en.ncol + 32 * (jj & 1)) * 4;
        uint4 qv;
        qv.x = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)av.x);
        qv.y = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)av.y);
        qv.z = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)av.z);
        qv.w = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)av.w);
        return qv;
      };
      auto fetch_c = [&](const Ent& en, int jj) -> uint4 { return *reinterpret_cast<const uint4*>(en.crow + ((jj ^ en.swz) * 16)); };
      uint32_t p0 = 0;
      for (; p0 + 64 < listed; p0 += 128) {   // double passes
        const uint32_t i0 = p0 + (uint32_t)lane, i1 = i0 + 64u;
        const bool v1 = i1 < listed;
        const Ent e0 = decode((uint32_t)lst[i0]), e1 = decode(v1 ? (uint32_t)lst[i1] : 0u);
        float ch0 = 0.f, ch1 = 0.f;
        // (operands of chunk jj + 1 are fetched while chunk jj is chained; the empty asm keeps the compiler from hoisting the
        // fetches of all sixteen chunks to the top, which spills)
        uint4 c0v = fetch_c(e0, 0), q0v = fetch_q(e0, 0), c1v = fetch_c(e1, 0), q1v = fetch_q(e1, 0);
#pragma unroll
        for (int jj = 0; jj < CH; ++jj) {
          uint4 c0n = c0v, q0n = q0v, c1n = c1v, q1n = q1v;
          if (jj + 1 < CH) {
            c0n = fetch_c(e0, jj + 1); q0n = fetch_q(e0, jj + 1);
            c1n = fetch_c(e1, jj + 1); q1n = fetch_q(e1, jj + 1);
          }
          asm volatile("" ::: "memory");
          const uint32_t a0[4] = {c0v.x, c0v.y, c0v.z, c0v.w}, b0[4] = {q0v.x, q0v.y, q0v.z, q0v.w};
          const uint32_t a1[4] = {c1v.x, c1v.y, c1v.z, c1v.w}, b1[4] = {q1v.x, q1v.y, q1v.z, q1v.w};
#pragma unroll
          for (int u = 0; u < 4; ++u)
            asm volatile("v_fma_mix_f32 %0, %2, %3, %0 op_sel_hi:[1,1,0]\n\t"
                         "v_fma_mix_f32 %1, %4, %5, %1 op_sel_hi:[1,1,0]\n\t"
                         "v_fma_mix_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]\n\t"
                         "v_fma_mix_f32 %1, %4, %5, %1 op_sel:[1,1,0] op_sel_hi:[1,1,0]"
                         : "+v"(ch0), "+v"(ch1)
                         : "v"(a0[u]), "v"(b0[u]), "v"(a1[u]), "v"(b1[u]));
          c0v = c0n; q0v = q0n; c1v = c1n; q1v = q1n;
        }
        res[i0] = __builtin_bit_cast(uint16_t, (half_t)ch0);
        if (v1) res[i1] = __builtin_bit_cast(uint16_t, (half_t)ch1);
      }
      for (; p0 < listed; p0 += 64) {   // the last (or only) 64
        const uint32_t idx = p0 + (uint32_t)lane;
        const bool valid = idx < listed;
        const Ent en = decode(valid ? (uint32_t)lst[idx] : 0u);
        float chn = 0.f;
        uint4 cv = fetch_c(en, 0), qv = fetch_q(en, 0);
#pragma unroll
        for (int jj = 0; jj < CH; ++jj) {
          uint4 cvn = cv, qvn = qv;
          if (jj + 1 < CH) {
            cvn = fetch_c(en, jj + 1);
            qvn = fetch_q(en, jj + 1);
          }
          asm volatile("" ::: "memory");
          s1_chain8(chn, cv, qv);
          cv = cvn;
          qv = qvn;
        }
        if (valid) res[idx] = __builtin_bit_cast(uint16_t, (half_t)chn);
      }
    }
    s1_stage<2, 4>(hw, cur, tidv, (int64_t)t * 128, C, cmax != nullptr);
    if (chains && flagged_total <= S1X_CAP && !xe.stats) {
      // the re-evaluated values go over the staged upper candidates: the lane that ran an entry's chain writes it (the staging
      // stores of the whole wave are older in its LDS queue)
      for (uint32_t p0 = 0; p0 < listed; p0 += 64) {
        const uint32_t idx = p0 + (uint32_t)lane;
        if (idx < listed) {
          const uint32_t e = (uint32_t)lst[idx];
          const int ls = (int)(e >> 5), p = (int)(e & 31u), pj = p & 15, rr = 2 * (pj & 7) + (p >> 4);
          *s1_staged_slot(cur, wr, wc * 64 + (pj >> 3) * 32 + (ls & 31), rr, ls >> 5) = res[idx];
        }
      }
    } else if (chains) {
      // the general form (a wave with more flagged entries than its list holds -- never seen on unit vectors --, or the counters
      // are wanted): every lane walks ITS entries (list positions lbase, lbase + 1, ...); entries beyond the list are re-evaluated
      // here with the centroid row from global memory -- the tile's LDS copy is gone.
      uint32_t rem = A, slot = lbase, nchg = 0u, nunf = 0u, nslow = 0u;
      for (;; ++slot) {
        const bool act = rem != 0u;
        if (!__any(act)) break;
        int p = 0;
        if (act) { p = __builtin_ctz(rem); rem &= rem - 1u; }
        const int pj = p & 15, rr = 2 * (pj & 7) + (p >> 4);
        const int cl = wc * 64 + (pj >> 3) * 32 + l31;
        uint16_t val = 0;
        const bool slow = act && slot >= S1X_CAP;
        if (__any(slow)) {
          const int ncol = s1_col_of(rr, hi);
          const int64_t crow_i = (int64_t)t * 128 + cl;
          const uint16_t* crow = cent + (crow_i < C ? crow_i : 0) * DK;
          float chn = 0.f;
#pragma unroll
          for (int jj = 0; jj < CH; ++jj) {
            const uint4 cv = *reinterpret_cast<const uint4*>(crow + jj * 8);
            const uint4 av = __builtin_bit_cast(uint4, af[jj >> 1]);
            const int src = (ncol + 32 * (jj & 1)) * 4;
            uint4 qv;
            qv.x = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)av.x);
            qv.y = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)av.y);
            qv.z = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)av.z);
            qv.w = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)av.w);
            asm volatile("" ::: "memory");   // (rare path: one chunk at a time)
            s1_chain8(chn, cv, qv);
          }
          val = __builtin_bit_cast(uint16_t, (half_t)chn);
          if (slow) ++nslow;
        }
        if (act) {
          if (slot < S1X_CAP) val = res[slot];
          uint16_t* sp = s1_staged_slot(cur, wr, cl, rr, hi);
          if (*sp != val) { ++nchg; if (!((Aw >> p) & 1u)) ++nunf; }
          *sp = val;
        }
      }
      if (xe.stats) {
        const uint32_t nflag = (uint32_t)__popc(A);
        if (nflag) atomicAdd(&xe.stats[0], (unsigned long long)nflag);
        if (nchg) atomicAdd(&xe.stats[1], (unsigned long long)nchg);
        if (nslow) atomicAdd(&xe.stats[2], (unsigned long long)nslow);
        if (nunf) atomicAdd(&xe.stats[3], (unsigned long long)nunf);
      }
    }
    s1_writeout<2, true, 4>(cur, tidv, (int64_t)tn * 128, (int64_t)t * 128, t, S, C, Ntot, Qp, S8, cmax, nch, ex, G);
  }
}

int fpk_centroid_scores(const FpIndexDev& ix, const uint16_t* qpad, uint16_t* S, int B, int Qp, uint8_t* S8, uint16_t* cmax,
                        hipStream_t st, int64_t n_rows, int64_t row_stride, const FpS1Excess* exc, const FpS1Exact* exact) {
  // n_rows / row_stride: score only the centroids 0, row_stride, 2 row_stride, ... (n_rows of them; S / S8 are then n_rows tall)
  const int64_t Ntot = (int64_t)B * Qp;
  const int64_t C = n_rows > 0 ? n_rows : ix.C;
  const int64_t rs = n_rows > 0 ? row_stride : 1;
  FpS1Excess ex{};
  if (exc) ex = *exc;
  FpS1Exact xe{};
  if (exact && exact->wcol) xe = *exact;
  const int nch = (int)((C + S1_TILE - 1) / S1_TILE);
  dim3 grid((unsigned)nch, (unsigned)((Ntot + 127) / 128));
  if (Qp != 32 && Qp != 64 && Qp != 128) S8 = nullptr;   // the bound stages handle one, two or (level 0 only) four 32-column chunks
  const size_t out_lds = 4 * 128 * 64 + 4 * 128 * 2;
  const int D = ix.dim;
  if (D % 8 != 0 || D < 8) return -1;
  // streaming form: main pass only (the sampled pre-pass walks a strided table), dim 64 / 128 / 256.  Tiles per workgroup: 8 for
  // the big batches (cfg2: 16384 tiles), fewer while that would leave CUs without a workgroup -- the kernel is used for small
  // batches too since round 4, because its exact mode repairs from the LDS tile where the one-tile kernel fetches both rows of
  // every flagged score from L2 (cfg2's table, S1 stage: B = 1 79 -> 52 us, B = 4 143 -> 71 us, B = 8 120 -> 99 us)
  static const int stream_env = [] { const char* e = getenv("FP_S1_STREAM"); return e ? atoi(e) : 1; }();   // 0: one tile per workgroup; n > 1: tiles per workgroup
  if ((D == 128 || D == 64 || D == 256) && stream_env && n_rows <= 0) {
    const int nrt = (int)grid.y;
    const int64_t tiles = (int64_t)nch * grid.y;
    const int nct = stream_env > 1 ? stream_env : (int)std::min<int64_t>(8, std::max<int64_t>(1, tiles / 512));
    const unsigned nwg = (unsigned)((nch + nct - 1) / nct) * (unsigned)nrt;
    const size_t lds = 2 * (size_t)s1_buf_bytes(D) + 512 + 8 * 2 * S1X_CAP * 2;   // 74.5 KiB (tile buffers + windows + the waves' entry / value lists; 138.5 KiB at dim 256): above the 64 KiB that need no opt-in
    static std::atomic<uint64_t> ok128{0}, ok64{0}, ok256{0};
    if (D == 128) fp_allow_big_lds((const void*)k_centroid_scores_stream<128>, ok128, 80 * 1024);
    else if (D == 256) fp_allow_big_lds((const void*)k_centroid_scores_stream<256>, ok256, 144 * 1024);
    else fp_allow_big_lds((const void*)k_centroid_scores_stream<64>, ok64, 80 * 1024);
    if (D == 256)
      hipLaunchKernelGGL(k_centroid_scores_stream<256>, dim3(nwg), dim3(512), lds, st, ix.centroids, qpad, S, C, Ntot, Qp, S8, cmax, nch, nct, nrt, ex, xe);
    else if (D == 128)
      hipLaunchKernelGGL(k_centroid_scores_stream<128>, dim3(nwg), dim3(512), lds, st, ix.centroids, qpad, S, C, Ntot, Qp, S8, cmax, nch, nct, nrt, ex, xe);
    else
      hipLaunchKernelGGL(k_centroid_scores_stream<64>, dim3(nwg), dim3(512), lds, st, ix.centroids, qpad, S, C, Ntot, Qp, S8, cmax, nch, nct, nrt, ex, xe);
    return 0;
  }
  const int nwc = 4;   // eight waves (4 x 2); the 2 x 2-wave layout of round 1 and the XCD-aware tile order (315 vs 297 us) are gone
  const int nrt = 0;
  if (D % 64 == 0) {
    const size_t lds = std::max<size_t>(2 * 128 * 64 * 2, out_lds);
    if (nwc == 4) hipLaunchKernelGGL((k_centroid_scores<64, 4>), grid, dim3(512), lds, st, ix.centroids, qpad, S, C, Ntot, Qp, S8, cmax, nch, D, rs, ex, nrt, xe);
    else hipLaunchKernelGGL((k_centroid_scores<64, 2>), grid, dim3(256), lds, st, ix.centroids, qpad, S, C, Ntot, Qp, S8, cmax, nch, D, rs, ex, nrt, xe);
  } else if (D % 32 == 0) {
    const size_t lds = std::max<size_t>(2 * 128 * 32 * 2, out_lds);
    if (nwc == 4) hipLaunchKernelGGL((k_centroid_scores<32, 4>), grid, dim3(512), lds, st, ix.centroids, qpad, S, C, Ntot, Qp, S8, cmax, nch, D, rs, ex, nrt, xe);
    else hipLaunchKernelGGL((k_centroid_scores<32, 2>), grid, dim3(256), lds, st, ix.centroids, qpad, S, C, Ntot, Qp, S8, cmax, nch, D, rs, ex, nrt, xe);
  } else {
    const size_t lds = std::max<size_t>(2 * 128 * 16 * 2, out_lds);
    if (nwc == 4) hipLaunchKernelGGL((k_centroid_scores<16, 4>), grid, dim3(512), lds, st, ix.centroids, qpad, S, C, Ntot, Qp, S8, cmax, nch, D, rs, ex, nrt, xe);
    else hipLaunchKernelGGL((k_centroid_scores<16, 2>), grid, dim3(256), lds, st, ix.centroids, qpad, S, C, Ntot, Qp, S8, cmax, nch, D, rs, ex, nrt, xe);
  }
  return 0;
}

// ============================================================================================
// S2  probe selection.  key = mono16(score) << 32 | (0xFFFFFFFF - c): larger key = higher
// score, ties -> lower centroid id.  key 0 = "none".
// ============================================================================================
template <int NP>
__global__ __launch_bounds__(256) void k_probe_partial(const uint16_t* __restrict__ S, int64_t C, int Q, int Qp, int nchunk,
                                                        const uint32_t* __restrict__ allow, int64_t Cw,
                                                        unsigned long long* __restrict__ partial, const int32_t* __restrict__ flag) {
  // grid: x = chunk, y = b * (Qp/32) + colgroup.  thread: col = tid&31, slice = tid>>5 (8 slices)
  if (flag && !*flag) return;  // fallback path of the threshold probe
  const int groups = Qp / 32;
  const int b = blockIdx.y / groups, g = blockIdx.y % groups;
  const int col = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const int q = g * 32 + col;
  const int chunk = blockIdx.x;
  const int64_t per = ((C + nchunk - 1) / nchunk + 7) & ~(int64_t)7;
  const int64_t cb = chunk * per;
  const int64_t ce = (cb + per < C) ? cb + per : C;
  unsigned long long best[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) best[i] = 0ull;
  if (q < Q) {
    const uint16_t* Sb = S + (int64_t)b * C * Qp + q;
    const uint32_t* al = allow ? allow + (int64_t)b * Cw : nullptr;
    for (int64_t c = cb + slice; c < ce; c += 8) {
      if (al && !((al[c >> 5] >> (c & 31)) & 1u)) continue;
      uint16_t s = Sb[c * Qp];
      unsigned long long key = ((unsigned long long)mono16(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)c);
      if (key > best[NP - 1]) {
        best[NP - 1] = key;
#pragma unroll
        for (int i = NP - 1; i > 0; --i) {
          if (best[i] > best[i - 1]) {
            unsigned long long t = best[i];
            best[i] = best[i - 1];
            best[i - 1] = t;
          }
        }
      }
    }
  }
  unsigned long long* dst = partial + ((((int64_t)b * Qp + q) * nchunk + chunk) * 8 + slice) * NP;
#pragma unroll
  for (int i = 0; i < NP; ++i) dst[i] = best[i];
}

// one wave per (b, q) column: n_probe rounds of "largest key below the previous pick"
__global__ __launch_bounds__(64) void k_probe_merge(const unsigned long long* __restrict__ partial, int Q, int Qp, int entries_stride,
                                                    int n_probe, int32_t* __restrict__ cells, const int32_t* __restrict__ flag,
                                                    int run_if, const uint32_t* __restrict__ cnt /*nullable: valid entries per column*/,
                                                    const uint16_t* __restrict__ cent /*nullable: S1's lazy form, see below*/,
                                                    const uint16_t* __restrict__ qpad, int D,
                                                    const float* __restrict__ wcol /*nullable: the columns' certification windows (0 = an all-zero query row)*/) {
  int entries = entries_stride;
  const int b = blockIdx.x / Q, q = blockIdx.x % Q;
  const int lane = threadIdx.x;
  const bool zero_row = wcol && !(wcol[(int64_t)b * Qp + q] > 0.f);
  // (S1's lazy form: a lane's collected entry is fetched together with the flag and the count, not behind them)
  const unsigned long long k0_pre = (cent && lane < entries_stride) ? partial[((int64_t)b * Qp + q) * entries_stride + lane] : 0ull;
  if (flag && ((*flag != 0) != (run_if != 0))) {
    // the threshold path gave up (a column overflowed its list): no cells, unless the fallback that follows writes them
    if (run_if == 0)
      for (int r = lane; r < n_probe; r += 64) cells[((int64_t)b * Q + q) * n_probe + r] = -1;
    return;
  }
  if (zero_row) {   // every centroid ties at 0: (score desc, centroid id asc) = the lowest-numbered ones (k_probe_tau collected nothing)
    for (int r = lane; r < n_probe; r += 64) cells[((int64_t)b * Q + q) * n_probe + r] = r;
    return;
  }
  if (cnt) {
    const int have = (int)cnt[(int64_t)b * Qp + q];
    entries = have < entries ? have : entries;
  }
  const unsigned long long* src = partial + ((int64_t)b * Qp + q) * entries_stride;
  unsigned long long bound = ~0ull;
  if (cent) {
    // S1's lazy form: the collected scores are upper candidates h(x + u).  Every one of them (<= PROBE_CAP = 64: one per lane) is
    // re-evaluated HERE with the reference's ascending chain, so that the top-n_probe below is taken over the reference's values
    // (the collect pass took everything that can reach the true cut: k_probe_tau lowered its threshold by the window).
    unsigned long long mine = 0ull;
    if (lane < entries) {
      const unsigned long long k0 = k0_pre;
      const uint32_t cix = 0xFFFFFFFFu - (uint32_t)k0;
      const uint16_t* crow = cent + (int64_t)cix * D;
      const uint16_t* qrow = qpad + ((int64_t)b * Qp + q) * D;
      float ch = 0.f;
#pragma unroll 1
      for (int k0d = 0; k0d < D; k0d += 64) {   // 64 dims of both rows in flight (a load pair per chain step was 16 dependent round trips)
        uint4 cv[8], qv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool in = k0d + j * 8 < D;
          cv[j] = in ? *reinterpret_cast<const uint4*>(crow + k0d + j * 8) : make_uint4(0, 0, 0, 0);
          qv[j] = in ? *reinterpret_cast<const uint4*>(qrow + k0d + j * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (k0d + j * 8 < D) s1_chain8(ch, cv[j], qv[j]);
      }
      mine = ((unsigned long long)mono16(__builtin_bit_cast(uint16_t, (half_t)ch)) << 32) | (k0 & 0xFFFFFFFFull);
    }
    for (int r = 0; r < n_probe; ++r) {
      unsigned long long m = (mine < bound) ? mine : 0ull;
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) {
        unsigned long long o = __shfl_xor(m, s, 64);
        m = o > m ? o : m;
      }
      if (lane == 0) cells[((int64_t)b * Q + q) * n_probe + r] = m ? (int32_t)(0xFFFFFFFFu - (uint32_t)m) : -1;
      if (m == 0ull) {
        for (int r2 = r + 1; r2 < n_probe && lane == 0; ++r2) cells[((int64_t)b * Q + q) * n_probe + r2] = -1;
        break;
      }
      bound = m;
    }
    return;
  }
  for (int r = 0; r < n_probe; ++r) {
    unsigned long long m = 0ull;
    for (int i = lane; i < entries; i += 64) {
      unsigned long long k = src[i];
      if (k < bound && k > m) m = k;
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      unsigned long long o = __shfl_xor(m, s, 64);
      m = o > m ? o : m;
    }
    if (lane == 0) cells[((int64_t)b * Q + q) * n_probe + r] = m ? (int32_t)(0xFFFFFFFFu - (uint32_t)m) : -1;
    if (m == 0ull) {
      for (int r2 = r + 1; r2 < n_probe && lane == 0; ++r2) cells[((int64_t)b * Q + q) * n_probe + r2] = -1;
      break;
    }
    bound = m;
  }
}

// ---- any n_probe: exact radix select per column (generic path: n_probe > 32, or the overflow fallback of those) -------------
// One workgroup per (query, 32-column group).  Two histogram passes over the monotone 16-bit keys give the n_probe-th largest key
// of the column and how many centroids at that key are still needed; the collect pass takes everything above it and, among the
// equal ones, the lowest centroid ids (threads own contiguous centroid ranges, so a prefix over the ranges' equal counts orders
// them).  Four reads of S: slower than the threshold probe, any n_probe <= C.
// Round 6: thread = (16-byte piece of the group's 64-byte row, one of 256 contiguous centroid ranges) on 1024 threads -- a wave
// reads 16 WHOLE rows per load, four rows ahead of their histogram updates.  (The first form, thread = (column, one of 8 ranges)
// on 256 threads, read two bytes per lane with a 64-byte stride and waited for every load: 12 ms per cfg2 batch at n_ivf_probe
// 64, twice that through the replayed graph -- see the engine's note on Pipe::probe_no_fb.)
__global__ __launch_bounds__(1024) void k_probe_select(const uint16_t* __restrict__ S, int64_t C, int Q, int Qp, int n_probe,
                                                       const uint32_t* __restrict__ allow, int64_t Cw, int32_t* __restrict__ cells,
                                                       const int32_t* __restrict__ flag) {
  if (flag && !*flag) return;
  constexpr int NSL = 256;
  __shared__ uint32_t hist[32][257];
  __shared__ uint32_t thr_key[32], need_eq[32], out_pos[32];
  __shared__ uint16_t cnt_eq[32][NSL];   // (a range holds C / 256 <= 65535 centroids up to C = 2^24)
  const int groups = Qp / 32;
  const int b = blockIdx.x / groups, g = blockIdx.x % groups;
  const int tid = threadIdx.x, p = tid & 3, slice = tid >> 2;
  uint32_t livem = 0u;   // the group's real columns
  for (int j = 0; j < 32; ++j) livem |= (g * 32 + j < Q) ? (1u << j) : 0u;
  const int64_t per = (C + NSL - 1) / NSL;
  const int64_t cb = slice * per, ce = (cb + per < C) ? cb + per : C;
  const uint16_t* Sb = S + (int64_t)b * C * Qp + g * 32 + p * 8;
  const uint32_t* al = allow ? allow + (int64_t)b * Cw : nullptr;
  auto allowed = [&](int64_t c) -> bool { return !al || ((al[c >> 5] >> (c & 31)) & 1u); };
  auto key8 = [&](const uint4& v, int j) -> uint32_t {
    const uint32_t w = j < 2 ? v.x : (j < 4 ? v.y : (j < 6 ? v.z : v.w));
    return mono16((uint16_t)((j & 1) ? (w >> 16) : (w & 0xFFFFu)));
  };
  // visits the range's allowed rows four loads at a time: f(c, row)
  auto walk = [&](auto&& f) {
    for (int64_t c0 = cb; c0 < ce; c0 += 4) {
      uint4 v[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ok[u] = c0 + u < ce && allowed(c0 + u);
        v[u] = make_uint4(0, 0, 0, 0);
        if (ok[u]) v[u] = *reinterpret_cast<const uint4*>(Sb + (c0 + u) * Qp);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ok[u]) f(c0 + u, v[u]);
    }
  };
  uint32_t prefix[8], rem[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { prefix[j] = 0u; rem[j] = (uint32_t)n_probe; }
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = tid; i < 32 * 257; i += 1024) (&hist[0][0])[i] = 0u;
    __syncthreads();
    walk([&](int64_t, const uint4& v) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (!((livem >> (p * 8 + j)) & 1u)) continue;
        const uint32_t k = key8(v, j);
        if (pass == 0) atomicAdd(&hist[p * 8 + j][k >> 8], 1u);
        else if ((k >> 8) == prefix[j]) atomicAdd(&hist[p * 8 + j][k & 0xFFu], 1u);
      }
    });
    __syncthreads();
    if (tid < 32 && ((livem >> tid) & 1u)) {
      const uint32_t pf = pass == 1 ? thr_key[tid] : 0u, rm = pass == 1 ? need_eq[tid] : (uint32_t)n_probe;
      uint32_t acc = 0;
      int d = 255;
      for (; d > 0; --d) {
        if (acc + hist[tid][d] >= rm) break;
        acc += hist[tid][d];
      }
      // fewer than `rm` allowed centroids in total: d ends at 0 and everything is taken
      thr_key[tid] = pass == 0 ? (uint32_t)d : ((pf << 8) | (uint32_t)d);
      need_eq[tid] = rm - acc;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) { prefix[j] = thr_key[p * 8 + j]; rem[j] = need_eq[p * 8 + j]; }
    __syncthreads();
  }
  // equal-key counts per range -> exclusive prefix over the ranges (ascending centroid order)
  uint32_t my_eq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) my_eq[j] = 0u;
  walk([&](int64_t, const uint4& v) {
#pragma unroll
    for (int j = 0; j < 8; ++j) my_eq[j] += (key8(v, j) == prefix[j]) ? 1u : 0u;
  });
#pragma unroll
  for (int j = 0; j < 8; ++j) cnt_eq[p * 8 + j][slice] = (uint16_t)my_eq[j];
  if (tid < 32) out_pos[tid] = 0u;
  __syncthreads();
  uint32_t eq_seen[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    uint32_t e = 0u;
    if ((livem >> (p * 8 + j)) & 1u)
      for (int s2 = 0; s2 < slice; ++s2) e += cnt_eq[p * 8 + j][s2];
    eq_seen[j] = e;
  }
  walk([&](int64_t c, const uint4& v) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = p * 8 + j;
      if (!((livem >> col) & 1u)) continue;
      const uint32_t k = key8(v, j);
      bool take = k > prefix[j];
      if (k == prefix[j]) { take = eq_seen[j] < rem[j]; ++eq_seen[j]; }
      if (take) {
        const uint32_t pos = atomicAdd(&out_pos[col], 1u);
        if (pos < (uint32_t)n_probe) cells[((int64_t)b * Q + g * 32 + col) * n_probe + pos] = (int32_t)c;
      }
    }
  });
  __syncthreads();
  if (tid < 32 && ((livem >> tid) & 1u)) {
    int32_t* out = cells + ((int64_t)b * Q + g * 32 + tid) * n_probe;
    for (uint32_t i = out_pos[tid]; i < (uint32_t)n_probe; ++i) out[i] = -1;
  }
}

// ---- sorted unique cells through a bitmap over the centroid ids (any number of probed cells per query) --------------------
__global__ void k_cells_mark(const int32_t* __restrict__ cells, int n, int64_t C, uint32_t* __restrict__ bm, int64_t Cw) {
  const int b = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int32_t c = cells[(int64_t)b * n + i];
    if (c >= 0 && c < C) atomicOr(&bm[(int64_t)b * Cw + (c >> 5)], 1u << (c & 31));
  }
}
__global__ __launch_bounds__(256) void k_cells_from_bitmap(const uint32_t* __restrict__ bm, int64_t Cw, int n, int32_t* __restrict__ ucells,
                                                           int32_t* __restrict__ ncells) {
  __shared__ int s[256];
  const int b = blockIdx.x;
  const uint32_t* w = bm + (int64_t)b * Cw;
  int32_t* out = ucells + (int64_t)b * n;
  int base = 0;
  for (int64_t start = 0; start < Cw; start += 256) {
    const int64_t i = start + threadIdx.x;
    const uint32_t x = i < Cw ? w[i] : 0u;
    const int c = __popc(x);
    s[threadIdx.x] = c;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const int t = ((int)threadIdx.x >= off) ? s[threadIdx.x - off] : 0;
      __syncthreads();
      s[threadIdx.x] += t;
      __syncthreads();
    }
    int pos = base + s[threadIdx.x] - c;
    uint32_t y = x;
    while (y) {
      const int bit = __ffs(y) - 1;
      if (pos < n) out[pos] = (int32_t)(i * 32 + bit);
      ++pos;
      y &= y - 1;
    }
    const int tot = s[255];
    __syncthreads();
    base += tot;
  }
  if (threadIdx.x == 0) ncells[b] = base < n ? base : n;
}

// LDS bitonic sort (ascending) of n <= cap int32 values, then unique.  one block per query.
// the same through a bitmap over the centroid ids in LDS (tables up to 2^18 centroids: 32 KiB): mark, then read the set bits back
// in order -- four barriers instead of the ~50 of the sort + scan below (a one-query search spends ~10 us in that kernel)
#define CELLS_BM_MAXC (1 << 18)
__global__ __launch_bounds__(256) void k_cells_unique_bm(const int32_t* __restrict__ cells, int n, int64_t C,
                                                         int32_t* __restrict__ ucells, int32_t* __restrict__ ncells) {
  __shared__ uint32_t bm[CELLS_BM_MAXC / 32];
  __shared__ int wtot[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int32_t* src = cells + (int64_t)b * n;
  const int nw = (int)((C + 31) / 32);                  // words in use
  const int wpt = (nw + 255) / 256;                     // consecutive words per thread
  int32_t pre[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) pre[k] = tid + 256 * k < n ? src[tid + 256 * k] : -1;   // (in flight while the bitmap is cleared)
  for (int i = tid; i < nw; i += 256) bm[i] = 0u;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (pre[k] >= 0 && pre[k] < C) atomicOr(&bm[pre[k] >> 5], 1u << (pre[k] & 31));
  for (int i = tid + 1024; i < n; i += 256) {
    const int32_t x = src[i];
    if (x >= 0 && x < C) atomicOr(&bm[x >> 5], 1u << (x & 31));
  }
  __syncthreads();
  const int w0 = tid * wpt;
  int cnt = 0;
  for (int k = 0; k < wpt; ++k)
    if (w0 + k < nw) cnt += __popc(bm[w0 + k]);
  const int lane = tid & 63, wave = tid >> 6;
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(incl, d, 64);
    if (lane >= d) incl += y;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int t = wtot[w];
    before += w < wave ? t : 0;
    total += t;
  }
  int pos = before + incl - cnt;
  int32_t* out = ucells + (int64_t)b * n;
  for (int k = 0; k < wpt; ++k) {
    if (w0 + k >= nw) break;
    uint32_t w = bm[w0 + k];
    while (w) {
      const int bit = __ffs(w) - 1;
      out[pos++] = (w0 + k) * 32 + bit;
      w &= w - 1;
    }
  }
  if (tid == 0) ncells[b] = total;
}

__global__ __launch_bounds__(256) void k_cells_unique(const int32_t* __restrict__ cells, int n, int npow2,
                                                      int32_t* __restrict__ ucells, int32_t* __restrict__ ncells) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int32_t* v = reinterpret_cast<int32_t*>(smem);  // [npow2]
  const int b = blockIdx.x;
  const int32_t* src = cells + (int64_t)b * n;
  for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
    int32_t x = (i < n) ? src[i] : -1;
    v[i] = (x < 0) ? 0x7FFFFFFF : x;
  }
  __syncthreads();
  for (int k = 2; k <= npow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
        int ixj = i ^ j;
        if (ixj > i) {
          int32_t a = v[i], c = v[ixj];
          bool up = ((i & k) == 0);
          if ((a > c) == up) { v[i] = c; v[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
  // ordered unique: position = number of distinct values before i.  Serial over chunks of
  // blockDim with a running base (n is small: <= 8192).
  int32_t* out = ucells + (int64_t)b * n;
  __shared__ int s_scan[256];
  int base = 0;
  for (int start = 0; start < npow2; start += blockDim.x) {
    int i = start + threadIdx.x;
    int flag = 0;
    if (i < npow2) {
      int32_t x = v[i];
      flag = (x != 0x7FFFFFFF) && (i == 0 || v[i - 1] != x);
    }
    s_scan[threadIdx.x] = flag;
    __syncthreads();
    for (int off = 1; off < (int)blockDim.x; off <<= 1) {
      int t = (threadIdx.x >= (unsigned)off) ? s_scan[threadIdx.x - off] : 0;
      __syncthreads();
      s_scan[threadIdx.x] += t;
      __syncthreads();
    }
    int incl = s_scan[threadIdx.x];
    if (flag) out[base + incl - 1] = v[i];
    int tot = s_scan[blockDim.x - 1];
    __syncthreads();
    base += tot;
  }
  if (threadIdx.x == 0) ncells[b] = base;
}

// ---- threshold probe: two streaming passes over S with 16-byte loads ---------------------------------
// pass A: maxima of 1024-row chunks per query column; tau = the n_probe-th largest chunk maximum
//         (each chunk maximum IS an element, so at least n_probe elements are >= tau and the true
//         top-n_probe all are);
// pass B: append every element >= tau (a handful per column) to a per-column list;
// then the exact top-n_probe (score desc, id asc) of the list.  A column whose list overflows
// (e.g. an all-equal zero-padded query column) raises a flag and the register top-k kernel above
// redoes the batch -- decided on the device, no host round trip.
#define PROBE_CHUNK 1024
#define PROBE_CAP 64
__global__ __launch_bounds__(256) void k_probe_chunkmax(const uint16_t* __restrict__ S, int64_t C, int Qp, int nchunk,
                                                        const uint32_t* __restrict__ allow, int64_t Cw,
                                                        uint16_t* __restrict__ cmax /*[B*Qp][nchunk]*/) {
  const int groups = Qp / 32;
  const int b = blockIdx.y / groups, g = blockIdx.y % groups;
  const int chunk = blockIdx.x;
  const int piece = threadIdx.x & 3, rl = threadIdx.x >> 2;  // 8 columns, row lane 0..63
  const half_t ninf = __builtin_bit_cast(half_t, (uint16_t)0xFC00);
  h2 m[4] = {h2{ninf, ninf}, h2{ninf, ninf}, h2{ninf, ninf}, h2{ninf, ninf}};
  const uint16_t* Sb = S + ((int64_t)b * C) * Qp + g * 32 + piece * 8;
  const uint32_t* al = allow ? allow + (int64_t)b * Cw : nullptr;
#pragma unroll 4
  for (int i = 0; i < PROBE_CHUNK / 64; ++i) {
    const int64_t c = (int64_t)chunk * PROBE_CHUNK + i * 64 + rl;
    if (c < C && (!al || ((al[c >> 5] >> (c & 31)) & 1u))) {
      const uint4 v = *reinterpret_cast<const uint4*>(Sb + c * Qp);
      m[0] = pk_max(m[0], u32_as_h2(v.x)); m[1] = pk_max(m[1], u32_as_h2(v.y));
      m[2] = pk_max(m[2], u32_as_h2(v.z)); m[3] = pk_max(m[3], u32_as_h2(v.w));
    }
  }
#pragma unroll
  for (int s = 4; s < 64; s <<= 1)
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = pk_max(m[j], u32_as_h2(shfl_xor_u32(h2_as_u32(m[j]), s)));
  __shared__ uint32_t red[4][4][4];  // [wave][piece][j]
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave][piece][j] = h2_as_u32(m[j]);
  }
  __syncthreads();
  if (threadIdx.x < 16) {  // thread = (piece p, j): two columns
    const int p = threadIdx.x >> 2, j = threadIdx.x & 3;
    h2 r = u32_as_h2(red[0][p][j]);
#pragma unroll
    for (int w = 1; w < 4; ++w) r = pk_max(r, u32_as_h2(red[w][p][j]));
    const int q = g * 32 + p * 8 + j * 2;
    // (explicit integer extraction: `bit_cast<uint16_t>(r.y)` was compiled to a store of the LOW half)
    const uint32_t ru = h2_as_u32(r);
    cmax[((int64_t)b * Qp + q) * nchunk + chunk] = (uint16_t)(ru & 0xFFFFu);
    cmax[((int64_t)b * Qp + q + 1) * nchunk + chunk] = (uint16_t)(ru >> 16);
  }
}

// one wave per (b, q): tau = n_probe-th largest chunk maximum (as a mono16 key; 0 = collect everything).
// The chunk maxima of the column are read ONCE into registers (KPL keys per lane); the n_probe selection rounds then run
// on registers (re-reading global memory every round cost 24 us at cfg2).
template <int KPL>
__global__ __launch_bounds__(64) void k_probe_tau(const uint16_t* __restrict__ cmax, int Q, int Qp, int nchunk, int n_probe,
                                                  uint32_t* __restrict__ tau /*[B*Qp]*/, int upper /*the maxima are upper bounds, at most one fp16 step above an element*/,
                                                  const float* __restrict__ wcol /*nullable: S1's lazy form -- the column's window, tau is lowered by s1_lower16*/, float kappa,
                                                  float* __restrict__ lz_tight /*[B*Qp], with wcol*/, float* __restrict__ lz_loose, float inv_w0) {
  const int b = blockIdx.x / Q, q = blockIdx.x % Q;
  const int lane = threadIdx.x;
  const uint16_t* src = cmax + ((int64_t)b * Qp + q) * nchunk;
  if (nchunk < n_probe) {
    if (lane == 0) {
      tau[(int64_t)b * Qp + q] = 0u;
      if (wcol && lz_tight) { lz_tight[(int64_t)b * Qp + q] = __builtin_inff(); lz_loose[(int64_t)b * Qp + q] = __builtin_inff(); }   // (never with the lazy form: fpk_probe_lazy_ok)
    }
    return;
  }
  const float w_pre = wcol ? wcol[(int64_t)b * Qp + q] : 0.f;   // (in flight with the maxima: its consumers sit behind the selection)
  uint32_t key[KPL];   // (mono16 << 12 | reversed chunk index): unique, chunks < 4096; 0 = no chunk
#pragma unroll
  for (int j = 0; j < KPL; ++j) {
    const int i = lane + 64 * j;
    key[j] = (i < nchunk) ? ((mono16(src[i]) << 12) | (uint32_t)(nchunk - 1 - i)) : 0u;
  }
  uint32_t bound = 0xFFFFFFFFu, m = 0, m_first = 0;
  for (int r = 0; r < n_probe; ++r) {
    m = 0;
#pragma unroll
    for (int j = 0; j < KPL; ++j) m = (key[j] < bound && key[j] > m) ? key[j] : m;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      const uint32_t o = shfl_xor_u32(m, s);
      m = o > m ? o : m;
    }
    bound = m;
    if (r == 0) m_first = m;
  }
  if (wcol && lz_tight && lane == 0) {
    // S1's lazy form, the selection's slack (FpLazyS1): a stored column maximum s of a document exceeds the reference's by at most
    // ulp16(s) + 2 u2(s) (s1_lower16: one fp16 step while the window is small against the step, window + step otherwise; ulp16
    // clamped from below at 2^-14).  tight: s <= the column's overall maximum (the first pick above), valid while s >= 0; loose: any
    // |s| <= |q| |c|max = wcol / w0.  Zero columns hold exact zeros.
    const float w = w_pre;
    float tg = 0.f, ls = 0.f;
    if (w > 0.f) {
      const uint16_t hmax = unmono16(m_first >> 12);
      const float smax = __builtin_fabsf((float)__builtin_bit_cast(half_t, hmax));
      uint32_t e = hmax & 0x7C00u;
      e = (e < 0x2C00u ? 0x2C00u : e) - 0x2800u;
      tg = (float)__builtin_bit_cast(half_t, (uint16_t)e) + 2.f * s1_u2(smax, w, kappa);
      const float bq = w * inv_w0 * 1.002f;
      int ex = 0;
      (void)__builtin_frexpf(bq, &ex);   // bq = f * 2^ex, f in [0.5, 1): values below 2^ex have fp16 steps of at most 2^(ex - 11)
      ex = ex - 11 < -14 ? -14 : ex - 11;
      ls = __builtin_ldexpf(1.f, ex) + 2.f * s1_u2(bq, w, kappa);
      if (!(ls >= tg)) ls = tg;   // (NaN / a maximum beyond the norm bound: keep the larger)
    }
    lz_tight[(int64_t)b * Qp + q] = tg;
    lz_loose[(int64_t)b * Qp + q] = ls;
  }
  if (lane == 0) {
    uint32_t t = m >> 12;  // mono16 of the n_probe-th largest chunk maximum
    // S1's exact mode takes the maxima over the upper candidates of the flagged entries: every chunk then holds an element of
    // at least the fp16 value one step BELOW its maximum, so that is what n_probe elements are guaranteed to reach
    // The maxima S1 emits are taken over upper candidates h(x + u) (of the flagged entries in the eager form, of every entry in the
    // lazy one), and near zero the window u spans several fp16 steps: the sound lower end of what an element stored as t can be is
    // s1_lower16 (round 4 stepped one fp16 value down, which undershoots for |x| < ~2^-10).
    // An all-zero (padded) query row -- window 0 -- scores exactly 0 against every centroid: C-way tie.  Its cells are the
    // lowest-numbered centroids by the tie rule (k_probe_merge writes them); nothing is collected (0x10000 is above every key), so
    // list-padded batches (fast_plaid.py:772-780) neither overflow the tie room nor leave the lazy form.
    if (wcol && !(w_pre > 0.f)) t = 0x10000u;
    else if (wcol) { if (t > 0u) t = mono16(s1_lower16(unmono16(t), w_pre, kappa)); }
    else if (upper && t > 0u) { t -= 1u; if (t == 0x7FFFu) t = 0x7FFEu; }   // (no window given: one step; key 0x7FFF is -0, which mono16 folds into +0)
    tau[(int64_t)b * Qp + q] = t;
  }
}

static void launch_probe_tau(const uint16_t* cmax, const FpSearchShape& sh, int nchunk, uint32_t* tau, hipStream_t st,
                             const FpS1Exact* up = nullptr /*the maxima are S1's upper candidates: the threshold is lowered by the window*/,
                             const FpLazyS1* lz = nullptr) {
  const dim3 grid((unsigned)(sh.B * sh.Q));
  const int upper = up ? 1 : 0;
  const float* wcol = lz ? lz->wcol : (up ? up->wcol : nullptr);
  const float kappa = lz ? lz->kappa : (up ? up->kappa : 0.f), inv_w0 = lz ? lz->inv_w0 : 0.f;
  float* tg = lz ? lz->tight : nullptr;
  float* ls = lz ? lz->loose : nullptr;
  if (nchunk <= 64 * 4) hipLaunchKernelGGL(k_probe_tau<4>, grid, dim3(64), 0, st, cmax, sh.Q, sh.Qp, nchunk, sh.n_probe, tau, upper, wcol, kappa, tg, ls, inv_w0);
  else if (nchunk <= 64 * 16) hipLaunchKernelGGL(k_probe_tau<16>, grid, dim3(64), 0, st, cmax, sh.Q, sh.Qp, nchunk, sh.n_probe, tau, upper, wcol, kappa, tg, ls, inv_w0);
  else hipLaunchKernelGGL(k_probe_tau<64>, grid, dim3(64), 0, st, cmax, sh.Q, sh.Qp, nchunk, sh.n_probe, tau, upper, wcol, kappa, tg, ls, inv_w0);   // nchunk <= 4096
}

__global__ __launch_bounds__(256) void k_probe_collect(const uint16_t* __restrict__ S, int64_t C, int Q, int Qp,
                                                       const uint32_t* __restrict__ allow, int64_t Cw,
                                                       const uint32_t* __restrict__ tau, uint32_t* __restrict__ cnt /*[B*Qp]*/,
                                                       unsigned long long* __restrict__ cand /*[B*Qp][PROBE_CAP]*/,
                                                       int32_t* __restrict__ flag, const uint16_t* __restrict__ cmax128, int nch128) {
  const int groups = Qp / 32;
  const int b = blockIdx.y / groups, g = blockIdx.y % groups;
  const int chunk = blockIdx.x;
  const int piece = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int q0 = g * 32 + piece * 8;
  // which of this block's eight 128-row sub-chunks hold a value >= tau in at least one of the 32 columns
  __shared__ uint32_t s_live;
  uint32_t live = 0xFFu;
  if (cmax128) {
    if (threadIdx.x == 0) s_live = 0u;
    __syncthreads();
    const int sc = threadIdx.x >> 5, j = threadIdx.x & 31;       // (sub-chunk, column)
    const int64_t sci = (int64_t)chunk * (PROBE_CHUNK / 128) + sc;
    if (sci < nch128 && g * 32 + j < Q) {
      const int64_t col = (int64_t)b * Qp + g * 32 + j;
      if (mono16(cmax128[col * nch128 + sci]) >= tau[col]) atomicOr(&s_live, 1u << sc);
    }
    __syncthreads();
    live = s_live;
    if (live == 0u) return;
  }
  uint32_t tq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) tq[j] = (q0 + j < Q) ? tau[(int64_t)b * Qp + q0 + j] : 0xFFFFFFFFu;  // pad columns collect nothing
  const uint16_t* Sb = S + ((int64_t)b * C) * Qp + q0;
  const uint32_t* al = allow ? allow + (int64_t)b * Cw : nullptr;
  // (the rows of a live sub-chunk pair are fetched together, four loads in flight per thread: one load per iteration in front of
  // its compares left the kernel at 3.2 TB/s on the ~22 % of S it reads)
  constexpr int NIT = PROBE_CHUNK / 64;
#pragma unroll 1
  for (int i0 = 0; i0 < NIT; i0 += 4) {
    if (!((live >> (i0 >> 1)) & 3u)) continue;   // (uniform: two sub-chunks per group of four iterations)
    uint4 vv[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k;
      const int64_t c = (int64_t)chunk * PROBE_CHUNK + i * 64 + rl;
      ok[k] = ((live >> (i >> 1)) & 1u) && c < C && !(al && !((al[c >> 5] >> (c & 31)) & 1u));
      vv[k] = make_uint4(0, 0, 0, 0);
      if (ok[k]) vv[k] = *reinterpret_cast<const uint4*>(Sb + c * Qp);
    }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!ok[k]) continue;
    const int i = i0 + k;
    const int64_t c = (int64_t)chunk * PROBE_CHUNK + i * 64 + rl;
    const uint4 v = vv[k];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint16_t hv = (uint16_t)(w[j >> 1] >> (16 * (j & 1)));
      const uint32_t key = mono16(hv);
      if (key >= tq[j]) {
        const int64_t col = (int64_t)b * Qp + q0 + j;
        const uint32_t pos = atomicAdd(&cnt[col], 1u);
        if (pos < PROBE_CAP)
          cand[col * PROBE_CAP + pos] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)c);
        else
          *flag = 1;
      }
    }
  }
  }
}

template <int NP>
static void launch_probe_partial(const uint16_t* S, const FpIndexDev& ix, const FpSearchShape& sh, int nchunk,
                                 const uint32_t* allow, int64_t Cw, unsigned long long* partial, const int32_t* flag, hipStream_t st) {
  dim3 grid((unsigned)nchunk, (unsigned)(sh.B * (sh.Qp / 32)));
  hipLaunchKernelGGL(k_probe_partial<NP>, grid, dim3(256), 0, st, S, ix.C, sh.Q, sh.Qp, nchunk, allow, Cw, partial, flag);
}

static int next_pow2(int x) { return fp_next_pow2(x); }

// scratch layout inside `partial` (bytes): [cand B*Qp*CAP*8][fallback partial B*Qp*nchunk*8*NP*8][cmax B*Qp*nch2*2][tau B*Qp*4][cnt B*Qp*4][flag 4]
size_t fpk_probe_scratch_bytes(const FpIndexDev& ix, const FpSearchShape& sh, int nchunk) {
  int NP = 1;
  while (NP < (sh.n_probe > 1 ? sh.n_probe : 1)) NP <<= 1;
  const size_t cols = (size_t)sh.B * sh.Qp;
  const size_t nch2 = (size_t)((ix.C + PROBE_CHUNK - 1) / PROBE_CHUNK);
  if (NP > 32) NP = 32;
  const size_t bitmap = (int64_t)sh.Q * sh.n_probe > FP_MAX_CELLS ? (size_t)sh.B * ((ix.C + 31) / 32) * 4 : 0;
  return cols * PROBE_CAP * 8 + cols * nchunk * 8 * NP * 8 + ((cols * nch2 * 2 + 15) & ~(size_t)15) + cols * 4 + cols * 4 + 64 + bitmap;
}

namespace {
struct ProbeLayout {
  int NP, nch2;
  bool big_probe, threshold_ok;
  size_t cols;
  unsigned long long *cand, *fb_partial;
  uint16_t* cmax;
  uint32_t *tau, *cnt;
  int32_t* flag;
};
ProbeLayout probe_layout(const FpIndexDev& ix, const FpSearchShape& sh, int nchunk, unsigned long long* partial) {
  ProbeLayout L;
  L.NP = 1;
  while (L.NP < sh.n_probe && L.NP < 32) L.NP <<= 1;   // register top-k fallback: n_probe <= 32 (beyond: k_probe_select)
  L.big_probe = sh.n_probe > 32;
  L.cols = (size_t)sh.B * sh.Qp;
  L.nch2 = (int)((ix.C + PROBE_CHUNK - 1) / PROBE_CHUNK);
  unsigned char* base = reinterpret_cast<unsigned char*>(partial);
  L.cand = reinterpret_cast<unsigned long long*>(base);
  L.fb_partial = reinterpret_cast<unsigned long long*>(base + L.cols * PROBE_CAP * 8);
  unsigned char* p2 = base + L.cols * PROBE_CAP * 8 + L.cols * nchunk * 8 * L.NP * 8;
  L.cmax = reinterpret_cast<uint16_t*>(p2);
  p2 += (L.cols * L.nch2 * 2 + 15) & ~(size_t)15;
  L.tau = reinterpret_cast<uint32_t*>(p2);
  L.cnt = L.tau + L.cols;
  L.flag = reinterpret_cast<int32_t*>(L.cnt + L.cols);
  static const int force_fb = getenv("FP_PROBE_FALLBACK") ? 1 : 0;
  // the chunk index must fit the 12 low bits of the tau keys; room for ties at the cut in the candidate list
  L.threshold_ok = L.nch2 <= 4096 && sh.n_probe <= PROBE_CAP / 2 && !force_fb;
  return L;
}
}  // namespace

bool fpk_probe_lazy_ok(const FpIndexDev& ix, const FpSearchShape& sh, int nchunk) {
  const ProbeLayout L = probe_layout(ix, sh, nchunk, nullptr);
  return L.threshold_ok && sh.n_probe >= 1 && (ix.C + 127) / 128 <= 4096;
}
bool fpk_probe_zero_region(const FpIndexDev& ix, const FpSearchShape& sh, int nchunk, unsigned long long* partial, void** p, size_t* bytes) {
  const ProbeLayout L = probe_layout(ix, sh, nchunk, partial);
  *p = L.cnt;                     // (16-byte aligned: cols is a multiple of 32)
  *bytes = L.cols * 4 + 4;        // counters + flag; rounded up to 16 it ends where the cell bitmap begins
  return L.threshold_ok;
}
const int32_t* fpk_probe_flag(const FpIndexDev& ix, const FpSearchShape& sh, int nchunk, const unsigned long long* partial) {
  return probe_layout(ix, sh, nchunk, const_cast<unsigned long long*>(partial)).flag;
}

int fpk_probe(const FpIndexDev& ix, const uint16_t* S, const FpSearchShape& sh, const uint32_t* allow,
              unsigned long long* partial, int nchunk, int32_t* cells, int32_t* ucells, int32_t* ncells,
              const uint16_t* cmax128 /*nullable: [B*Qp][ceil(C/128)] from S1*/, hipStream_t st, bool prezeroed, bool with_fallback,
              const FpS1Exact* cmax_upper, const FpLazyS1* lz) {
  const int64_t Cw = (ix.C + 31) / 32;
  const ProbeLayout L = probe_layout(ix, sh, nchunk, partial);
  const int NP = L.NP, nch2 = L.nch2;
  const bool big_probe = L.big_probe;
  unsigned long long *cand = L.cand, *fb_partial = L.fb_partial;
  uint16_t* cmax = L.cmax;
  uint32_t *tau = L.tau, *cnt = L.cnt;
  int32_t* flag = L.flag;
  const bool threshold_ok = L.threshold_ok;
  if (threshold_ok) {
    // counters and the flag (the merge reads cnt[col] entries of a column's candidate list, nothing behind them)
    if (!prezeroed) (void)hipMemsetAsync(cnt, 0, L.cols * 4 + 4, st);
    dim3 grid((unsigned)nch2, (unsigned)(sh.B * (sh.Qp / 32)));
    // S1's 128-centroid column maxima serve as the chunk maxima when no subset masks centroids and their index fits 12 bits:
    // tau is tighter, and the collect pass skips the 128-row sub-chunks that hold nothing >= tau in any of their 32 columns
    const int nch128 = (int)((ix.C + 127) / 128);
    const bool fused = cmax128 != nullptr && allow == nullptr && nch128 <= 4096;
    if (lz && !fused) return -2;   // (the caller only asks for the lazy form where the fused threshold probe applies)
    if (fused) {
      launch_probe_tau(cmax128, sh, nch128, tau, st, cmax_upper, lz);
      hipLaunchKernelGGL(k_probe_collect, grid, dim3(256), 0, st, S, ix.C, sh.Q, sh.Qp, allow, Cw, tau, cnt, cand, flag, cmax128, nch128);
    } else {
      hipLaunchKernelGGL(k_probe_chunkmax, grid, dim3(256), 0, st, S, ix.C, sh.Qp, nch2, allow, Cw, cmax);
      launch_probe_tau(cmax, sh, nch2, tau, st);
      hipLaunchKernelGGL(k_probe_collect, grid, dim3(256), 0, st, S, ix.C, sh.Q, sh.Qp, allow, Cw, tau, cnt, cand, flag,
                         (const uint16_t*)nullptr, 0);
    }
    hipLaunchKernelGGL(k_probe_merge, dim3((unsigned)(sh.B * sh.Q)), dim3(64), 0, st, cand, sh.Q, sh.Qp, PROBE_CAP, sh.n_probe, cells,
                       flag, 0, cnt, lz ? ix.centroids : (const uint16_t*)nullptr, lz ? lz->qpad : (const uint16_t*)nullptr, ix.dim,
                       fused ? (lz ? lz->wcol : (cmax_upper ? cmax_upper->wcol : (const float*)nullptr)) : (const float*)nullptr);
  } else {
    if (lz) return -2;
    (void)hipMemsetAsync(flag, 0xFF, 4, st);  // force the register top-k path
  }
  // fallback (its kernels exit at once unless the flag is set).  Without it a flagged batch has no cells at all (k_probe_merge
  // blanks them) and the caller, who sees the flag after its final sync, runs the batch again with the fallback.
  if (threshold_ok && !with_fallback) {
  } else if (big_probe) {
    hipLaunchKernelGGL(k_probe_select, dim3((unsigned)(sh.B * (sh.Qp / 32))), dim3(1024), 0, st, S, ix.C, sh.Q, sh.Qp, sh.n_probe, allow, Cw,
                       cells, flag);
  } else {
  switch (NP) {
    case 1: launch_probe_partial<1>(S, ix, sh, nchunk, allow, Cw, fb_partial, flag, st); break;
    case 2: launch_probe_partial<2>(S, ix, sh, nchunk, allow, Cw, fb_partial, flag, st); break;
    case 4: launch_probe_partial<4>(S, ix, sh, nchunk, allow, Cw, fb_partial, flag, st); break;
    case 8: launch_probe_partial<8>(S, ix, sh, nchunk, allow, Cw, fb_partial, flag, st); break;
    case 16: launch_probe_partial<16>(S, ix, sh, nchunk, allow, Cw, fb_partial, flag, st); break;
    default: launch_probe_partial<32>(S, ix, sh, nchunk, allow, Cw, fb_partial, flag, st); break;
  }
  hipLaunchKernelGGL(k_probe_merge, dim3((unsigned)(sh.B * sh.Q)), dim3(64), 0, st, fb_partial, sh.Q, sh.Qp, nchunk * 8 * NP, sh.n_probe,
                     cells, flag, 1, (const uint32_t*)nullptr, (const uint16_t*)nullptr, (const uint16_t*)nullptr, 0, (const float*)nullptr);
  }
  const int64_t n64 = (int64_t)sh.Q * sh.n_probe;
  if (n64 > 0x7FFFFFFFll / 4) return -1;
  const int n = (int)n64;
  if (n <= FP_MAX_CELLS) {
    const int np2 = next_pow2(n);
    static const bool cells_bm = fp_test_opt("cells_bm", 1) != 0;
    if (cells_bm && ix.C <= CELLS_BM_MAXC)
      hipLaunchKernelGGL(k_cells_unique_bm, dim3((unsigned)sh.B), dim3(256), 0, st, cells, n, ix.C, ucells, ncells);
    else
      hipLaunchKernelGGL(k_cells_unique, dim3((unsigned)sh.B), dim3(256), (size_t)np2 * 4, st, cells, n, np2, ucells, ncells);
  } else {   // more probed cells than the LDS sort holds: mark them in a bitmap over the centroid ids and read it back in order
    uint32_t* bm = reinterpret_cast<uint32_t*>(flag + 4);   // [B][Cw] behind the flag (fpk_probe_scratch_bytes)
    (void)hipMemsetAsync(bm, 0, (size_t)sh.B * Cw * 4, st);
    hipLaunchKernelGGL(k_cells_mark, dim3((unsigned)std::min<int64_t>(64, (n + 255) / 256), (unsigned)sh.B), dim3(256), 0, st, cells, n, ix.C, bm, Cw);
    hipLaunchKernelGGL(k_cells_from_bitmap, dim3((unsigned)sh.B), dim3(256), 0, st, bm, Cw, n, ucells, ncells);
  }
  return 0;
}

// ============================================================================================
// subset support (search.rs:494-517, :407-439): subset doc bitmap, allowed-centroid bitmap
// ============================================================================================
// Subset bitmaps (search.rs:494-517, :430-439): per query the documents of its subset and the centroids those documents
// contain.  Round 6: half a wave per document (its unique codes read coalesced), the centroid bitmap accumulated in LDS and
// flushed once per workgroup (the first form -- a thread per id walking its codes with one global atomic each, 8192 threads per
// query -- took 4.8 ms for 64 x 50 k ids and 23 ms for 64 x 300 k).
__global__ __launch_bounds__(256) void k_subset_prepare(const int64_t* __restrict__ ids, const int64_t* __restrict__ off, int64_t N,
                                 const int64_t* __restrict__ uoff, const int32_t* __restrict__ ucodes,
                                 uint32_t* __restrict__ subbm, int64_t W, uint32_t* __restrict__ allow, int64_t Cw,
                                 int32_t* __restrict__ invalid, int lds_words /*Cw when the centroid bitmap fits LDS, else 0*/) {
  extern __shared__ uint32_t lbm[];
  const int b = blockIdx.y;
  const int64_t beg = off[b], end = off[b + 1];
  const int64_t chunk = (end - beg + gridDim.x - 1) / gridDim.x;
  const int64_t lo = beg + (int64_t)blockIdx.x * chunk, hi = lo + chunk < end ? lo + chunk : end;
  if (lo >= hi) return;   // (uniform)
  for (int w = threadIdx.x; w < lds_words; w += 256) lbm[w] = 0u;
  __syncthreads();
  const int l = threadIdx.x & 31, hw = threadIdx.x >> 5;   // 8 half-waves, a document each
  uint32_t* abm = allow + (int64_t)b * Cw;
  for (int64_t i = lo + hw; i < hi; i += 8) {
    const int64_t d = ids[i];
    if (d < 0 || d >= N) { if (l == 0) invalid[b] = 1; continue; }  // index_select out of range -> Err -> empty result
    if (l == 0) atomicOr(&subbm[(int64_t)b * W + (d >> 5)], 1u << (d & 31));
    const int64_t u0 = uoff[d];
    const int len = (int)(uoff[d + 1] - u0);
    for (int t = l; t < len; t += 32) {
      const int32_t c = ucodes[u0 + t];
      if (lds_words) atomicOr(&lbm[c >> 5], 1u << (c & 31));
      else atomicOr(&abm[c >> 5], 1u << (c & 31));
    }
  }
  __syncthreads();
  for (int w = threadIdx.x; w < lds_words; w += 256) {
    const uint32_t v = lbm[w];
    if (v) atomicOr(&abm[w], v);
  }
}
// rows 1 .. B-1 of a [B][words] array = row 0 (one subset shared by every query)
__global__ __launch_bounds__(256) void k_replicate_row(uint32_t* __restrict__ a, int64_t words, int B) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= words) return;
  const uint32_t v = a[w];
  for (int b = 1; b < B; ++b) a[(int64_t)b * words + w] = v;
}
void fpk_subset_prepare(const FpIndexDev& ix, const int64_t* sub_ids, const int64_t* sub_off, int B, uint32_t* subbm,
                        int64_t W, uint32_t* allow, int64_t Cw, int32_t* invalid, hipStream_t st, int64_t max_len, int replicate_to) {
  // B lists (B = 1 with replicate_to > 1: one list shared by replicate_to queries); max_len = the longest of them
  int64_t bx = (max_len + 2047) / 2048;
  if (bx > 512) bx = 512;
  if (bx < 1) bx = 1;
  const int lds_words = Cw <= 32768 ? (int)Cw : 0;   // <= 2^20 centroids: 128 KiB
  static std::atomic<uint64_t> ok{0};
  fp_allow_big_lds((const void*)k_subset_prepare, ok, 160 * 1024);
  hipLaunchKernelGGL(k_subset_prepare, dim3((unsigned)bx, (unsigned)B), dim3(256), (size_t)lds_words * 4, st, sub_ids, sub_off, ix.N, ix.uoff, ix.ucodes,
                     subbm, W, allow, Cw, invalid, lds_words);
  if (B == 1 && replicate_to > 1) {
    hipLaunchKernelGGL(k_replicate_row, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, st, subbm, W, replicate_to);
    hipLaunchKernelGGL(k_replicate_row, dim3((unsigned)((Cw + 255) / 256)), dim3(256), 0, st, allow, Cw, replicate_to);
    hipLaunchKernelGGL(k_replicate_row, dim3(1), dim3(256), 0, st, reinterpret_cast<uint32_t*>(invalid), (int64_t)1, replicate_to);
  }
}

// ============================================================================================
// S3  IVF gather as a per-query document bitmap
// ============================================================================================
// One workgroup per (query, tile of MARK_TILE_DOCS documents): the tile's bitmap lives in LDS
// (ds_or atomics instead of tens of millions of global atomics), the probed cells' lists are
// walked as ONE concatenated stream so that skewed (Zipf) list lengths stay balanced, and the
// finished tile is written out with coalesced 16-byte stores (no memset of the bitmap needed).
// search == 0: every tile reads every list and filters by range (cheap when there are few
//              tiles: lists are short and L2-resident);
// search == 1: each list's sub-range for this tile is found by binary search first (many
//              tiles / long lists).
#define MARK_TILE_WORDS_MAX 8192   // 32 KiB of LDS bitmap = 262144 documents per tile
__global__ __launch_bounds__(1024) void k_ivf_mark(const int32_t* __restrict__ ucells, const int32_t* __restrict__ ncells,
                                                   int maxcells /*cells per pass (LDS capacity)*/, int ucstride /*row stride of ucells*/,
                                                   const int64_t* __restrict__ ivf_off,
                                                   const int32_t* __restrict__ ivf_pids, int64_t P, uint32_t* __restrict__ bitmap,
                                                   int64_t W, int search, int tw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* tile = reinterpret_cast<uint32_t*>(smem);                              // [tw]
  long long* lo_s = reinterpret_cast<long long*>(smem + (size_t)tw * 4);      // [maxcells]
  uint32_t* pre = reinterpret_cast<uint32_t*>(smem + (size_t)tw * 4 + (size_t)maxcells * 8);  // [maxcells + 1]
  __shared__ uint32_t s_scan[1024];
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int64_t word0 = (int64_t)blockIdx.x * tw;
  const int32_t tile_lo = (int32_t)(word0 * 32);
  const int64_t tile_hi64 = (word0 + tw) * 32;
  const int32_t tile_hi = tile_hi64 > 0x7FFFFFFF ? 0x7FFFFFFF : (int32_t)tile_hi64;
  for (int i = tid; i < tw; i += 1024) tile[i] = 0u;
  const int nc_all = ncells[b];
  const int lane = tid & 63, wave = tid >> 6;
  // the probed cells are walked in chunks of at most `maxcells` (the LDS arrays' capacity)
  for (int cb = 0; cb < nc_all; cb += maxcells) {
  const int nc = (nc_all - cb) < maxcells ? (nc_all - cb) : maxcells;
  __syncthreads();   // the previous chunk's lo_s / pre are no longer read
  // phase 1: per-cell [lo, lo+len) and exclusive prefix of the lengths.  With the per-tile range search, TWO threads per cell: one
  // finds the list's first entry >= tile_lo, its neighbour the first >= tile_hi (the two searches one after the other were ~20
  // dependent L2 round trips, a quarter of this workgroup's life)
  uint32_t base = 0;
  const int cpr = search ? 512 : 1024;            // cells per round
  const int sl = search ? (tid >> 1) : tid;       // this thread's cell slot in the round
  const int which = search ? (tid & 1) : 0;
  for (int start = 0; start < nc; start += cpr) {
    const int j = start + sl;
    uint32_t len = 0;
    long long lo = 0;
    if (j < nc) {
      const int32_t cell = ucells[(int64_t)b * ucstride + cb + j];
      if (cell >= 0 && cell < P) {
        long long beg = ivf_off[cell], end = ivf_off[cell + 1];
        if (search) {
          const int32_t target = which ? tile_hi : tile_lo;
          long long l = beg, h = end;
          while (l < h) { long long m = (l + h) >> 1; if (ivf_pids[m] < target) l = m + 1; else h = m; }
          beg = l;   // which == 0: the range's start; which == 1: its end
        }
        lo = beg;
        len = (uint32_t)(end - beg);
      }
    }
    if (search) {   // (lanes 2k, 2k + 1 hold the same cell; both valid or both not)
      const long long other = (long long)(((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)((unsigned long long)lo >> 32), 1, 64) << 32) |
                                          (unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)lo, 1, 64));
      len = (which == 0 && j < nc) ? (uint32_t)(other - lo) : 0u;
      if (which == 0) s_scan[sl] = len; else s_scan[512 + sl] = 0u;
    } else {
      s_scan[tid] = len;
    }
    if (which == 0 && j < nc) lo_s[j] = lo;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      uint32_t v = (tid >= off) ? s_scan[tid - off] : 0u;
      __syncthreads();
      s_scan[tid] += v;
      __syncthreads();
    }
    if (which == 0 && j < nc) pre[j] = base + s_scan[sl] - len;
    const uint32_t tot = s_scan[1023];
    __syncthreads();
    base += tot;
  }
  if (tid == 0) pre[nc] = base;
  __syncthreads();
  const uint32_t total = base;
  // phase 2: walk the concatenated stream, 512 elements per wave-chunk
  for (uint32_t c0 = (uint32_t)wave * 512u; c0 < total; c0 += 16u * 512u) {
    int cell = 0;
    {  // last cell with pre[cell] <= c0
      int l = 0, h = nc;
      while (h - l > 1) { int m = (l + h) >> 1; if (pre[m] <= c0) l = m; else h = m; }
      cell = l;
    }
    int32_t pid[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t g = c0 + (uint32_t)k * 64u + (uint32_t)lane;
      pid[k] = -1;
      if (g < total) {
        while (g >= pre[cell + 1]) ++cell;
        pid[k] = ivf_pids[lo_s[cell] + (long long)(g - pre[cell])];
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int32_t p = pid[k];
      if (p >= tile_lo && p < tile_hi) atomicOr(&tile[(p - tile_lo) >> 5], 1u << (p & 31));
    }
  }
  }
  __syncthreads();
  // phase 3: write the tile (W is a multiple of 64 words; tiles may overhang the end)
  uint32_t* dst = bitmap + (int64_t)b * W + word0;
  for (int i = tid * 4; i < tw; i += 4096) {
    if (word0 + i < W) *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(tile + i);
  }
}

void fpk_ivf_mark(const FpIndexDev& ix, const int32_t* ucells, const int32_t* ncells, int ucstride, int B, uint32_t* bitmap,
                  int64_t W, hipStream_t st) {
  const int maxcells = ucstride < FP_MAX_CELLS ? ucstride : FP_MAX_CELLS;   // probed cells per pass of the kernel
  // tile size: the largest that still gives the chip ~4 workgroups per CU (measured at cfg2: B=64 -> 2048 words 0.11 ms vs 8192 words 0.26 ms; B=8 -> 512 words 0.04 vs 0.22 ms)
  int tw = MARK_TILE_WORDS_MAX;
  while (tw > 512 && ((W + tw - 1) / tw) * B < 1024) tw >>= 1;
  const int ntile = (int)((W + tw - 1) / tw);
  const size_t lds = (size_t)tw * 4 + (size_t)maxcells * 8 + (size_t)(maxcells + 1) * 4 + 16;
  static std::atomic<uint64_t> lds_ok{0};
  fp_allow_big_lds((const void*)k_ivf_mark, lds_ok, 144 * 1024);
  const int search = ntile > 8 ? 1 : 0;
  hipLaunchKernelGGL(k_ivf_mark, dim3((unsigned)ntile, (unsigned)B), dim3(1024), lds, st, ucells, ncells, maxcells, ucstride, ix.ivf_off,
                     ix.ivf_pids, ix.P, bitmap, W, search, tw);
}

// ---- ordered compaction of the bitmaps ------------------------------------------------------
// words per thread / per workgroup of the counting and compaction kernels: 2 / 512 (16384 documents: at the ~33 % candidate
// density of cfg2 a workgroup's ids fit ONE pass through its 8192-id LDS stage; with 4 / 1024 they took two passes, each of
// which walks every set bit: 46 against 31 us)
#define CAND_WPT_N 2
int fpk_cand_words_per_block() { return 256 * CAND_WPT_N; }
// ---- "last workgroup finishes the job": the count -> scan -> offsets chains below were three or four launches of which only the
// first has real work; each extra launch is ~4.5 us of dispatch tail (40 % of a one-query search was such tails).  Every
// workgroup publishes its count, then takes a ticket; the one that draws the last ticket of its query scans the query's
// counts, and the one that finishes the last query writes the offsets.  Tickets are zeroed by the batch's first kernel.
//   No fences: an agent-scope release (__threadfence) writes back the XCD's whole L2 -- with 10^4 workgroups doing it
//   k_l0_count took 2.1 ms instead of 20 us.  Counts are PUBLISHED with device-scope atomic exchanges (performed at the
//   coherence point; thread 0 waits for their return before it draws the ticket) and READ with device-scope atomic loads
//   after the ticket said "last": per-location coherence of atomics is all this needs.  What the last workgroup writes
//   is read by the NEXT kernel only.
__device__ __forceinline__ void fp_publish(int32_t* slot, int32_t v) {   // thread 0 only
  const int32_t old = atomicExch(slot, v);
  asm volatile("" ::"v"(old));   // keep the returning form
}
__device__ __forceinline__ int32_t fp_read_published(const int32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool fp_ticket_last(uint32_t* ctr, uint32_t expect) {
  __shared__ uint32_t s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // thread 0's fp_publish calls have been performed
    s_last = atomicAdd(ctr, 1u) == expect - 1u ? 1u : 0u;
  }
  __syncthreads();
  return s_last != 0u;
}
// Self-test of exactly this publish / ticket / read pattern (run once per device at the first index creation, fp_engine.cpp): every
// workgroup publishes a round-dependent value over a poisoned slot and draws a ticket; the last one reads ALL slots back.  A part
// (or a driver) on which device-scope atomics were not coherent at one point would show stale slots here -- the engine then takes
// the plain count -> scan -> offsets launches for good (what FP_TICKETS=0 selects by hand).
__global__ __launch_bounds__(256) void k_ticket_selftest(int32_t* __restrict__ slots, uint32_t* __restrict__ ctr, int32_t* __restrict__ bad, int round) {
  const int n = (int)gridDim.x;
  if (threadIdx.x == 0) fp_publish(slots + blockIdx.x, (int32_t)((blockIdx.x * 2654435761u) ^ (uint32_t)round) | 1);
  if (!fp_ticket_last(ctr + round, (uint32_t)n)) return;
  int wrong = 0;
  for (int i = (int)threadIdx.x; i < n; i += 256)
    if (fp_read_published(slots + i) != ((int32_t)(((uint32_t)i * 2654435761u) ^ (uint32_t)round) | 1)) ++wrong;
  if (wrong) atomicAdd(bad, wrong);
  if (threadIdx.x == 0) atomicAdd(bad + 1, 1);   // rounds that reached their last workgroup
}
__global__ void k_ticket_poison(int32_t* __restrict__ slots, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) slots[i] = 0;   // (a published value is never 0: bit 0 is set)
}
// 0 = the pattern holds on this device; != 0: stale reads, missing rounds or a HIP error
int fpk_ticket_selftest(hipStream_t st) {
  constexpr int N = 1024, ROUNDS = 16;   // more workgroups than fit the chip at once: several waves of them per round
  int32_t* buf = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&buf), (size_t)(N + ROUNDS + 2) * 4) != hipSuccess) return -1;
  int32_t* slots = buf;
  uint32_t* ctr = reinterpret_cast<uint32_t*>(buf + N);
  int32_t* bad = buf + N + ROUNDS;
  (void)hipMemsetAsync(ctr, 0, (size_t)(ROUNDS + 2) * 4, st);
  for (int r = 0; r < ROUNDS; ++r) {
    hipLaunchKernelGGL(k_ticket_poison, dim3(N / 256), dim3(256), 0, st, slots, N);
    hipLaunchKernelGGL(k_ticket_selftest, dim3(N), dim3(256), 0, st, slots, ctr, bad, r);
  }
  int32_t h[2] = {-1, -1};
  const hipError_t e1 = hipMemcpyAsync(h, bad, 8, hipMemcpyDeviceToHost, st);
  const hipError_t e2 = hipStreamSynchronize(st);
  (void)hipFree(buf);
  if (e1 != hipSuccess || e2 != hipSuccess) { (void)hipGetLastError(); return -1; }
  return (h[0] == 0 && h[1] == ROUNDS) ? 0 : 1;
}
// exclusive scan of the published v[0..nblk) in place by a 256-thread workgroup; returns the total
__device__ __forceinline__ int fp_scan_counts_256(int32_t* v, int nblk, int* s /*[256] LDS*/) {
  int base = 0;
  for (int start = 0; start < nblk; start += 256) {
    const int i = start + (int)threadIdx.x;
    const int x = (i < nblk) ? fp_read_published(v + i) : 0;
    int tot = 0;
    __syncthreads();   // (the caller may still be reading s)
    const int incl = fp_block_scan_incl<int>(x, s, &tot);
    if (i < nblk) v[i] = base + incl - x;
    base += tot;
  }
  return base;
}
// cand_off[0..B] from the per-query totals; cap / invalid / total_out as k_cand_offsets below; a 256-thread workgroup
__device__ __forceinline__ void fp_offsets_256(int32_t* ncand, int B, int64_t* cand_off, int64_t cap, int32_t* invalid, int64_t* total_out,
                                               const int32_t* probe_flag) {
  __shared__ long long so[16];
  long long base = 0;
  for (int start = 0; start < B; start += 256) {
    const int i = start + (int)threadIdx.x;
    const long long x = (i < B) ? (long long)fp_read_published(ncand + i) : 0ll;
    long long tot = 0;
    const long long incl = fp_block_scan_incl<long long>(x, so, &tot);
    if (i < B) cand_off[i] = base + incl - x;
    base += tot;
  }
  const bool over = cap > 0 && base > cap;
  if (over)
    for (int i = threadIdx.x; i < B; i += 256) { cand_off[i] = 0; invalid[i] = 1; }
  if (threadIdx.x == 0) {
    cand_off[B] = over ? 0 : base;
    if (total_out) {
      *total_out = base;
      if (probe_flag) reinterpret_cast<int32_t*>(total_out)[4] = *probe_flag;   // travels to the host with the total
    }
  }
}

// ctr != nullptr: fused form -- also leaves blkcnt scanned (exclusive, per query), ncand, cand_off and *total_out
// One workgroup counts TWO consecutive compaction blocks (waves 0 - 1 the first, waves 2 - 3 the second: 2 x CAND_WPT words per
// thread): halving the compaction blocks to 512 words doubled this kernel's workgroups, and at cfg3 (156 k of them) that cost
// more than the compaction gained.  nblk = compaction blocks per query; the grid is ceil(nblk / 2) x B.
template <int CAND_WPT>
__global__ __launch_bounds__(256) void k_cand_count(const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ subbm,
                                                    const int32_t* invalid, int64_t W, int32_t* blkcnt, int nblk,
                                                    uint32_t* ctr = nullptr, int32_t* ncand = nullptr, int B = 0, int64_t* cand_off = nullptr,
                                                    int64_t cap = 0, int32_t* invalid_rw = nullptr, int64_t* total_out = nullptr,
                                                    const int32_t* probe_flag = nullptr) {
  constexpr int WPT2 = 2 * CAND_WPT;
  const int b = blockIdx.y;
  int cnt = 0;
  if (!(invalid && invalid[b])) {
    const int64_t w0 = (int64_t)blockIdx.x * (256 * WPT2) + threadIdx.x * WPT2;   // (threads 0 .. 127: the first 256 * CAND_WPT words)
#pragma unroll
    for (int k = 0; k < WPT2; ++k) {
      int64_t w = w0 + k;
      if (w < W) {
        uint32_t x = bitmap[(int64_t)b * W + w];
        if (subbm) x &= subbm[(int64_t)b * W + w];
        cnt += __popc(x);
      }
    }
  }
  __shared__ int s[256];
#pragma unroll
  for (int x = 32; x > 0; x >>= 1) cnt += __shfl_xor(cnt, x, 64);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = cnt;
  __syncthreads();
  const int c0 = s[0] + s[1], c1 = s[2] + s[3];
  const int k0 = 2 * (int)blockIdx.x;
  const bool two = k0 + 1 < nblk;
  if (!ctr) {
    if (threadIdx.x == 0) {
      blkcnt[(int64_t)b * nblk + k0] = c0;
      if (two) blkcnt[(int64_t)b * nblk + k0 + 1] = c1;
    }
    return;
  }
  if (threadIdx.x == 0) {
    fp_publish(blkcnt + (int64_t)b * nblk + k0, c0);
    if (two) fp_publish(blkcnt + (int64_t)b * nblk + k0 + 1, c1);
  }
  if (!fp_ticket_last(ctr + b, gridDim.x)) return;
  const int total = fp_scan_counts_256(blkcnt + (int64_t)b * nblk, nblk, s);
  if (threadIdx.x == 0) fp_publish(ncand + b, total);
  if (!fp_ticket_last(ctr + B, (uint32_t)B)) return;
  fp_offsets_256(ncand, B, cand_off, cap, invalid_rw, total_out, probe_flag);
}

// exclusive scan of blkcnt per query (in place) + per-query totals; one block per query.  Optionally a second array of counts
// (blkcnt2 -> ncand2) and, with a zeroed ticket, the offsets of the FIRST array's lists by the workgroup that finishes last.
__global__ __launch_bounds__(256) void k_cand_scan(int32_t* blkcnt, int nblk, int32_t* ncand, int32_t* blkcnt2 = nullptr, int32_t* ncand2 = nullptr,
                                                   uint32_t* ticket = nullptr, int B = 0, int64_t* cand_off = nullptr, int64_t cap = 0,
                                                   int32_t* invalid = nullptr, int64_t* total_out = nullptr, const int32_t* probe_flag = nullptr) {
  const int b = blockIdx.x;
  __shared__ int s[256];
  const int total = fp_scan_counts_256(blkcnt + (int64_t)b * nblk, nblk, s);
  if (blkcnt2) {
    const int total2 = fp_scan_counts_256(blkcnt2 + (int64_t)b * nblk, nblk, s);
    if (threadIdx.x == 0) ncand2[b] = total2;
  }
  if (!ticket) {
    if (threadIdx.x == 0) ncand[b] = total;
    return;
  }
  if (threadIdx.x == 0) fp_publish(ncand + b, total);
  if (!fp_ticket_last(ticket, (uint32_t)B)) return;
  fp_offsets_256(ncand, B, cand_off, cap, invalid, total_out, probe_flag);
}

// cap > 0 (the host sized the candidate buffers from earlier batches instead of waiting for this total): a total above the
// capacity empties every list and marks every query invalid, so that nothing downstream writes; the host sees the true total
// in *total_out after the call's final sync and runs the batch again with buffers of the right size.
__global__ __launch_bounds__(256) void k_cand_offsets(int32_t* ncand, int B, int64_t* cand_off, int64_t cap = 0, int32_t* invalid = nullptr,
                                                      int64_t* total_out = nullptr, const int32_t* probe_flag = nullptr) {
  fp_offsets_256(ncand, B, cand_off, cap, invalid, total_out, probe_flag);
}

// ctr (nullable): [B + 1] zeroed tickets.  Few workgroups (a one-query search: 31): ONE launch.  Many: the tickets themselves
// would cost more than the launches they save (device-scope atomics on one address retire at ~10 ns each: 10^4 workgroups ->
// 0.3 ms), so only the B scanning workgroups take tickets -- two launches.
#define FP_TICKET_MAX_WGS 384
void fpk_cand_count(const uint32_t* bitmap, const uint32_t* subbm, const int32_t* invalid, int B, int64_t W, int32_t* blkcnt,
                    int nblk, int32_t* ncand, int64_t* cand_off, hipStream_t st, int64_t cap, int32_t* invalid_rw, int64_t* total_out,
                    uint32_t* ctr, const int32_t* probe_flag) {
  const auto count_kernel = k_cand_count<CAND_WPT_N>;
  const unsigned ngrid = (unsigned)((nblk + 1) / 2);   // (a counting workgroup covers two compaction blocks)
  if (ctr && (int64_t)ngrid * B <= FP_TICKET_MAX_WGS) {
    hipLaunchKernelGGL(count_kernel, dim3(ngrid, (unsigned)B), dim3(256), 0, st, bitmap, subbm, invalid, W, blkcnt, nblk, ctr, ncand, B,
                       cand_off, cap, invalid_rw, total_out, probe_flag);
    return;
  }
  hipLaunchKernelGGL(count_kernel, dim3(ngrid, (unsigned)B), dim3(256), 0, st, bitmap, subbm, invalid, W, blkcnt, nblk, (uint32_t*)nullptr,
                     (int32_t*)nullptr, 0, (int64_t*)nullptr, (int64_t)0, (int32_t*)nullptr, (int64_t*)nullptr, (const int32_t*)nullptr);
  if (ctr) {
    hipLaunchKernelGGL(k_cand_scan, dim3((unsigned)B), dim3(256), 0, st, blkcnt, nblk, ncand, (int32_t*)nullptr, (int32_t*)nullptr, ctr + B, B,
                       cand_off, cap, invalid_rw, total_out, probe_flag);
    return;
  }
  hipLaunchKernelGGL(k_cand_scan, dim3((unsigned)B), dim3(256), 0, st, blkcnt, nblk, ncand, (int32_t*)nullptr, (int32_t*)nullptr, (uint32_t*)nullptr,
                     0, (int64_t*)nullptr, (int64_t)0, (int32_t*)nullptr, (int64_t*)nullptr, (const int32_t*)nullptr);
  hipLaunchKernelGGL(k_cand_offsets, dim3(1), dim3(256), 0, st, ncand, B, cand_off, cap, invalid_rw, total_out, probe_flag);
}

template <int CAND_WPT>
__global__ __launch_bounds__(256) void k_cand_compact(const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ subbm,
                                                      const int32_t* __restrict__ invalid, int64_t W,
                                                      const int32_t* __restrict__ blkoff, int nblk,
                                                      const int64_t* __restrict__ cand_off, int32_t* __restrict__ cand_pid) {
  // a block covers CAND_WPB words; its ids are expanded into LDS in order and then written out coalesced (the per-thread
  // `cand_pid[pos++] = ...` stores it replaces ran at 0.85 TB/s: 108 us for 84 MB)
  constexpr int CAND_WPB = 256 * CAND_WPT;
  constexpr int STAGE = 8192;                      // ids (32 KiB) per pass
  __shared__ int32_t stage[STAGE];
  const int b = blockIdx.y;
  if (invalid && invalid[b]) return;
  const int64_t w0 = (int64_t)blockIdx.x * CAND_WPB + threadIdx.x * CAND_WPT;
  uint32_t x[CAND_WPT];
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < CAND_WPT; ++k) {
    int64_t w = w0 + k;
    x[k] = 0;
    if (w < W) {
      x[k] = bitmap[(int64_t)b * W + w];
      if (subbm) x[k] &= subbm[(int64_t)b * W + w];
    }
    cnt += __popc(x[k]);
  }
  // (independent of the scan below: in flight with the bitmap words)
  int32_t* out = cand_pid + cand_off[b] + blkoff[(int64_t)b * nblk + blockIdx.x];
  // inclusive scan of the threads' counts: inside a wave by shuffles, across the four waves through LDS (one barrier; the
  // shared-memory scan it replaces was sixteen)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(incl, d, 64);
    if (lane >= d) incl += y;
  }
  __shared__ int wtot[4];
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int t = wtot[w];
    before += w < wave ? t : 0;
    total += t;
  }
  const int mine0 = before + incl - cnt;          // rank of this thread's first id inside the block
  for (int base = 0; base < total; base += STAGE) {   // one pass up to 8192 ids
    int pos = mine0;
#pragma unroll
    for (int k = 0; k < CAND_WPT; ++k) {
      uint32_t w = x[k];
      const int32_t basepid = (int32_t)((w0 + k) * 32);
      while (w) {
        const int bit = __ffs(w) - 1;
        if (pos >= base && pos < base + STAGE) stage[pos - base] = basepid + bit;
        ++pos;
        w &= w - 1;
      }
    }
    __syncthreads();
    const int nthis = (total - base) < STAGE ? (total - base) : STAGE;
    for (int i = threadIdx.x; i < nthis; i += 256) out[base + i] = stage[i];
    __syncthreads();
  }
}

void fpk_cand_compact(const uint32_t* bitmap, const uint32_t* subbm, const int32_t* invalid, int B, int64_t W,
                      const int32_t* blkoff, int nblk, const int64_t* cand_off, int32_t* cand_pid, hipStream_t st) {
  const auto compact_kernel = k_cand_compact<CAND_WPT_N>;
  hipLaunchKernelGGL(compact_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(256), 0, st, bitmap, subbm, invalid, W, blkoff, nblk,
                     cand_off, cand_pid);
}

// ============================================================================================
// S4  approximate scores  approx[d] = sum_q max_{c in codes(d)} S[c, q]   (fp16 max, fp32 sum).
// The stage is gather-LATENCY bound (candidate -> its code list -> one 64-byte score row per
// code), so the layout maximises loads in flight rather than lanes per document:
//   * 4 lanes per candidate (each lane 16 B = 8 query columns of a 32-column chunk), i.e. 16
//     candidates per wave, 64 per 256-thread block;
//   * the per-document loop walks the UNIQUE code list, unrolled x8: eight code loads, then
//     eight independent row gathers per lane in flight;
//   * grid = (blocks, queries): the query is blockIdx.y, no per-candidate search.
// ============================================================================================
// PPD = lane quads per candidate.  1: a quad walks the document's whole code list, 8 codes a step (many candidates per query:
// every quad has several documents to pipeline).  4: quad p takes the code groups p, p + 4, ... and the four partial maxima are
// combined at the end -- a 33-code document is ONE step of row gathers instead of five dependent ones.  For the refine calls
// of the bound stages (a few thousand documents per query: the chip is short of independent chains, not of lanes; measured at
// cfg2, S4 refine: 0.457 ms with PPD 1, 0.397 ms with PPD 1 and sixteen times the workgroups, see fpk_approx).
// LZ (S1's lazy form, FpLazyS1): S holds upper candidates, so the sum below is an UPPER bound A_up of the approximate score.  How far
// the reference's score can lie below it is bounded per QUERY from the column maxima (k_probe_tau: lz_tight) -- valid as long as
// every column maximum of every scored document is non-negative, which this kernel checks on its way (negflag[b] = 1 otherwise:
// the consumers then take the bound that holds for any value, lz_loose).
template <int PPD, bool LZ = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_approx(const uint16_t* __restrict__ S, int64_t C, int Q, int Qp,
                                                const int64_t* __restrict__ cand_off, const int32_t* __restrict__ cand_pid,
                                                const int64_t* __restrict__ uoff, const int32_t* __restrict__ ucodes,
                                                float* __restrict__ approx, const int32_t* __restrict__ cnt, int64_t cap,
                                                float* __restrict__ scat, const int32_t* __restrict__ scat_idx,
                                                const int64_t* __restrict__ scat_off, uint32_t* __restrict__ negflag = nullptr) {
  // candidate lists: rows start at cand_off[b] (CSR) or, without cand_off, at b * cap (fixed-capacity rows); a row holds
  // min(cnt[b], cap) entries when cnt is given, else the whole CSR row.  Results go to approx[row position] (if given) and/or
  // scat[scat_off[b] + scat_idx[row position]].
  const int b = blockIdx.y;
  const int bxi = blockIdx.x;
  const int bx = gridDim.x;
  const int64_t beg = cand_off ? cand_off[b] : (int64_t)b * cap;
  const int64_t n = cnt ? (cnt[b] < cap ? (int64_t)cnt[b] : cap) : cand_off[b + 1] - beg;
  const int64_t sbase = scat ? scat_off[b] : 0;
  constexpr int CPB = 64 / PPD;               // candidates per block
  constexpr int TSTEP = 8 * PPD;              // codes between two steps of one quad
  const int sub = threadIdx.x & 3;            // 16-byte piece of the 64-byte row chunk
  const int grp = threadIdx.x >> 2;           // quad within the block (0..63)
  const int pp = grp % PPD;                   // which quad of the candidate
  const int t0 = 8 * pp;
  const half_t negm = (half_t)NEG_MASK_F;
  const h2 neg2 = {negm, negm};
  const uint16_t* Sb = S + (int64_t)b * C * Qp + sub * 8;
  const int64_t stride = (int64_t)bx * CPB;
  const int nch = Qp / 32;
  int64_t i = (int64_t)bxi * CPB + grp / PPD;
  // Software pipeline over the dependent chain  pid -> offsets -> codes -> rows:
  //   document metadata is fetched one document ahead, the next 8 codes (of this document, or
  //   the first 8 of the next one) are fetched while the current 8 row gathers are in flight,
  //   so only ONE round trip per 8 rows stays on the critical path.
  int64_t u0 = 0;
  int len = 0;
  int32_t code[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) code[k] = 0;
  if (i < n) {
    const int32_t pid = cand_pid[beg + i];
    u0 = uoff[pid];
    len = (int)(uoff[pid + 1] - u0);
    if (len > t0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) code[k] = ucodes[u0 + ((t0 + k < len) ? t0 + k : (len - 1))];
    }
  }
  for (; i < n; i += stride) {
    int64_t nu0 = 0;
    int nlen = 0;
    if (i + stride < n) {
      const int32_t npid = cand_pid[beg + i + stride];
      nu0 = uoff[npid];
      nlen = (int)(uoff[npid + 1] - nu0);
    }
    const int32_t* cp = ucodes + u0;
    float total = 0.f;
    bool next_loaded = false;
    for (int ch = 0; ch < nch; ++ch) {
      h2 m0 = neg2, m1 = neg2, m2 = neg2, m3 = neg2;
      const uint16_t* Sc = Sb + ch * 32;
      if (ch > 0 && len > t0) {  // further 32-column chunks restart from the quad's first codes
#pragma unroll
        for (int k = 0; k < 8; ++k) code[k] = cp[(t0 + k < len) ? t0 + k : (len - 1)];
      }
      for (int t = t0; t < len; t += TSTEP) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const uint4*>(Sc + (int64_t)code[k] * Qp);
        // next codes while the rows are in flight (max is idempotent: the tail re-reads the last code)
        if (t + TSTEP < len) {
#pragma unroll
          for (int k = 0; k < 8; ++k) code[k] = cp[(t + TSTEP + k < len) ? (t + TSTEP + k) : (len - 1)];
        } else if (ch == nch - 1 && nlen > t0) {
#pragma unroll
          for (int k = 0; k < 8; ++k) code[k] = ucodes[nu0 + ((t0 + k < nlen) ? t0 + k : (nlen - 1))];
          next_loaded = true;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          m0 = pk_max(m0, u32_as_h2(v[k].x));
          m1 = pk_max(m1, u32_as_h2(v[k].y));
          m2 = pk_max(m2, u32_as_h2(v[k].z));
          m3 = pk_max(m3, u32_as_h2(v[k].w));
        }
      }
      if constexpr (PPD > 1) {   // the candidate's quads hold maxima over disjoint code groups (a quad without codes: the mask value)
#pragma unroll
        for (int x = 4; x < 4 * PPD; x <<= 1) {
          m0 = pk_max(m0, u32_as_h2((uint32_t)__shfl_xor((int)h2_as_u32(m0), x, 64)));
          m1 = pk_max(m1, u32_as_h2((uint32_t)__shfl_xor((int)h2_as_u32(m1), x, 64)));
          m2 = pk_max(m2, u32_as_h2((uint32_t)__shfl_xor((int)h2_as_u32(m2), x, 64)));
          m3 = pk_max(m3, u32_as_h2((uint32_t)__shfl_xor((int)h2_as_u32(m3), x, 64)));
        }
      }
      // fp32 sum over the columns in ASCENDING order, like the oracle's (sum(dtype=Float), search.rs:401): the running total
      // walks the quad's four lanes, eight columns each (every lane adds its columns to the incoming total, lane l's result is
      // the one passed on).  A blocked order differs in the last bit whenever a column maximum is tiny against the total.
      const int q0 = ch * 32 + sub * 8;
      const float c0 = (q0 + 0 < Q) ? (float)m0.x : 0.f, c1 = (q0 + 1 < Q) ? (float)m0.y : 0.f;
      const float c2 = (q0 + 2 < Q) ? (float)m1.x : 0.f, c3 = (q0 + 3 < Q) ? (float)m1.y : 0.f;
      const float c4 = (q0 + 4 < Q) ? (float)m2.x : 0.f, c5 = (q0 + 5 < Q) ? (float)m2.y : 0.f;
      const float c6 = (q0 + 6 < Q) ? (float)m3.x : 0.f, c7 = (q0 + 7 < Q) ? (float)m3.y : 0.f;
#define AP_LANE_STEP(CTRL)                                                                                     \
      {                                                                                                            \
        float r = total;                                                                                           \
        r += c0; r += c1; r += c2; r += c3; r += c4; r += c5; r += c6; r += c7;                                    \
        total = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), CTRL, 0xF, 0xF, false));               \
      }
      AP_LANE_STEP(0x00)   // quad_perm [0,0,0,0]: lane 0's sum of the columns 0..7 ...
      AP_LANE_STEP(0x55)   // ... then lane 1's on top of it, ...
      AP_LANE_STEP(0xAA)
      AP_LANE_STEP(0xFF)
#undef AP_LANE_STEP
      if constexpr (LZ) {   // (columns beyond Q are zero query rows: their maxima are +0)
        if ((h2_as_u32(m0) | h2_as_u32(m1) | h2_as_u32(m2) | h2_as_u32(m3)) & 0x80008000u) negflag[b] = 1u;
      }
    }
    if (!next_loaded && nlen > t0) {  // a quad without codes in this document never reaches the prefetch slot above
#pragma unroll
      for (int k = 0; k < 8; ++k) code[k] = ucodes[nu0 + ((t0 + k < nlen) ? t0 + k : (nlen - 1))];
    }
    // (the lanes of one candidate run the same outer trip counts, so the quad exchanges above are convergent within the
    // candidate; other candidates may have exited)
    if (sub == 0 && pp == 0) {
      if (approx) approx[beg + i] = total;
      if (scat) scat[sbase + scat_idx[beg + i]] = total;
    }
    u0 = nu0;
    len = nlen;
  }
}

void fpk_approx(const FpIndexDev& ix, const uint16_t* S, const FpSearchShape& sh, const int64_t* cand_off,
                const int32_t* cand_pid, int64_t M, float* approx, hipStream_t st, const int32_t* cnt, int64_t cap, float* scat,
                const int32_t* scat_idx, const int64_t* scat_off, const FpLazyS1* lz) {
  if (M <= 0) return;
  // enough blocks to cover the largest per-query candidate list a few times over
  int64_t per_q = (M + sh.B - 1) / sh.B;
  // PPD 4 when the lists are short (the refine calls: at most a few x R documents per query)
  const bool wide = per_q <= 16 * sh.R;
  const int cpb = wide ? 16 : 64;
  const int bxmul = 2;   // block slots per expected candidate
  int64_t bx = (per_q * bxmul + cpb - 1) / cpb;
  if (bx > (wide ? 4096 : 2048)) bx = wide ? 4096 : 2048;
  if (bx < 1) bx = 1;
  // walks the per-document UNIQUE code lists (max over a multiset == max over its set).
  // (an XCD-affine query assignment was measured slower when every candidate is scored: 8.3 vs 7.6 ms, round 1; and makes no
  // difference for the refine calls of the bound stages, ~6000 documents per query: 0.445 ms either way, round 3 -- their
  // ~200 k row gathers per query hardly repeat a row, FETCH_SIZE equals the logical bytes)
  if (lz) {
    if (wide)
      hipLaunchKernelGGL((k_approx<4, true>), dim3((unsigned)bx, (unsigned)sh.B), dim3(256), 0, st, S, ix.C, sh.Q, sh.Qp, cand_off, cand_pid,
                         ix.uoff, ix.ucodes, approx, cnt, cap, scat, scat_idx, scat_off, lz->negflag);
    else
      hipLaunchKernelGGL((k_approx<1, true>), dim3((unsigned)bx, (unsigned)sh.B), dim3(256), 0, st, S, ix.C, sh.Q, sh.Qp, cand_off, cand_pid,
                         ix.uoff, ix.ucodes, approx, cnt, cap, scat, scat_idx, scat_off, lz->negflag);
    return;
  }
  if (wide)
    hipLaunchKernelGGL((k_approx<4, false>), dim3((unsigned)bx, (unsigned)sh.B), dim3(256), 0, st, S, ix.C, sh.Q, sh.Qp, cand_off, cand_pid,
                       ix.uoff, ix.ucodes, approx, cnt, cap, scat, scat_idx, scat_off, (uint32_t*)nullptr);
  else
    hipLaunchKernelGGL((k_approx<1, false>), dim3((unsigned)bx, (unsigned)sh.B), dim3(256), 0, st, S, ix.C, sh.Q, sh.Qp, cand_off, cand_pid,
                       ix.uoff, ix.ucodes, approx, cnt, cap, scat, scat_idx, scat_off, (uint32_t*)nullptr);
}

// ============================================================================================
// S4, bound-and-refine form.  The flat kernel above is pinned at the gather ceiling of an 8 MB
// table (tools/probe/row_probe: 100 G rows/s from 8 MB, 171 from 4 MB, 230 from <= 2 MB; the
// kernel runs at 92 G rows/s).  The selection only needs the exact approximate score of
// documents near the top-R cut, so:
//   (1) S1's epilogue writes S8: S -> 8-bit bins,  bin(x) = clamp(floor(128 x) + 100, 0, 255).  128 x and the
//       floor are exact for fp16 x, so bin k means (k-100)/128 <= x < (k-99)/128 with no
//       rounding caveat, and the map is monotone: the max over a document's codes of the bins
//       IS the bin of the max.  Rows shrink to 32 B, one query's slice to C*32 B (4 MB at
//       C = 2^17).
//   (2) k_approx_q8: per candidate K = sum over the real query columns of the max bin.  The
//       exact score A then satisfies  K <= 128 A + 100 Q < K + Q.  Columns whose max bin is 0
//       (x < -0.78) void the lower bound (K_lo = 0), 255 (x >= 1.21) the upper one
//       (K_hi = 0xFFFF): such documents simply always survive.
//   (3) k_q8_cut: T = R'-th largest K_lo of the query (R' = min(R, n_full, n)).  R' documents
//       have 128 A + 100 Q >= T, hence so has the R'-th best exact score, and every document of
//       the exact top-R' has K_hi + Q > T.  Survivors: K_hi >= T - Q + 1 (ties at the cut included).
//   (4) survivors are compacted in order and go through the exact fp16 kernel and the same
//       selection as before -> the selected set and its scores are identical to scoring all
//       candidates exactly (tests: fp_search == fp_search_trace, which keeps the all-exact path).
// ============================================================================================
#define Q8_OFFSET 100
#define Q8_BINS_SUM 8192   // 32 columns x 255 < 8192

__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// 2 lanes per lane PAIR (16 B = 16 query columns each), PPD pairs per candidate: pair p of a candidate takes the code
// groups p, p + PPD, ... of its list and the PPD partial maxima are combined at the end (max is associative).
// PPD = 1: 32 candidates per wave (many short candidates per query); PPD = 4: one query with few or long candidates
// still fills the chip instead of leaving several queries' slices in flight.  kq[cand] = K_hi << 16 | K_lo.
// (96 VGPRs = 5 waves/SIMD; forcing 6 changes nothing, 7-8 spill and are slower: the kernel is bound by L2 misses, not latency)
template <int PPD>
__global__ __launch_bounds__(512) void k_approx_q8(const uint8_t* __restrict__ S8, int64_t C, int Q,
                                                   const int64_t* __restrict__ cand_off, const int32_t* __restrict__ cand_pid,
                                                   const int64_t* __restrict__ uoff, const int32_t* __restrict__ ucodes,
                                                   uint32_t* __restrict__ kq, int nch, int64_t M) {
  const int DPB = (int)(blockDim.x >> 1) / PPD;   // candidates per workgroup
  constexpr int TSTEP = 8 * PPD;      // codes between two groups of one pair
  // blockIdx.y = query * nch + 32-column chunk; chunk ch writes its partial sums to kq + ch * M (nch == 1: the final ones)
  const int b = blockIdx.y / nch, chq = blockIdx.y % nch, bxi = blockIdx.x;
  Q = Q - 32 * chq;
  Q = Q > 32 ? 32 : Q;
  kq += (int64_t)chq * M;
  const int nbx = gridDim.x;
  const int64_t beg = cand_off[b];
  const int64_t n = cand_off[b + 1] - beg;
  const int sub = threadIdx.x & 1;
  const int grp = threadIdx.x >> 1;   // lane pair within the block (0..127)
  const int pp = grp % PPD;           // which pair of the candidate
  const int tfirst = 8 * pp;
  const uint8_t* Sb = S8 + ((int64_t)b * nch + chq) * C * 32 + sub * 16;
  const int64_t stride = (int64_t)nbx * DPB;
  int64_t i = (int64_t)bxi * DPB + grp / PPD;
  // A step's 8 codes are ONE 16-byte load per lane (codes 4*sub .. 4*sub+3 of the group; dword-aligned only,
  // the list buffer is padded) plus an exchange inside the lane pair: 8 four-byte loads per step cost 1.6 of
  // the kernel's 5.6 ms (ablation with synthetic codes).  Positions past the end of the list repeat the
  // group's first code (max is idempotent).
  auto load_codes = [&](const int32_t* p, int pos, int ln, int32_t (&code)[8]) {
    int4 mine;
    __builtin_memcpy(&mine, p + pos + 4 * sub, 16);
    int4 oth;  // the partner lane's four codes: DPP quad_perm [1,0,3,2]
    oth.x = __builtin_amdgcn_update_dpp(0, mine.x, 0xB1, 0xF, 0xF, true);
    oth.y = __builtin_amdgcn_update_dpp(0, mine.y, 0xB1, 0xF, 0xF, true);
    oth.z = __builtin_amdgcn_update_dpp(0, mine.z, 0xB1, 0xF, 0xF, true);
    oth.w = __builtin_amdgcn_update_dpp(0, mine.w, 0xB1, 0xF, 0xF, true);
    const int4 lo = sub ? oth : mine, hi = sub ? mine : oth;
    const int rem = ln - pos;  // >= 1
    code[0] = lo.x;
    code[1] = rem > 1 ? lo.y : lo.x;
    code[2] = rem > 2 ? lo.z : lo.x;
    code[3] = rem > 3 ? lo.w : lo.x;
    code[4] = rem > 4 ? hi.x : lo.x;
    code[5] = rem > 5 ? hi.y : lo.x;
    code[6] = rem > 6 ? hi.z : lo.x;
    code[7] = rem > 7 ? hi.w : lo.x;
  };
  int64_t u0 = 0;
  int len = 0;
  int32_t code[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) code[k] = 0;
  if (i < n) {
    const int32_t pid = cand_pid[beg + i];
    u0 = uoff[pid];
    len = (int)(uoff[pid + 1] - u0);
  }
  // (the lanes of a pair always agree on len / nlen, so the exchanges inside load_codes are convergent per pair)
  if (tfirst < len) load_codes(ucodes + u0, tfirst, len, code);
  for (; i < n; i += stride) {
    int64_t nu0 = 0;
    int nlen = 0;
    if (i + stride < n) {
      const int32_t npid = cand_pid[beg + i + stride];
      nu0 = uoff[npid];
      nlen = (int)(uoff[npid + 1] - nu0);
    }
    const int32_t* cp = ucodes + u0;
    uint32_t me[4] = {0u, 0u, 0u, 0u}, mo[4] = {0u, 0u, 0u, 0u};  // running maxima: even / odd bytes as u16 pairs
    for (int t = tfirst; t < len; t += TSTEP) {
      uint4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const uint4*>(Sb + (int64_t)code[k] * 32);
      if (t + TSTEP < len) {
        load_codes(cp, t + TSTEP, len, code);
      } else if (tfirst < nlen) {
        load_codes(ucodes + nu0, tfirst, nlen, code);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          me[j] = pk_max_u16(me[j], w[j] & 0x00FF00FFu);
          mo[j] = pk_max_u16(mo[j], (w[j] >> 8) & 0x00FF00FFu);
        }
      }
    }
    if (tfirst >= len && tfirst < nlen) load_codes(ucodes + nu0, tfirst, nlen, code);  // this pair had no group in this list
    if constexpr (PPD > 1) {  // combine the pairs of the candidate (its 2*PPD lanes are consecutive and converged here)
#pragma unroll
      for (int m = 2; m < 2 * PPD; m <<= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          me[j] = pk_max_u16(me[j], shfl_xor_u32(me[j], m));
          mo[j] = pk_max_u16(mo[j], shfl_xor_u32(mo[j], m));
        }
      }
    }
    // sum / min / max of this lane's real columns (column = sub*16 + 4*j + byte)
    uint32_t sum = 0, mn = 255u, mx = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint32_t val = (((t & 1) ? mo[j] : me[j]) >> (16 * (t >> 1))) & 0xFFFFu;
        const bool real = (sub * 16 + 4 * j + t) < Q;
        sum += real ? val : 0u;
        mn = real ? min(mn, val) : mn;
        mx = real ? max(mx, val) : mx;
      }
    }
    sum += __shfl_xor(sum, 1, 64);
    mn = min(mn, (uint32_t)__shfl_xor(mn, 1, 64));
    mx = max(mx, (uint32_t)__shfl_xor(mx, 1, 64));
    if (sub == 0 && pp == 0) {
      const uint32_t klo = (len == 0 || mn == 0u) ? 0u : sum;       // a bin-0 column (or an empty document) voids the lower bound
      const uint32_t khi = (mx == 255u) ? 0xFFFFu : sum;            // a bin-255 column voids the upper bound
      kq[beg + i] = (khi << 16) | klo;
    }
    u0 = nu0;
    len = nlen;
  }
}

// two 32-column chunks (32 < q_len <= 64): the candidate's bounds are the sums of the chunks' bounds; a void bound in either
// chunk voids the sum
__global__ void k_q8_combine(const uint32_t* __restrict__ part, int nch, int64_t M, uint32_t* __restrict__ kq) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t lo = 0, hi = 0;
    bool lo_void = false, hi_void = false;
    for (int c = 0; c < nch; ++c) {
      const uint32_t v = part[(int64_t)c * M + i];
      const uint32_t l = v & 0xFFFFu, h = v >> 16;
      lo_void |= (l == 0u);
      hi_void |= (h == 0xFFFFu);
      lo += l;
      hi += h;
    }
    kq[i] = ((hi_void ? 0xFFFFu : hi) << 16) | (lo_void ? 0u : lo);
  }
}

// histogram of K_lo per query: Q8_HIST_BLOCKS workgroups per query build LDS histograms of their share
// and flush the non-empty bins into hist [B][bins] (zeroed by the caller); bins = Q8_BINS_SUM * chunks
#define Q8_HIST_BLOCKS 16
__global__ __launch_bounds__(1024) void k_q8_hist(const uint32_t* __restrict__ kq, const int64_t* __restrict__ cand_off,
                                                  uint32_t* __restrict__ hist, int bins) {
  extern __shared__ uint32_t h_dyn[];
  uint32_t* h = h_dyn;
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int64_t beg = cand_off[b];
  const int64_t n = cand_off[b + 1] - beg;
  for (int i = tid; i < bins; i += 1024) h[i] = 0u;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 1024 + tid; i < n; i += (int64_t)Q8_HIST_BLOCKS * 1024) atomicAdd(&h[kq[beg + i] & 0xFFFFu], 1u);
  __syncthreads();
  uint32_t* hg = hist + (int64_t)b * bins;
  for (int i = tid; i < bins; i += 1024) {
    const uint32_t v = h[i];
    if (v) atomicAdd(&hg[i], v);
  }
}

// one workgroup per query: T = keep-th largest K_lo from the histogram, cut = max(T - Q + 1, 0)
__global__ __launch_bounds__(1024) void k_q8_cut(const uint32_t* __restrict__ hist, const int64_t* __restrict__ cand_off, int64_t n_full,
                                                 int64_t R, int Q, int32_t* __restrict__ cut, int bins) {
  __shared__ uint32_t part[1024];
  __shared__ int s_T;
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int64_t n = cand_off[b + 1] - cand_off[b];
  int64_t keep = n;
  if (n_full < keep) keep = n_full;
  if (R < keep) keep = R;
  if (keep >= n) {  // nothing is pruned: everything survives
    if (tid == 0) cut[b] = 0;
    return;
  }
  if (tid == 0) s_T = 0;
  const uint32_t* hg = hist + (int64_t)b * bins;
  // thread t owns bins [bpt*t, bpt*(t+1)); suffix counts from the top
  const int bpt = bins / 1024;   // 8 or 16
  uint32_t own[16];
  uint32_t loc = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    own[k] = (k < bpt) ? hg[tid * bpt + k] : 0u;
    loc += own[k];
  }
  part[tid] = loc;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // inclusive suffix scan
    const uint32_t v = (tid + off < 1024) ? part[tid + off] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  const uint32_t above = part[tid] - loc;  // candidates in bins above this thread's bins
  if (above < (uint32_t)keep && part[tid] >= (uint32_t)keep) {
    uint32_t acc = above;
    int T = tid * bpt;
#pragma unroll
    for (int k = 15; k >= 0; --k) {
      if (k < bpt) {
        acc += own[k];
        if (acc >= (uint32_t)keep) { T = tid * bpt + k; break; }
      }
    }
    s_T = T;
  }
  __syncthreads();
  if (tid == 0) {
    const int c = s_T - Q + 1;
    cut[b] = (s_T <= 0 || c < 0) ? 0 : c;
  }
}

// ordered compaction of the survivors (K_hi >= cut): count per 2048-candidate chunk, then write
#define SURV_CHUNK 2048
__global__ __launch_bounds__(256) void k_surv_count(const uint32_t* __restrict__ kq, const int64_t* __restrict__ cand_off,
                                                    const int32_t* __restrict__ cut, int32_t* __restrict__ blkcnt, int nblk) {
  const int b = blockIdx.y;
  const int64_t beg = cand_off[b];
  const int64_t n = cand_off[b + 1] - beg;
  const int64_t c0 = (int64_t)blockIdx.x * SURV_CHUNK;
  int cnt = 0;
  if (c0 < n) {
    const uint32_t ct = (uint32_t)cut[b];
#pragma unroll
    for (int k = 0; k < SURV_CHUNK / 256; ++k) {
      const int64_t i = c0 + threadIdx.x * (SURV_CHUNK / 256) + k;
      if (i < n) cnt += ((kq[beg + i] >> 16) >= ct) ? 1 : 0;
    }
  }
  __shared__ int s[256];
  s[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) blkcnt[(int64_t)b * nblk + blockIdx.x] = s[0];
}

__global__ __launch_bounds__(256) void k_surv_compact(const uint32_t* __restrict__ kq, const int64_t* __restrict__ cand_off,
                                                      const int32_t* __restrict__ cand_pid, const int32_t* __restrict__ cut,
                                                      const int32_t* __restrict__ blkoff, int nblk,
                                                      const int64_t* __restrict__ surv_off, int32_t* __restrict__ surv_pid) {
  const int b = blockIdx.y;
  const int64_t beg = cand_off[b];
  const int64_t n = cand_off[b + 1] - beg;
  const int64_t c0 = (int64_t)blockIdx.x * SURV_CHUNK;
  if (c0 >= n) return;
  const uint32_t ct = (uint32_t)cut[b];
  constexpr int PER = SURV_CHUNK / 256;
  bool keep[PER];
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int64_t i = c0 + threadIdx.x * PER + k;
    keep[k] = (i < n) && ((kq[beg + i] >> 16) >= ct);
    cnt += keep[k] ? 1 : 0;
  }
  __shared__ int s[256];
  s[threadIdx.x] = cnt;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int t = ((int)threadIdx.x >= off) ? s[threadIdx.x - off] : 0;
    __syncthreads();
    s[threadIdx.x] += t;
    __syncthreads();
  }
  int64_t pos = surv_off[b] + blkoff[(int64_t)b * nblk + blockIdx.x] + (s[threadIdx.x] - cnt);
  if (!cnt) return;
  int32_t pid[PER];   // the gathers first, then the stores (interleaved, every survivor costs a round trip of its own: k_l0_compact)
#pragma unroll
  for (int k = 0; k < PER; ++k) pid[k] = keep[k] ? cand_pid[beg + c0 + threadIdx.x * PER + k] : 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    if (keep[k]) surv_pid[pos++] = pid[k];
  }
}

// (2) of the scheme: kq [M] (behind it, for two chunks, the per-chunk partial sums) <- bounds of every candidate
void fpk_approx_q8_bounds(const FpIndexDev& ix, const uint8_t* S8, const FpSearchShape& sh, const int64_t* cand_off, const int32_t* cand_pid,
                          int64_t M, uint32_t* kq /*[M * (1 + (Qp > 32 ? Qp/32 : 0))]*/, hipStream_t st) {
  const int B = sh.B;
  // Lane pairs per candidate: about 2-3 code groups per pair (cfg2, 33 codes = 5 groups: 1/2/4/8 pairs -> S4 5.15/4.34/
  // 4.53/4.63 ms), more when one query would otherwise not fill the chip (~1280 resident workgroups x 128 pairs; long
  // documents, small shards).  One workgroup pass per query wherever possible: a grid cap of 2048 costs 0.5 ms, and
  // longer-lived workgroups (2/4/8 candidates per pair: 6.6/8.3/10.3 ms at 1 pair) put several queries' slices in flight.
  static const int ppd_env = (int)fp_test_opt("q8_ppd", 0);
  const int64_t per_q = (M + B - 1) / B;
  const int64_t groups = ix.N > 0 ? (ix.U / ix.N + 7) / 8 : 1;   // average code groups per document
  int ppd = 1;
  while (ppd < 8 && ppd * 4 <= groups) ppd <<= 1;                // >= 2 groups per pair
  while (ppd < 8 && per_q * ppd < 131072) ppd <<= 1;             // fill the chip with one query
  if (ppd_env == 1 || ppd_env == 2 || ppd_env == 4 || ppd_env == 8) ppd = ppd_env;
  const int nch = sh.Qp / 32;                            // 1 or 2 chunks of 32 query columns
  uint32_t* kq_part = nch > 1 ? kq + M : kq;             // per-chunk partial sums behind the final ones
  const int tpb = 256;   // 64 / 128 / 512 threads per workgroup measured the same (4.34-4.41 ms)
  const int dpb = (tpb / 2) / ppd;
  int64_t bx = (per_q + dpb - 1) / dpb;
  const int64_t cap = 65535;   // grid.x limit; caps of 2048 / 4096 cost 0.5 / 0.1 ms at cfg2
  if (bx > cap) bx = cap;
  if (bx < 1) bx = 1;
  // (an XCD-affine mapping -- one query per XCD so that each L2 holds one 4 MB slice -- measured slower: 6.9 vs 6.05 ms;
  //  one launch per 1/2/4 queries: 6.2/6.0/5.9 vs 5.8 ms)
#define FP_Q8_LAUNCH(PPD_)                                                                                                        \
  hipLaunchKernelGGL(k_approx_q8<PPD_>, dim3((unsigned)bx, (unsigned)(B * nch)), dim3(tpb), 0, st, S8, ix.C, sh.Q, cand_off, cand_pid, \
                     ix.uoff, ix.ucodes, kq_part, nch, M)
  switch (ppd) {
    case 1: FP_Q8_LAUNCH(1); break;
    case 2: FP_Q8_LAUNCH(2); break;
    case 8: FP_Q8_LAUNCH(8); break;
    default: FP_Q8_LAUNCH(4); break;
  }
#undef FP_Q8_LAUNCH
}

// (3)-(4): per-query cut from the histogram of the lower bounds, ordered survivors in surv_off [B+1] / surv_pid
void fpk_approx_q8_cut(const FpSearchShape& sh, const int64_t* cand_off, const int32_t* cand_pid, int64_t M, uint32_t* q8hist /*[B][8192 * Qp/32]*/,
                       uint32_t* kq, int32_t* cut, int32_t* blkcnt, int nblk, int32_t* nsurv, int64_t* surv_off, int32_t* surv_pid,
                       hipStream_t st, bool lazy_bins) {
  const int B = sh.B;
  const int nch = sh.Qp / 32;
  const int bins = Q8_BINS_SUM * nch;
  if (nch > 1) hipLaunchKernelGGL(k_q8_combine, dim3(fp_grid_cap((M + 255) / 256, 256)), dim3(256), 0, st, kq + M, nch, M, kq);
  (void)hipMemsetAsync(q8hist, 0, (size_t)B * bins * 4, st);
  static std::atomic<uint64_t> lds_ok{0};
  fp_allow_big_lds((const void*)k_q8_hist, lds_ok, 144 * 1024);
  hipLaunchKernelGGL(k_q8_hist, dim3(Q8_HIST_BLOCKS, (unsigned)B), dim3(1024), (size_t)bins * 4, st, kq, cand_off, q8hist, bins);
  // (S1's lazy form: the bins come from upper candidates, at most one bin above the true one per column (an fp16 step is below
  // 1/128 wherever a bin is not clamped), so a document's TRUE lower bound is K_lo - Q at worst: the cut moves down by another Q)
  hipLaunchKernelGGL(k_q8_cut, dim3((unsigned)B), dim3(1024), 0, st, q8hist, cand_off, sh.n_full, sh.R, lazy_bins ? 2 * sh.Q : sh.Q, cut, bins);
  hipLaunchKernelGGL(k_surv_count, dim3((unsigned)nblk, (unsigned)B), dim3(256), 0, st, kq, cand_off, cut, blkcnt, nblk);
  hipLaunchKernelGGL(k_cand_scan, dim3((unsigned)B), dim3(256), 0, st, blkcnt, nblk, nsurv, (int32_t*)nullptr, (int32_t*)nullptr, (uint32_t*)nullptr,
                     0, (int64_t*)nullptr, (int64_t)0, (int32_t*)nullptr, (int64_t*)nullptr, (const int32_t*)nullptr);
  hipLaunchKernelGGL(k_cand_offsets, dim3(1), dim3(256), 0, st, nsurv, B, surv_off, (int64_t)0, (int32_t*)nullptr, (int64_t*)nullptr,
                     (const int32_t*)nullptr);
  hipLaunchKernelGGL(k_surv_compact, dim3((unsigned)nblk, (unsigned)B), dim3(256), 0, st, kq, cand_off, cand_pid, cut, blkcnt, nblk, surv_off,
                     surv_pid);
}

// ============================================================================================
// S4, level 0: one scalar per centroid, held in LDS.  The bound stage above still gathers one 32-byte row per (candidate,
// unique code) from a table that overflows an XCD's L2; this stage needs NO row gathers at all.
//   f_q      = a per-column floor (bin units), here the (1 - tail)-quantile of column q over a sample of the centroids
//   e(c)     = sum_q max(0, S8[c][q] - f_q)                        "excess" of centroid c over the floors
//   UB0(d)   = F + sum_{c in codes(d)} e(c),  F = sum_q f_q
// Since max_c S8[c][q] <= f_q + sum_c max(0, S8[c][q] - f_q), UB0(d) >= K(d) = sum_q max_c S8[c][q] for every document, and
// with 128 A + 100 Q < K + Q (see above; a bin-255 entry makes e(c) infinite) a document can only reach an exact
// approximate score A >= A_T if UB0(d) > 128 A_T + 99 Q.  e(c) is a BYTE table of C entries (128 KiB at C = 2^17) that
// every workgroup copies into LDS; a candidate then costs its code line (128 B) and one LDS byte per code.  The table is
// written by S1's epilogue (FpS1Excess; k_l0_table builds it from a full 8-bit table in the fallback form).
//   A_T: the top-R' documents by UB0 (R' = 4 R) are scored exactly; A_T = the keep-th largest of THEIR exact scores, which
//        is <= the keep-th largest exact score overall.  Survivors = { UB0 >= floor(128 A_T + 99 Q) + 1 }: they contain
//        every document with A >= A_T, hence the exact top-keep with all its ties -> the selection that follows is
//        identical to scoring every candidate exactly.  On the benchmark corpus 1.9 % of the candidates are scored exactly
//        (pilot group + survivors outside it; tools/sim_s4_bounds.py reproduces the bound on the CPU).
// ============================================================================================
// byte code of an excess e:  0..159 = e itself;  160..239 = 160 + ceil((e - 160) / 8), decoded as 160 + 8 (v - 160) >= e (at
// most 7 looser; these are the query's own topic centroids, e = 200..600, one or two per document -- as escapes they cost a
// second pass over a third of the candidates, and a table in units of 4 loosened EVERY code by up to 3, which doubled the
// survivors);  240..254 = escape slot 0..14 (u32 values in LDS) for e >= 800;  255 = infinite.
// decode of v <= 239:  max(v, 8 v - 1120)
#define L0_INF 0xFFFFu
#define L0_SAMPLE 8192          // centroids sampled for the column quantiles (from a full 8-bit table)
#define L0_SAMPLE_PRE 2048      // ... by S1's sampled pre-pass (FP_L0_SAMPLE; 8192 / 4096 / 2048 / 1024 rows: S1 stage 0.379 / 0.354 / 0.345 / 0.341 ms, same survivors)

// floors[b][Qp] u8 (pad columns 0), Fsum[b]; one workgroup per query
__global__ __launch_bounds__(1024) void k_l0_floor(const uint8_t* __restrict__ S8, int64_t C, int Q, int nch, float tail,
                                                   uint8_t* __restrict__ floors, uint32_t* __restrict__ Fsum, uint32_t* __restrict__ esc,
                                                   half_t* __restrict__ gfl /*nullable*/, int gfl_rd = 0) {
  extern __shared__ uint32_t l0h[];   // [nch*32][257]: a column's bins, rows padded by one word so that the per-column scans below hit 32 banks
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < 64) esc[(int64_t)b * 64 + tid] = 0u;   // the query's escape slots (filled by whoever builds the table next)
  const int ncol = nch * 32;
  for (int i = tid; i < ncol * 257; i += 1024) l0h[i] = 0u;
  __syncthreads();
  const int64_t stride = C > L0_SAMPLE ? C / L0_SAMPLE : 1;
  const int64_t ns = (C + stride - 1) / stride;
  // (sample, chunk, 16-byte half) items, four per thread and round: the four loads are issued before any of their atomics
  // (one load per loop trip left the kernel waiting on memory latency 16 times: 34 us)
  const int64_t items = ns * nch * 2;
  const int rot = tid & 15;
  for (int64_t i0 = tid; i0 < items; i0 += 4 * 1024) {
    uint4 vv[4];
    int colbase[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + (int64_t)u * 1024;
      vv[u] = make_uint4(0, 0, 0, 0);
      colbase[u] = -1;
      if (i < items) {
        const int half = (int)(i & 1);
        const int ch = (int)((i >> 1) % nch);
        const int64_t c = ((i >> 1) / nch) * stride;
        vv[u] = *reinterpret_cast<const uint4*>(S8 + (((int64_t)b * nch + ch) * C + c) * 32 + half * 16);
        colbase[u] = ch * 32 + half * 16;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (colbase[u] < 0) continue;
      const uint4 v = vv[u];
      // the lanes of a wave hold different samples of the SAME 16 columns, and a column's bins cluster around its mode: a
      // lane-rotated column order spreads a wave's simultaneous atomics over 16 histograms instead of one
#pragma unroll
      for (int j0 = 0; j0 < 16; ++j0) {
        const int j = (j0 + rot) & 15;
        const uint32_t word = (j & 8) ? ((j & 4) ? v.w : v.z) : ((j & 4) ? v.y : v.x);
        atomicAdd(&l0h[(colbase[u] + j) * 257 + ((word >> (8 * (j & 3))) & 0xFFu)], 1u);
      }
    }
  }
  __syncthreads();
  __shared__ uint32_t fl[128];
  // f = the smallest bin such that at most tail * ns samples lie above it = the largest k >= 1 whose suffix count
  // #{bin >= k} exceeds lim (0 if none).  One wave per column: lane l holds bins 4 l .. 4 l + 3, the suffix counts over the
  // lanes come from six shuffles (a thread per column walking down from bin 255 was ~120 dependent LDS reads: 5 of the
  // kernel's 16 us, at every batch size)
  const uint32_t lim = (uint32_t)(tail * (float)ns);
  const int lane = tid & 63;
  for (int col = tid >> 6; col < ncol; col += 16) {   // (wave-uniform)
    uint32_t f = 0;
    if (col < Q) {
      const uint32_t* h = &l0h[col * 257 + 4 * lane];
      const uint32_t b0 = h[0], b1 = h[1], b2 = h[2], b3 = h[3];
      const uint32_t own = b0 + b1 + b2 + b3;
      uint32_t suf = own;   // bins >= 4 lane
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_down(suf, d, 64);
        if (lane + d < 64) suf += o;
      }
      const unsigned long long bal = __ballot(suf > lim);   // (suf falls with the lane: the set bits are lanes 0 .. L)
      if (bal) {
        const int L = 63 - __builtin_clzll(bal);
        uint32_t above = suf - own;   // bins >= 4 (lane + 1)
        int k = 4 * lane;
        if (above + b3 > lim) k = 4 * lane + 3;
        else if (above + b3 + b2 > lim) k = 4 * lane + 2;
        else if (above + b3 + b2 + b1 > lim) k = 4 * lane + 1;
        f = (uint32_t)__shfl(k, L, 64);
      }
    }
    if (lane == 0) {
      fl[col] = f;
      floors[(int64_t)b * ncol + col] = (uint8_t)f;
      if (gfl) {   // S1's epilogue form of the floors: g = f - 100 (pads 2000), or 1024 - g for its one-fma form (FpS1Excess::rd)
        const float g = col < Q ? (float)f - 100.f : 2000.f;
        gfl[(int64_t)b * ncol + col] = (half_t)(gfl_rd ? 1024.f - g : g);
      }
    }
  }
  __syncthreads();
  if (tid < 64) {
    uint32_t s = (tid < ncol ? fl[tid] : 0u) + (tid + 64 < ncol ? fl[tid + 64] : 0u);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if (tid == 0) Fsum[b] = s;
  }
}

// e8[b][c]; esc[b][64]: [0..62] escape values, [63] = slots handed out
__global__ __launch_bounds__(256) void k_l0_table(const uint8_t* __restrict__ S8, int64_t C, int64_t Cpad, int Q, int nch,
                                                  const uint8_t* __restrict__ floors, uint8_t* __restrict__ e8, uint32_t* __restrict__ esc) {
  const int b = blockIdx.y;
  __shared__ uint8_t fl[128];
  if (threadIdx.x < nch * 32) fl[threadIdx.x] = floors[(int64_t)b * nch * 32 + threadIdx.x];
  __syncthreads();
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= Cpad) return;
  uint32_t out = 0;
  if (c < C) {
    uint32_t e = 0;
    bool inf = false;
    for (int ch = 0; ch < nch; ++ch) {
      const uint4* row = reinterpret_cast<const uint4*>(S8 + (((int64_t)b * nch + ch) * C + c) * 32);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint4 v = row[h];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int q = ch * 32 + h * 16 + j;
          const uint32_t x = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
          if (q < Q) {
            inf |= (x == 255u);   // clamped bin: no upper bound on the score behind it
            const uint32_t f = fl[q];
            e += x > f ? x - f : 0u;
          }
        }
      }
    }
    out = l0_encode(e, inf, esc + (int64_t)b * 64);
  }
  e8[(int64_t)b * Cpad + c] = (uint8_t)out;
}

// ub[cand] = min(F + sum e(c), 0xFFFE), or 0xFFFF when some code has an infinite excess.
// The candidates' codes come from the packed lines (fp_synth.hip "packed unique codes"): one aligned 128-byte line per document,
// line <document id> (documents with > 48 distinct codes: extra lines behind the first lines, found through a side table on
// the rare path), 8 lanes per line, one 16-byte piece = 6 codes per lane, nothing shared between lanes but the final sum.
// A lane group takes FOUR consecutive candidates per iteration; the line loads of the next iteration and the ids of the two
// after that are in flight while it computes.  (Until round 3 a {first line, line count} lookup sat between the id and the
// line: 8 B more per candidate and one more dependent load in the chain.)  Escape
// slots / infinite entries take a rare second pass over the candidate.  The histogram of (ub - F) >> 2 that the pilot cut
// needs is built here in LDS.
#define L0_HBINS 4096
#define L0_UNROLL 4
#define L0_PILOT_MAX 98304    // capacity of a query's pilot group
// the bounds of query b live at ub[l0_row(cand_off, b) + i], i = position in the query's candidate list: rows start on 16-byte
// boundaries (ub holds M + 8 B + 16 entries) so that the passes over them use 16-byte loads
__device__ __forceinline__ int64_t l0_row(const int64_t* __restrict__ cand_off, int b) { return ((cand_off[b] + 7) & ~(int64_t)7) + 8 * (int64_t)b; }
template <int LPC>
__device__ __forceinline__ uint32_t l0_red(uint32_t v) {   // sum over the LPC (8 or 4) lanes of a group, in all of them (no lane outside the group is read)
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  if constexpr (LPC == 8) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror: the other quad of the group
  return v;
}
template <int LPC>
__device__ __forceinline__ uint32_t l0_maxg(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));
  if constexpr (LPC == 8) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));
  return v;
}
// Tables beyond 2^17 centroids are walked in ranges of 2^17 (one launch each, the range's slice of the table in LDS and the
// index's per-range code lines): the first launch writes F + its sum, the later ones add theirs (saturating, infinite stays
// infinite), the last one builds the histogram from the final values.
typedef __attribute__((address_space(3))) const uint8_t l0_lds_u8;
__device__ __forceinline__ uint32_t l0_u16(const uint4& v, int k) {
  const uint32_t w = k < 2 ? v.x : (k < 4 ? v.y : (k < 6 ? v.z : v.w));
  return (k & 1) ? (w >> 16) : (w & 0xFFFFu);
}
#define FP_L0_RANGE_C (1ll << 17)   // == FP_L0_RANGE (defined with the launchers below)
// LPC = lanes per candidate = 16-byte pieces per code line: 8 (128-byte lines, 48 codes: one line per document for tables of one
// range) or 4 (64-byte lines, 24 codes: tables of several ranges, where a document has ~8 codes per range and 8 lanes per
// candidate would spend three quarters of their lookups on empty pieces).
template <int LPC>
__global__ __launch_bounds__(1024) void k_l0_scan(const uint8_t* __restrict__ e8, int64_t Cpad, int64_t tab_off, int tab_bytes,
                                                  const uint32_t* __restrict__ esc,
                                                  const uint32_t* __restrict__ Fsum, const int64_t* __restrict__ cand_off,
                                                  const int32_t* __restrict__ cand_pid, const int32_t* __restrict__ poff,
                                                  const uint4* __restrict__ pcodes, uint16_t* __restrict__ ub, uint32_t* __restrict__ hist,
                                                  int first, int last, int bxn, int xcd_affine, FpL0Multi mr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char l0s[];
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)l0s != 0u) __builtin_trap();   // the table must sit at LDS address 0
  // xcd_affine == 2: ALL RANGES OF A MULTI-RANGE TABLE IN ONE LAUNCH (see fpk_l0_scan): workgroup = (query, chunk, range r); it
  // takes range r's slice of the table, reads line <document id> * nr + r of the interleaved first lines and writes its
  // partial sums (no floor sum, no histogram) to part r of `ub`; k_l0_combine adds the parts up.
  const bool multi = xcd_affine == 2;
  const bool inter = mr.nr > 1;          // interleaved first lines (tables of several ranges)
  int rng = inter ? mr.seq_r : 0;        // (one launch per range, FP_L0_MULTI=0: the host names the range)
  if (multi) {
    const int W = ((bxn + 7) & ~7) * mr.nr;           // workgroups per query
    const int l = (int)(blockIdx.x % (unsigned)W);
    rng = (l % (8 * mr.nr)) / 8;
    tab_off = (int64_t)rng * FP_L0_RANGE_C;
    tab_bytes = (int)(Cpad - tab_off < FP_L0_RANGE_C ? Cpad - tab_off : FP_L0_RANGE_C);
    ub += (int64_t)rng * mr.ub_stride;
  }
  if (inter) poff = mr.po[rng];
  const uint4* __restrict__ xcodes = inter ? mr.x[rng] : pcodes;   // extra lines: the range's own array, or behind the first lines
  const int pstr = inter ? mr.nr : 1;
  // pair mode (LPC == 8 over the 64-byte lines of a multi-range table): the 8 lanes of a candidate load the 128-byte PAIR of
  // lines of ranges (2p, 2p + 1) in one request; only the half that belongs to this workgroup's range counts.  The partner
  // range's workgroup issues the same request moments later and finds the line in L2.
  const bool pair = multi && mr.pair != 0 && LPC == 8;
  const int xlp = pair ? 4 : LPC;            // pieces per extra line
  uint8_t* tab = l0s;                                                  // [tab_bytes + 16]
  uint32_t* es = reinterpret_cast<uint32_t*>(l0s + tab_bytes + 16);    // [64]
  uint32_t* hl = es + 64;                                              // [L0_HBINS]
  // XCD affinity (experiment, off by default: slower, see fpk_l0_scan): workgroups are dealt to the 8 XCDs round-robin by
  // linear id, and every workgroup starts by copying its query's table (up to 128 KiB) out of L2.  With the plain (chunk,
  // query) grid a query's workgroups land on all 8 XCDs and every L2 sees all B tables; with xcd_affine query b is served by
  // XCD b % 8 only.  bxn = chunks per query; needs B % 8 == 0.
  const int tid = threadIdx.x;
  int b, bxi;
  if (multi) {
    const int W = ((bxn + 7) & ~7) * mr.nr;
    const int l = (int)(blockIdx.x % (unsigned)W);
    b = (int)(blockIdx.x / (unsigned)W);
    bxi = (l / (8 * mr.nr)) * 8 + (l & 7);
    if (bxi >= bxn) return;
  } else if (xcd_affine) {
    const int lin = blockIdx.x;                 // 1-D launch of bxn * B workgroups
    const int xcd = lin & 7, slot = lin >> 3;   // slot-th workgroup of this XCD
    b = (slot / bxn) * 8 + xcd;
    bxi = slot % bxn;
  } else {
    b = blockIdx.y;
    bxi = blockIdx.x;
  }
  const int64_t beg = cand_off[b];
  const int64_t n = cand_off[b + 1] - beg;
  constexpr int CPI = (1024 / LPC) * L0_UNROLL;   // candidates per workgroup iteration
  if ((int64_t)bxi * CPI >= n) return;
  const uint32_t F = multi ? 0u : Fsum[b];
  const int sub = tid & (LPC - 1);
  const bool active = !pair || (sub >> 2) == (rng & 1);   // pair mode: this lane holds a piece of the workgroup's own range
  const int kq = sub & 3;                      // the candidate of the group's four whose id this lane fetches
  const int64_t stride = (int64_t)bxn * CPI;
  const int32_t* cpids = cand_pid + beg;
  const uint2* pmeta = reinterpret_cast<const uint2*>(poff);
  uint16_t* ubrow = ub + l0_row(cand_off, b);
  int64_t i = (int64_t)bxi * CPI + (tid / LPC) * L0_UNROLL;
  // Software pipeline over the dependent chain  id -> code line:  while iteration t is computed, the code lines of t+1 and the
  // ids of t+2 / t+3 are in flight (one workgroup per CU -- the table takes the LDS -- leaves only 4 waves per SIMD to hide
  // latency otherwise).  Ids are fetched cooperatively: each lane of a quad loads ONE of the group's four candidates and the
  // values are exchanged by quad broadcasts -- 1 load instruction per iteration instead of 4 through the CU's one address unit.
  auto qb = [](uint32_t v, int k) -> uint32_t {   // value of lane k of this lane's quad
    switch (k) {
      case 0: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xF, 0xF, true);
      case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x55, 0xF, 0xF, true);
      case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xAA, 0xF, 0xF, true);
      default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xFF, 0xF, 0xF, true);
    }
  };
  // The loads are unconditional: positions past the end of the list are clamped to its last entry (their results are never
  // stored); a document without codes owns an empty line (all counts 0: it sums to nothing) -- guards cost a branch, an
  // exec-mask save and four zeroing moves per load otherwise.
  const int64_t nlast = n - 1;
  auto pos = [&](int64_t at) -> int64_t { return at + kq < nlast ? at + kq : nlast; };
  auto load_line = [&](uint32_t line) -> uint4 {
    if (pair) return pcodes[((int64_t)line * pstr + (rng & ~1)) * 4 + sub];
    return pcodes[((int64_t)line * pstr + rng) * LPC + sub];
  };
  // (a document's first line is line <document id>: no lookup between the id and the line)
  int32_t p0 = cpids[pos(i)];                          // iteration t
  int32_t p1 = cpids[pos(i + stride)];                 // t+1
  int32_t p2 = cpids[pos(i + 2 * stride)];             // t+2
  uint4 pcA[L0_UNROLL], pcB[L0_UNROLL];                // first code lines of t / t+1, roles swapped every iteration
#pragma unroll
  for (int k = 0; k < L0_UNROLL; ++k) pcA[k] = load_line(qb((uint32_t)p0, k));
  // the query's table -> LDS, behind the pipeline's first loads (nothing above touches LDS): the copy's latency overlaps theirs
  {
    const uint4* src = reinterpret_cast<const uint4*>(e8 + (int64_t)b * Cpad + tab_off);
    uint4* dst = reinterpret_cast<uint4*>(tab);
    for (int j = tid; j < tab_bytes / 16; j += 1024) dst[j] = src[j];
    if (tid < 4) reinterpret_cast<uint32_t*>(tab + tab_bytes)[tid] = 0u;
    if (tid < 64) es[tid] = esc[(int64_t)b * 64 + tid];
    for (int j = tid; j < L0_HBINS; j += 1024) hl[j] = 0u;
  }
  __syncthreads();
  // sum of the 6 table entries of one piece (6 codes of 20 bits + the count in the top byte; 48 codes per line: 0.15 % of the
  // benchmark corpus' documents need a second line, against 6 % with 40 -- the second line is not prefetched).  Slots past the
  // piece's count repeat its last code (fp_synth.hip): all six are summed and (6 - count) x the last taken off again (an empty
  // piece: 6 x code 0 - 6 x code 0).  Bit 24 of the result flags an escape / infinite code (v >= 240) among the real ones.
  // The table sits at LDS address 0 (checked at kernel entry), so a code IS its ds_read address.
  auto codes6 = [](const uint4& pc, uint32_t (&c)[6]) {
    c[0] = pc.x & 0xFFFFFu;
    c[1] = __builtin_amdgcn_alignbit(pc.y, pc.x, 20) & 0xFFFFFu;
    c[2] = (pc.y >> 8) & 0xFFFFFu;
    c[3] = __builtin_amdgcn_alignbit(pc.z, pc.y, 28) & 0xFFFFFu;
    c[4] = __builtin_amdgcn_alignbit(pc.w, pc.z, 16) & 0xFFFFFu;
    c[5] = (pc.w >> 4) & 0xFFFFFu;
  };
  auto piece = [&](const uint4& pc) -> uint32_t {
    // (slots past the piece's count point at a zero byte behind the table slice: all six are summed as they are; bit 31 of the
    // word: "the document has extra lines", first line's last piece only)
    uint32_t c[6], v[6];
    codes6(pc, c);
#pragma unroll
    for (int j = 0; j < 6; ++j) v[j] = *reinterpret_cast<l0_lds_u8*>((uintptr_t)c[j]);
    uint32_t s6 = 0, m6 = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      s6 += (uint32_t)max((int)v[j], 8 * (int)v[j] - 8 * L0_LIN + L0_LIN);
      m6 = max(m6, v[j]);
    }
    return s6 + (m6 >= L0_ESC_BASE ? (1u << 24) : 0u) + ((pc.w >> 31) << 30);   // bit 30: "extra lines" (set in one piece of a first line at most)
  };
  auto body = [&](uint4 (&pc)[L0_UNROLL], uint4 (&pcn)[L0_UNROLL]) {
    // issue: lines of t+1, ids of t+3
#pragma unroll
    for (int k = 0; k < L0_UNROLL; ++k) pcn[k] = load_line(qb((uint32_t)p1, k));
    const int32_t p3 = cpids[pos(i + 3 * stride)];
    uint32_t outv[L0_UNROLL], sums[L0_UNROLL];
    uint16_t prev[L0_UNROLL] = {0, 0, 0, 0};   // the bounds so far (later ranges of a table beyond 2^17 centroids)
    if (!first) {
#pragma unroll
      for (int k = 0; k < L0_UNROLL; ++k)
        if (i + k < n) prev[k] = ubrow[i + k];
    }
#pragma unroll
    for (int k = 0; k < L0_UNROLL; ++k) sums[k] = piece(pc[k]);   // straight-line: all 24 table reads of the lane in flight together
    if (pair) {
#pragma unroll
      for (int k = 0; k < L0_UNROLL; ++k) sums[k] = active ? sums[k] : 0u;
    }
    uint32_t red[L0_UNROLL];
#pragma unroll
    for (int k = 0; k < L0_UNROLL; ++k) red[k] = l0_red<LPC>(sums[k]);
    // The common case -- none of the group's four candidates has an escaped code or extra lines (bits 24.. clear) -- is finished
    // "transposed": lane j of the group completes candidate j & 3 alone (floor sum, saturation, histogram bin) and the four
    // results return to lane 0 by quad broadcasts.  Every lane finishing all four candidates was 4 x ~20 vector instructions per
    // iteration beside the 4 x 32 of the lookups themselves.
    const bool plain4 = ((red[0] | red[1] | red[2] | red[3]) >> 24) == 0u;
    if (plain4) {
      const int kk = sub & 3;
      const uint32_t mys = kk == 0 ? red[0] : (kk == 1 ? red[1] : (kk == 2 ? red[2] : red[3]));
      const uint32_t myp = kk == 0 ? (uint32_t)prev[0] : (kk == 1 ? (uint32_t)prev[1] : (kk == 2 ? (uint32_t)prev[2] : (uint32_t)prev[3]));
      uint32_t v = (first ? F : myp) + mys;
      v = (!first && myp == L0_INF) ? L0_INF : (v > 0xFFFEu ? 0xFFFEu : v);
      if (last && sub < 4 && i + kk < n) {
        const uint32_t hb = (v >= F ? v - F : 0u) >> 2;
        atomicAdd(&hl[hb > L0_HBINS - 1 ? L0_HBINS - 1 : hb], 1u);
      }
#pragma unroll
      for (int k = 0; k < L0_UNROLL; ++k) outv[k] = qb(v, k);
    } else {
#pragma unroll
    for (int k = 0; k < L0_UNROLL; ++k) {
      // bits 24.. of the sum count the pieces that hold an escaped / infinite code (at most 8 in a first line); bit 30 says that
      // the document has extra lines (more than one line's worth of distinct codes in this range): rare, {first, count} then
      // come from the side table.  The group branches as one (the reduced values are the same in all its lanes).
      uint32_t sum = red[k];
      uint32_t xl0 = 0, nx = 0;
      if (sum & (1u << 30)) {
        sum -= 1u << 30;
        const uint2 mx = pmeta[qb((uint32_t)p0, k)];
        xl0 = mx.x;
        nx = mx.y;
        uint32_t sx = 0;
        for (uint32_t t = 0; t < nx; ++t) {
          const uint32_t px = piece(xcodes[((int64_t)xl0 + t) * xlp + (sub & (xlp - 1))]);
          sx += active ? px : 0u;
        }
        sum += l0_red<LPC>(sx);
      }
      if (sum >> 24) {
        // some code of this candidate has an escaped (e >= 800) or infinite excess: a second pass that takes the escape values
        // from LDS; the first line is still in registers.
        uint32_t s2 = 0, inf = 0;
        for (uint32_t t = 0; t <= nx; ++t) {
          const uint4 q = t == 0 ? pc[k] : xcodes[((int64_t)xl0 + t - 1) * xlp + (sub & (xlp - 1))];
          uint32_t c[6];
          codes6(q, c);
          uint32_t s6 = 0, i6 = 0;
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const uint32_t v = tab[c[j]];
            const uint32_t ev = es[(max(v, (uint32_t)L0_ESC_BASE) - L0_ESC_BASE) & 63];
            i6 |= (v == 255u) ? 1u : 0u;
            s6 += v >= L0_ESC_BASE ? ev : (uint32_t)max((int)v, 8 * (int)v - 8 * L0_LIN + L0_LIN);
          }
          s2 += active ? s6 : 0u;
          inf |= active ? i6 : 0u;
        }
        sum = l0_red<LPC>(s2);
        sum = l0_maxg<LPC>(inf) ? 0xFFFFFFu : sum;
      }
      uint32_t v = (first ? F : (uint32_t)prev[k]) + sum;
      v = (sum >= 0xFFFFFFu || (!first && prev[k] == L0_INF)) ? L0_INF : (v > 0xFFFEu ? 0xFFFEu : v);
      outv[k] = v;
      if (last && sub == 0 && i + k < n) {
        const uint32_t hb = (v >= F ? v - F : 0u) >> 2;
        atomicAdd(&hl[hb > L0_HBINS - 1 ? L0_HBINS - 1 : hb], 1u);
      }
    }
    }
    if (sub == 0) {   // the group's four bounds leave as one 8-byte store when they can (2-byte stores are one fabric write each)
      if (i + L0_UNROLL <= n) {
        *reinterpret_cast<uint2*>(ubrow + i) = make_uint2(outv[0] | (outv[1] << 16), outv[2] | (outv[3] << 16));
      } else {
#pragma unroll
        for (int k = 0; k < L0_UNROLL; ++k)
          if (i + k < n) ubrow[i + k] = (uint16_t)outv[k];
      }
    }
    p0 = p1; p1 = p2; p2 = p3;
  };
  while (i < n) {
    body(pcA, pcB);
    i += stride;
    if (i >= n) break;
    body(pcB, pcA);
    i += stride;
  }
  if (!last) return;   // (uniform)
  __syncthreads();
  uint32_t* hg = hist + (int64_t)b * L0_HBINS;
  for (int i2 = tid; i2 < L0_HBINS; i2 += 1024) {
    const uint32_t v = hl[i2];
    if (v) atomicAdd(&hg[i2], v);
  }
}

// ---- level 0, "hot" form (round 4): documents with MANY distinct codes -----------------------------------------------------
// The sum of excesses above grows with the number of codes of a document -- at 300 codes (1024-token documents) 61 % of the
// candidates survived it -- because it adds what the score takes a per-column maximum of.  This form keeps the per-column
// maximum and still touches S only where it can matter:
//   a code is HOT for the query when its excess byte is non-zero, i.e. some column's bin lies above that column's floor;
//   UBh(d) = sum_q max(f_q, max over the HOT codes c of d of bin(S[c][q]))
// A code that is not hot has bin <= f_q in every column, so leaving it out cannot lower a column's term below the true maximum's
// bin: UBh(d) >= K(d), the same inequality the pilot / threshold logic of level 0 rests on -- and UBh <= UB0.  With the floors at
// a quantile matched to the documents' code count (tail = 1 / (1.2 x distinct codes per document), FP_L0H_TAIL overrides: a
// document reaches about that quantile in a typical column anyway, so the floors cost little and few codes rise above them)
// ~6 % of a document's codes are hot: the scan walks the document's unique-code list (4 B per code, streamed), looks every code
// up in the LDS byte table, and gathers the 64-byte row of S for the hot ones only -- where the 8-bit stage gathers a row for
// EVERY code.  One lane quad per candidate (8 columns of a 32-column chunk per lane, like k_approx), bins from the fp16 maxima
// at the end (bin is monotone: the maximum's bin is the bins' maximum).  Output: the same ub / histogram as k_l0_scan, so the
// pilot, threshold and survivor kernels are shared.  Tables up to 2^17 centroids (one LDS slice).
__global__ __launch_bounds__(1024) void k_l0h_scan(const uint8_t* __restrict__ e8, int64_t Cpad, int tab_bytes, const uint8_t* __restrict__ floors,
                                                   const uint32_t* __restrict__ Fsum, const int64_t* __restrict__ cand_off,
                                                   const int32_t* __restrict__ cand_pid, const int64_t* __restrict__ uoff,
                                                   const int32_t* __restrict__ ucodes, const uint16_t* __restrict__ S, int64_t C, int Q, int Qp,
                                                   uint16_t* __restrict__ ub, uint32_t* __restrict__ hist, int bxn, int warm_t) {
  // Round 6: the kernel was a chain of dependent round trips -- per 16 codes one code load, then one gather round per hot code,
  // each waited for before the next step (19 steps for a 300-code document: 0.97 ms at cfg4 for work that is ~0.1 ms of LDS
  // look-ups and ~0.3 ms of row gathers).  Now a step is 64 codes of the document (16 per lane of the quad, four 16-byte loads in
  // flight, the next step's behind them), the hot test reads ONE BIT per centroid (16 KiB of LDS instead of the 128 KiB byte table:
  // all it ever asked was "non-zero?"), and every lane gathers the WHOLE 64-byte rows of its own hot codes, two rows at a time --
  // no code travels between lanes; the quad's per-column maxima are combined once per document.  The lane's 16 codes are staged
  // in its own 64 bytes of LDS so that "my k-th code" is one LDS read instead of a 15-deep select over registers.
  // What is left is the row gathers at the fabric's request rate (~55 G rows/s: 40 M hot rows per cfg4 batch).  So the table keeps
  // FOUR bits per centroid, min(excess, 15), and a code whose excess -- the sum over the columns of its bins above the floors --
  // is below warm_t is WARM: it is not gathered, its excess is added to the bound instead (level 0's sum form for those codes:
  // column by column, max over all codes <= max over the gathered ones + sum over the warm ones).  Most hot codes are barely
  // hot, so a small warm_t drops most gathers for a few bins of looseness.
  extern __shared__ __attribute__((aligned(16))) unsigned char l0s[];
  const int bit_bytes = ((tab_bytes / 2) + 15) & ~15;
  uint8_t* nib = reinterpret_cast<uint8_t*>(l0s);                             // nibble (c & 1) of byte c >> 1: min(excess of centroid c, 15)
  uint32_t* hl = reinterpret_cast<uint32_t*>(l0s + bit_bytes);                 // [L0_HBINS]
  uint8_t* fl = reinterpret_cast<uint8_t*>(hl + L0_HBINS);                     // [Qp] (<= 256)
  int32_t* stage = reinterpret_cast<int32_t*>(l0s + bit_bytes + L0_HBINS * 4 + 256);   // [1024][16]
  const int tid = threadIdx.x;
  // Workgroup L of the launch runs on XCD L % 8.  With 8 | B the queries are dealt to the XCDs (query b on XCD b % 8), so that the
  // rows of S a query's hot codes gather -- ~9 % of its slice at cfg4, 0.7 MB -- stay in ONE 4 MiB L2 instead of every L2 holding
  // the hot rows of all the queries (22 MB: misses all the way to the fabric).
  int b = blockIdx.y, bxi = blockIdx.x;
  if ((gridDim.y & 7) == 0) {
    const unsigned L = blockIdx.y * gridDim.x + blockIdx.x, slot = L >> 3;
    b = (int)((slot / gridDim.x) * 8u + (L & 7u));
    bxi = (int)(slot % gridDim.x);
  }
  const int64_t beg = cand_off[b];
  const int64_t n = cand_off[b + 1] - beg;
  if ((int64_t)bxi * 256 >= n) return;
  {
    const uint4* src = reinterpret_cast<const uint4*>(e8 + (int64_t)b * Cpad);
    for (int j = tid; j < tab_bytes / 16; j += 1024) {
      const uint4 v = src[j];
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      uint32_t lo = 0u, hi = 0u;   // 16 bytes -> 16 nibbles
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          const uint32_t e = (w[x] >> (8 * y)) & 0xFFu, v = e > 15u ? 15u : e;
          if (x < 2) lo |= v << (4 * (4 * x + y));
          else hi |= v << (4 * (4 * (x - 2) + y));
        }
      reinterpret_cast<uint2*>(nib)[j] = make_uint2(lo, hi);
    }
    for (int j = tid; j < L0_HBINS; j += 1024) hl[j] = 0u;
    if (tid < Qp) fl[tid] = floors[(int64_t)b * Qp + tid];
  }
  __syncthreads();
  const uint32_t F = Fsum[b];
  const int sub = tid & 3;
  const int nch = Qp / 32;
  const half_t ninf = __builtin_bit_cast(half_t, (uint16_t)0xFC00);
  const h2 ninf2 = {ninf, ninf};
  uint16_t* ubrow = ub + l0_row(cand_off, b);
  const uint16_t* Sb = S + (int64_t)b * C * Qp;
  int32_t* my = stage + tid * 16;
  for (int64_t i = (int64_t)bxi * 256 + (tid >> 2); i < n; i += (int64_t)bxn * 256) {
    const int32_t pid = cand_pid[beg + i];
    const int64_t u0 = uoff[pid];
    const int len = (int)(uoff[pid + 1] - u0);
    const int32_t* cp = ucodes + u0;
    uint32_t sum = 0u, inf = 0u, warm = 0u;
    // (the list buffer is padded by 16 codes: a 16-byte load that starts inside the document may run past its end; those
    // positions are masked below)
    auto load_step = [&](int t, int4 (&d)[4]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        d[u] = make_int4(0, 0, 0, 0);
        if (t + 16 * u < len) __builtin_memcpy(&d[u], cp + t + 16 * u + 4 * sub, 16);
      }
    };
    for (int ch = 0; ch < nch; ++ch) {
      h2 m[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) m[j] = ninf2;
      const uint16_t* Sc = Sb + ch * 32;
      int4 cur[4], nxt[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) nxt[u] = make_int4(0, 0, 0, 0);
      load_step(0, cur);
      for (int t = 0; t < len; t += 64) {
        if (t + 64 < len) load_step(t + 64, nxt);
        uint32_t hotm = 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          *reinterpret_cast<int4*>(my + 4 * u) = cur[u];
          const int cc[4] = {cur[u].x, cur[u].y, cur[u].z, cur[u].w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bool in = t + 16 * u + 4 * sub + k < len;
            const uint32_t code = in ? (uint32_t)cc[k] : 0u;   // (past the end: whatever follows in the buffer)
            const uint32_t e4 = ((uint32_t)nib[code >> 1] >> (4u * (code & 1u))) & 15u;
            const bool hot = in && e4 >= (uint32_t)warm_t;
            hotm |= hot ? (1u << (4 * u + k)) : 0u;
            if (ch == 0) warm += (in && !hot) ? e4 : 0u;   // (a code's excess covers all the columns: counted once, not per chunk)
          }
        }
        while (hotm) {   // (per lane: ~1 of its 16 codes is hot at cfg4, so mostly one round; lanes without a hot code sit it out)
          const int k1 = __builtin_ctz(hotm);
          hotm &= hotm - 1u;
          const int k2 = hotm ? __builtin_ctz(hotm) : k1;   // (no second one: the first again -- the maximum is idempotent)
          hotm &= hotm - 1u;                                 // (0 & anything = 0)
          const int64_t c1 = my[k1], c2 = my[k2];
          uint4 r1[4], r2[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) r1[p] = *reinterpret_cast<const uint4*>(Sc + c1 * Qp + 8 * p);
#pragma unroll
          for (int p = 0; p < 4; ++p) r2[p] = *reinterpret_cast<const uint4*>(Sc + c2 * Qp + 8 * p);
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            m[4 * p + 0] = pk_max(m[4 * p + 0], pk_max(u32_as_h2(r1[p].x), u32_as_h2(r2[p].x)));
            m[4 * p + 1] = pk_max(m[4 * p + 1], pk_max(u32_as_h2(r1[p].y), u32_as_h2(r2[p].y)));
            m[4 * p + 2] = pk_max(m[4 * p + 2], pk_max(u32_as_h2(r1[p].z), u32_as_h2(r2[p].z)));
            m[4 * p + 3] = pk_max(m[4 * p + 3], pk_max(u32_as_h2(r1[p].w), u32_as_h2(r2[p].w)));
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
      }
      // the quad's maxima, then this lane's 8 columns (registers 4 sub .. 4 sub + 3)
      uint32_t mine[4];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        uint32_t x = h2_as_u32(m[j]);
        x = pk_max_raw(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
        x = pk_max_raw(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
        if ((j >> 2) == 0) mine[j & 3] = x;
        else mine[j & 3] = (sub == (j >> 2)) ? x : mine[j & 3];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int q = ch * 32 + sub * 8 + j;
        const h2 pr = u32_as_h2(mine[j >> 1]);
        float x = (float)((j & 1) ? pr.y : pr.x);
        x = x > -2.f ? x : -2.f;                       // (no hot code in the column: bin 0)
        int bin = (int)floorf(x * 128.0f) + 100;       // == S1's bins
        bin = bin < 0 ? 0 : (bin > 255 ? 255 : bin);
        if (q < Q) {
          inf |= (bin == 255) ? 1u : 0u;               // clamped: no upper bound on the score behind it
          const int f = (int)fl[q];
          sum += bin > f ? (uint32_t)(bin - f) : 0u;
        }
      }
    }
    sum = l0_red<4>(sum + warm);
    inf = l0_maxg<4>(inf);
    uint32_t v = F + sum;
    v = inf ? L0_INF : (v > 0xFFFEu ? 0xFFFEu : v);
    if (sub == 0) {
      ubrow[i] = (uint16_t)v;
      const uint32_t hb = (v >= F ? v - F : 0u) >> 2;
      atomicAdd(&hl[hb > L0_HBINS - 1 ? L0_HBINS - 1 : hb], 1u);
    }
  }
  __syncthreads();
  uint32_t* hg = hist + (int64_t)b * L0_HBINS;
  for (int i2 = tid; i2 < L0_HBINS; i2 += 1024) {
    const uint32_t v = hl[i2];
    if (v) atomicAdd(&hg[i2], v);
  }
}

// ub[b][i] = min(F + sum over the ranges of part_r[b][i], 0xFFFE), 0xFFFF if any part is infinite; + the histogram of (ub - F) >> 2
__global__ __launch_bounds__(256) void k_l0_combine(const uint16_t* __restrict__ parts, int nr, int64_t pstride, const uint32_t* __restrict__ Fsum,
                                                    const int64_t* __restrict__ cand_off, uint16_t* __restrict__ ub, uint32_t* __restrict__ hist) {
  __shared__ uint32_t hl[L0_HBINS];
  const int b = blockIdx.y;
  const int64_t n = cand_off[b + 1] - cand_off[b];
  const int64_t row = l0_row(cand_off, b);
  const uint32_t F = Fsum[b];
  for (int i = threadIdx.x; i < L0_HBINS; i += 256) hl[i] = 0u;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 2048 + threadIdx.x * 8; i < n; i += (int64_t)gridDim.x * 2048) {
    uint32_t sum[8], inf = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum[k] = F;
    for (int r = 0; r < nr; ++r) {
      const uint4 v = *reinterpret_cast<const uint4*>(parts + (int64_t)r * pstride + row + i);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t u = l0_u16(v, k);
        sum[k] += u;
        inf |= (u == L0_INF) ? (1u << k) : 0u;
      }
    }
    uint32_t o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      o[k] = (inf & (1u << k)) ? L0_INF : (sum[k] > 0xFFFEu ? 0xFFFEu : sum[k]);
      if (i + k < n) {
        const uint32_t hb = (o[k] >= F ? o[k] - F : 0u) >> 2;
        atomicAdd(&hl[hb > L0_HBINS - 1 ? L0_HBINS - 1 : hb], 1u);
      }
    }
    *reinterpret_cast<uint4*>(ub + row + i) = make_uint4(o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16));
  }
  __syncthreads();
  uint32_t* hg = hist + (int64_t)b * L0_HBINS;
  for (int i2 = threadIdx.x; i2 < L0_HBINS; i2 += 256) {
    const uint32_t v = hl[i2];
    if (v) atomicAdd(&hg[i2], v);
  }
}

// cut[b] = F + 4 * (the largest histogram bin k with #{bin >= k} >= want), want = min(mult * keep, n); 0 when nothing is pruned
__global__ __launch_bounds__(1024) void k_l0_topcut(const uint32_t* __restrict__ hist, const int64_t* __restrict__ cand_off, int64_t n_full,
                                                    int64_t R, int mult, const uint32_t* __restrict__ Fsum, int32_t* __restrict__ cut,
                                                    int32_t* __restrict__ npilot) {
  __shared__ uint32_t part[16];
  __shared__ int s_k;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t n = cand_off[b + 1] - cand_off[b];
  int64_t keep = n;
  if (n_full < keep) keep = n_full;
  if (R < keep) keep = R;
  int64_t want = keep * mult;
  if (want > L0_PILOT_MAX / 4 * 3) want = keep > L0_PILOT_MAX / 4 * 3 ? keep : L0_PILOT_MAX / 4 * 3;   // room for the ties at the cut
  if (tid == 0) npilot[b] = 0;
  if (want >= n) {
    if (tid == 0) cut[b] = 0;
    return;
  }
  if (tid == 0) s_k = 0;
  const uint32_t* hg = hist + (int64_t)b * L0_HBINS;
  constexpr int BPT = L0_HBINS / 1024;
  uint32_t own[BPT];
  uint32_t loc = 0;
#pragma unroll
  for (int k = 0; k < BPT; ++k) {
    own[k] = hg[tid * BPT + k];
    loc += own[k];
  }
  // inclusive suffix sums from the prefix sums and the total
  uint32_t tot_all = 0;
  const uint32_t pincl = fp_block_scan_incl<uint32_t>(loc, part, &tot_all);
  const uint32_t sfx_incl = tot_all - pincl + loc;
  const uint32_t above = sfx_incl - loc;
  if (above < (uint32_t)want && sfx_incl >= (uint32_t)want) {
    uint32_t acc = above;
    int kk = tid * BPT;
#pragma unroll
    for (int k = BPT - 1; k >= 0; --k) {
      acc += own[k];
      if (acc >= (uint32_t)want) { kk = tid * BPT + k; break; }
    }
    s_k = kk;
  }
  __syncthreads();
  // the last bin collects everything above it (and the infinite bounds): cutting there keeps exactly that bin
  if (tid == 0) cut[b] = s_k <= 0 ? 0 : (int32_t)(Fsum[b] + 4u * (uint32_t)s_k);
}

// pilot group = { ub >= cut[b] }, in ANY order (only the keep-th largest of their exact scores is used).  A workgroup walks
// its slice 2048 bounds at a time (16-byte loads), stages the few that pass in LDS and reserves their range with ONE global
// atomic per step (a per-wave atomic on the query's counter cost 1.1 ms at cfg2: ~3000 same-address atomics per query
// serialise at the memory side).  pilot_pid / pilot_idx [b][0..min(npilot[b], L0_PILOT_MAX)); npilot may exceed the
// capacity (then nothing is pruned).
__global__ __launch_bounds__(256) void k_l0_pilot(const uint16_t* __restrict__ ub, const int64_t* __restrict__ cand_off,
                                                  const int32_t* __restrict__ cand_pid, const int32_t* __restrict__ cut,
                                                  int32_t* __restrict__ npilot, int32_t* __restrict__ pilot_pid,
                                                  int32_t* __restrict__ pilot_idx) {
  __shared__ int s_cnt, s_base;
  __shared__ int32_t s_idx[4096];   // flushed when more than half full (a step adds at most 2048), and at the end
  const int b = blockIdx.y;
  const int64_t beg = cand_off[b];
  const int64_t n = cand_off[b + 1] - beg;
  const uint32_t ct = (uint32_t)cut[b];
  const uint16_t* row = ub + l0_row(cand_off, b);
  int64_t per = (n + gridDim.x - 1) / gridDim.x;
  per = (per + 2047) & ~(int64_t)2047;
  const int64_t lo = (int64_t)blockIdx.x * per;
  const int64_t hi = lo + per < n ? lo + per : n;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  for (int64_t i00 = lo; i00 < hi; i00 += 4 * 2048) {
  // four steps' bounds are fetched together (one load per thread and step left the kernel at 1.7 TB/s)
  uint4 vv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i = i00 + j * 2048 + threadIdx.x * 8;
    vv[j] = make_uint4(0, 0, 0, 0);
    if (i < hi) vv[j] = *reinterpret_cast<const uint4*>(row + i);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i0 = i00 + j * 2048;
    if (i0 >= hi) break;   // (uniform)
    const int64_t i = i0 + threadIdx.x * 8;
    if (i < hi) {
      const uint4 v = vv[j];
      uint32_t m = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) m |= (i + k < hi && l0_u16(v, k) >= ct) ? (1u << k) : 0u;
      if (m) {
        int pos = atomicAdd(&s_cnt, __popc(m));   // LDS atomic: the order inside the group is irrelevant
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (m & (1u << k)) s_idx[pos++] = (int32_t)(i + k);
      }
    }
    __syncthreads();
    const int total = s_cnt;
    if (total > 2048 || (total && i0 + 2048 >= hi)) {
      if (threadIdx.x == 0) s_base = atomicAdd(&npilot[b], total);
      __syncthreads();
      const int base = s_base;
      for (int j = threadIdx.x; j < total; j += 256) {
        const int pos = base + j;
        if (pos < L0_PILOT_MAX) {
          const int32_t idx = s_idx[j];
          pilot_pid[(int64_t)b * L0_PILOT_MAX + pos] = cand_pid[beg + idx];
          pilot_idx[(int64_t)b * L0_PILOT_MAX + pos] = idx;   // position in the query's candidate list: where its exact score is kept
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) s_cnt = 0;
      __syncthreads();
    }
  }
  }
}

// S1's lazy form: how far below an upper-bound score `a` a document's true approximate score can lie -- the recorded slack with a
// margin for its own fp32 rounding and for the rounding of the two fp32 column sums (<= Q additions each, usually exact)
__device__ __forceinline__ float lz_delta(float slack, float a, int Q) {
  return slack * 1.01f + (float)Q * 2.4e-7f * (__builtin_fabsf(a) + 1.f);
}
// the slack of query b: sum over its real columns of the per-column bound (k_probe_tau) -- the tight one unless a scored document
// had a negative column maximum (k_approx).  Called by every thread of a workgroup of >= 64 threads; `red` is one shared float.
__device__ __forceinline__ float lz_query_slack(const float* __restrict__ tight, const float* __restrict__ loose, const uint32_t* __restrict__ negflag,
                                                int b, int Q, int Qp, float* red) {
  if (threadIdx.x < 64) {
    const float* src = (negflag[b] ? loose : tight) + (int64_t)b * Qp;
    float v = 0.f;
    for (int q = (int)threadIdx.x; q < Q; q += 64) v += src[q];
#pragma unroll
    for (int x = 32; x > 0; x >>= 1) v += __shfl_xor(v, x, 64);
    if (threadIdx.x == 0) *red = v;
  }
  __syncthreads();
  return *red;
}
// A_T = keep-th largest exact score of the pilot group (radix select over the monotone keys, 4 x 8 bits, LDS histogram)
// -> cut[b] = floor(128 A_T + 99 Q) + 1 (0 = keep everything).  One workgroup per query.
__global__ __launch_bounds__(1024) void k_l0_thr(const float* __restrict__ pilot, const int32_t* __restrict__ npilot,
                                                 const int64_t* __restrict__ cand_off, int64_t n_full, int64_t R, int Q,
                                                 int32_t* __restrict__ cut, const float* __restrict__ lz_tight /*nullable: S1's lazy form*/,
                                                 const float* __restrict__ lz_loose, const uint32_t* __restrict__ lz_neg, int Qp) {
  __shared__ uint32_t h[256], s_wtot[4];
  __shared__ uint32_t s_prefix, s_rem;
  __shared__ float s_slack;
  const int b = blockIdx.x, tid = threadIdx.x;
  // the first four scores of every thread (a pilot group of 4 R = 4096) are fetched once, ahead of the slack's dependent loads, and
  // kept in registers for the four passes
  const float* pv = pilot + (int64_t)b * L0_PILOT_MAX;
  float pre[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) pre[k] = pv[tid + 1024 * k];   // (unconditional: rows have L0_PILOT_MAX slots; masked by np below)
  const int64_t n = cand_off[b + 1] - cand_off[b];
  const int np = npilot[b];
  float qslack = 0.f;
  if (lz_tight) qslack = lz_query_slack(lz_tight, lz_loose, lz_neg, blockIdx.x, Q, Qp, &s_slack);   // (before any thread leaves: it holds a barrier)
  int64_t keep = n;
  if (n_full < keep) keep = n_full;
  if (R < keep) keep = R;
  if (keep >= n || np < keep || np > L0_PILOT_MAX || keep < 1) {
    if (tid == 0) cut[b] = 0;
    return;
  }
  if (tid == 0) { s_prefix = 0u; s_rem = (uint32_t)keep; }
  const int lane = tid & 63, wave = tid >> 6;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) h[tid] = 0u;
    __syncthreads();
    const uint32_t prefix = s_prefix;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // (scores of one query share their leading bits: the lanes whose digit equals the first active lane's go in with ONE atomic)
      const uint32_t key = mono32(pre[k]);
      const bool on = tid + 1024 * k < np && (pass == 0 || (key >> (shift + 8)) == prefix);
      const uint32_t dg = (key >> shift) & 0xFFu;
      const unsigned long long act = __ballot(on);
      if (act) {
        const uint32_t d0 = (uint32_t)__shfl((int)dg, __builtin_ctzll(act), 64);
        const unsigned long long same = __ballot(on && dg == d0);
        if (lane == __builtin_ctzll(act)) atomicAdd(&h[d0], (uint32_t)__builtin_popcountll(same));
        if (on && dg != d0) atomicAdd(&h[dg], 1u);
      }
    }
    for (int i = tid + 4096; i < np; i += 1024) {
      const uint32_t key = mono32(pv[i]);
      if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&h[(key >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    // the largest digit d with #{digit >= d} >= rem (d = 0 if none): inclusive suffix sums of the 256 bins -- a wave scan in each
    // of the first four waves plus their totals (a Hillis-Steele scan in LDS was sixteen barriers per pass)
    const uint32_t rem = s_rem;
    const uint32_t own = tid < 256 ? h[tid] : 0u;
    uint32_t sfx = own;
    if (tid < 256) {
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = (uint32_t)__shfl_down((int)sfx, off, 64);
        if (lane + off < 64) sfx += v;
      }
      if (lane == 0) s_wtot[wave] = sfx;
    }
    __syncthreads();
    if (tid < 256) {
      for (int w = wave + 1; w < 4; ++w) sfx += s_wtot[w];
      const uint32_t above = sfx - own;
      if (above < rem && (sfx >= rem || tid == 0)) {
        s_prefix = (prefix << 8) | (uint32_t)tid;
        s_rem = rem - above;
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    float at = unmono32(s_prefix);
    // S1's lazy form: the pilot scores are upper bounds A_up; every document's true score is at least A_up - the query's slack,
    // so the keep-th largest TRUE score is at least this:
    if (lz_tight) at = at - lz_delta(qslack, at, Q);
    int32_t c = 0;
    if (at == at) {   // NaN scores: prune nothing
      const double t = floor(128.0 * (double)at + 99.0 * (double)Q) + 1.0;
      c = t <= 0.0 ? 0 : (t >= 65535.0 ? 65535 : (int32_t)t);   // 65535 keeps only the infinite bounds
    }
    cut[b] = c;
  }
}

// ordered compaction of { ub >= thr[b] } (same scheme as k_surv_count / k_surv_compact).  A workgroup takes a chunk of
// FP_L0_CHUNK = 8192 bounds: four 16-byte loads per thread, all in flight together (with one load per thread and 2048-bound
// chunks the two kernels were 10^4 workgroups of three dependent round trips each: 22 + 44 us for 42 MB).  Candidate order
// inside a chunk: (vector v, thread t, element k) -> position c0 + 2048 v + 8 t + k.
#define L0_VPT 4
static_assert(FP_L0_CHUNK == 256 * 8 * L0_VPT, "L0_VPT 16-byte loads per thread cover a chunk");
// blkcnt: survivors per chunk; blkcntx: those of them outside the pilot group ("extras": ub < cutp[b], or the group overflowed)
__global__ __launch_bounds__(256) void k_l0_count(const uint16_t* __restrict__ ub, const int64_t* __restrict__ cand_off,
                                                  const int32_t* __restrict__ thr, const int32_t* __restrict__ cutp,
                                                  const int32_t* __restrict__ npilot, int32_t* blkcnt, int32_t* blkcntx, int nblk,
                                                  uint32_t* ctr /*nullable: [B + 1] zeroed tickets -> also the scans and surv_off*/,
                                                  int32_t* nsurv, int32_t* nextra, int B, int64_t* surv_off) {
  const int b = blockIdx.y;
  const int64_t n = cand_off[b + 1] - cand_off[b];
  const int64_t c0 = (int64_t)blockIdx.x * FP_L0_CHUNK;
  int cnt = 0, cx = 0;
  if (c0 < n) {
    const uint32_t ct = (uint32_t)thr[b];
    const uint32_t cp = npilot[b] <= L0_PILOT_MAX ? (uint32_t)cutp[b] : 0x10000u;
    const uint16_t* row = ub + l0_row(cand_off, b);
    uint4 v[L0_VPT];
#pragma unroll
    for (int j = 0; j < L0_VPT; ++j) {
      const int64_t i = c0 + j * 2048 + threadIdx.x * 8;
      v[j] = make_uint4(0, 0, 0, 0);
      if (i < n) v[j] = *reinterpret_cast<const uint4*>(row + i);
    }
#pragma unroll
    for (int j = 0; j < L0_VPT; ++j) {
      const int64_t i = c0 + j * 2048 + threadIdx.x * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t u = l0_u16(v[j], k);
        const bool keep = i + k < n && u >= ct;
        cnt += keep ? 1 : 0;
        cx += (keep && u < cp) ? 1 : 0;
      }
    }
  }
  __shared__ int s[256];
  s[threadIdx.x] = cnt | (cx << 16);   // both fit 16 bits (<= 8192 per chunk)
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (!ctr) {
    if (threadIdx.x == 0) {
      blkcnt[(int64_t)b * nblk + blockIdx.x] = s[0] & 0xFFFF;
      blkcntx[(int64_t)b * nblk + blockIdx.x] = s[0] >> 16;
    }
    return;
  }
  if (threadIdx.x == 0) {
    fp_publish(blkcnt + (int64_t)b * nblk + blockIdx.x, s[0] & 0xFFFF);
    fp_publish(blkcntx + (int64_t)b * nblk + blockIdx.x, s[0] >> 16);
  }
  if (!fp_ticket_last(ctr + b, (uint32_t)nblk)) return;
  const int tot = fp_scan_counts_256(blkcnt + (int64_t)b * nblk, nblk, s);
  const int totx = fp_scan_counts_256(blkcntx + (int64_t)b * nblk, nblk, s);
  if (threadIdx.x == 0) { fp_publish(nsurv + b, tot); nextra[b] = totx; }
  if (!fp_ticket_last(ctr + B, (uint32_t)B)) return;
  fp_offsets_256(nsurv, B, surv_off, 0, nullptr, nullptr, nullptr);
}

// A survivor that was in the pilot group (ub >= cutp[b], the group held in full) already has its exact score at
// cand_approx[position]: it is copied; the others go on the query's "extra" list, in candidate order, as (document,
// destination) and are scored by one more k_approx that scatters into surv_approx.
__global__ __launch_bounds__(256) void k_l0_compact(const uint16_t* __restrict__ ub, const int64_t* __restrict__ cand_off,
                                                    const int32_t* __restrict__ cand_pid, const int32_t* __restrict__ thr,
                                                    const int32_t* __restrict__ cutp, const int32_t* __restrict__ npilot,
                                                    const float* __restrict__ cand_approx, const int32_t* __restrict__ blkoff,
                                                    const int32_t* __restrict__ blkoffx, int nblk,
                                                    const int64_t* __restrict__ surv_off, int32_t* __restrict__ surv_pid,
                                                    float* __restrict__ surv_approx, int32_t* __restrict__ xpid, int32_t* __restrict__ xdst) {
  const int b = blockIdx.y;
  const int64_t beg = cand_off[b];
  const int64_t n = cand_off[b + 1] - beg;
  const int64_t c0 = (int64_t)blockIdx.x * FP_L0_CHUNK;
  if (c0 >= n) return;
  const uint32_t ct = (uint32_t)thr[b];
  const uint32_t cp = npilot[b] <= L0_PILOT_MAX ? (uint32_t)cutp[b] : 0x10000u;
  const uint16_t* row = ub + l0_row(cand_off, b);
  uint4 v[L0_VPT];
#pragma unroll
  for (int j = 0; j < L0_VPT; ++j) {
    const int64_t i = c0 + j * 2048 + threadIdx.x * 8;
    v[j] = make_uint4(0, 0, 0, 0);
    if (i < n) v[j] = *reinterpret_cast<const uint4*>(row + i);
  }
  uint32_t keep[L0_VPT], inp[L0_VPT];
  int packed[L0_VPT];   // survivors | extras << 16 of this thread's eight bounds of vector j
#pragma unroll
  for (int j = 0; j < L0_VPT; ++j) {
    const int64_t i = c0 + j * 2048 + threadIdx.x * 8;
    keep[j] = 0;
    inp[j] = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t u = l0_u16(v[j], k);
      if (i + k < n && u >= ct) {
        keep[j] |= 1u << k;
        if (u >= cp) inp[j] |= 1u << k;
      }
    }
    packed[j] = __popc(keep[j]) | (__popc(keep[j] & ~inp[j]) << 16);
  }
  // The gathers of everything this thread keeps, issued before the prefix is even known (they do not depend on it): by the time
  // the scans are through, the values are there.  (Interleaved with the stores -- a load, its store, the next load -- every
  // survivor of a wave cost a round trip of its own, up to 32 in a row per workgroup.)
  int32_t pid[L0_VPT][8];
  float ap[L0_VPT][8];
#pragma unroll
  for (int j = 0; j < L0_VPT; ++j) {
    const int64_t i = c0 + j * 2048 + threadIdx.x * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      pid[j][k] = (keep[j] & (1u << k)) ? cand_pid[beg + i + k] : 0;
      ap[j][k] = (inp[j] & (1u << k)) ? cand_approx[beg + i + k] : 0.f;
    }
  }
  // inclusive scans of the four vectors: inside a wave by shuffles, across the four waves through 16 words of LDS (one barrier
  // instead of the sixteen of a shared-memory scan)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl[L0_VPT];
#pragma unroll
  for (int j = 0; j < L0_VPT; ++j) {
    int x = packed[j];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    incl[j] = x;
  }
  __shared__ int wtot[L0_VPT][4];
  if (lane == 63) {
#pragma unroll
    for (int j = 0; j < L0_VPT; ++j) wtot[j][wave] = incl[j];
  }
  __syncthreads();
  const int64_t sb = surv_off[b];
  int64_t pos0 = sb + blkoff[(int64_t)b * nblk + blockIdx.x];
  int64_t xp0 = sb + blkoffx[(int64_t)b * nblk + blockIdx.x];
#pragma unroll
  for (int j = 0; j < L0_VPT; ++j) {
    int before = 0, tot = 0;   // survivors | extras of the earlier waves / of the whole vector
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int t = wtot[j][w];
      before += w < wave ? t : 0;
      tot += t;
    }
    if (keep[j]) {
      const int excl = before + incl[j] - packed[j];
      int64_t pos = pos0 + (excl & 0xFFFF);
      int64_t xp = xp0 + (excl >> 16);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (keep[j] & (1u << k)) {
          surv_pid[pos] = pid[j][k];
          if (inp[j] & (1u << k)) {
            surv_approx[pos] = ap[j][k];
          } else {
            xpid[xp] = pid[j][k];
            xdst[xp] = (int32_t)(pos - sb);
            ++xp;
          }
          ++pos;
        }
      }
    }
    pos0 += tot & 0xFFFF;
    xp0 += tot >> 16;
  }
}

#define FP_L0_RANGE (1ll << 17)
size_t fpk_l0_lds_bytes(const FpIndexDev& ix) {
  const int64_t Cpad = (ix.C + 15) & ~(int64_t)15;
  return (size_t)std::min<int64_t>(Cpad, FP_L0_RANGE) + 16 + 256 + L0_HBINS * 4;
}
bool fpk_l0_fits(const FpIndexDev& ix) { return ix.n_ranges >= 1 && ix.pcodes != nullptr; }
// the "hot" form of level 0 (k_l0h_scan): any index whose byte table is one LDS slice
bool fpk_l0h_fits(const FpIndexDev& ix) { return ix.C <= FP_L0_RANGE && ix.ucodes != nullptr; }
void fpk_l0h_scan(const FpIndexDev& ix, const uint16_t* S, const FpSearchShape& sh, const int64_t* cand_off, const int32_t* cand_pid, int64_t M,
                  FpL0Scratch& w, hipStream_t st) {
  const int B = sh.B;
  const int64_t Cpad = (ix.C + 15) & ~(int64_t)15;
  static std::atomic<uint64_t> okh{0};
  fp_allow_big_lds((const void*)k_l0h_scan, okh, 160 * 1024);
  const int64_t per_q = (M + B - 1) / B;
  int64_t bx = (per_q + 4095) / 4096;
  if (bx > 8192) bx = 8192;
  if (bx < 1) bx = 1;
  const size_t lds = (size_t)((Cpad / 2 + 15) & ~(int64_t)15) + L0_HBINS * 4 + 256 + 1024 * 64;   // excess nibbles | histogram | floors | the lanes' staged codes
  // codes whose excess is below this many bins are not gathered, their excess is added to the bound (1: gather every hot code)
  static const int warm_t = [] { const int v = (int)fp_test_opt("l0h_warm", 8); return v < 1 ? 1 : (v > 15 ? 15 : v); }();
  hipLaunchKernelGGL(k_l0h_scan, dim3((unsigned)bx, (unsigned)B), dim3(1024), lds, st, w.e8, Cpad, (int)Cpad, w.floors, w.Fsum, cand_off, cand_pid,
                     ix.uoff, ix.ucodes, S, ix.C, sh.Q, sh.Qp, w.ub, w.hist, (int)bx, warm_t);
}

// level 0 in launch groups so that the scan kernel can be timed alone:
//   prepare: floors + excess table;  scan: UB0 of every candidate (+ its histogram);  pilot: the top mult*keep documents by UB0
// floors (+ their sum) from the bins of a centroid sample: S8s is [B][nch][ns][32], the bins of the centroids 0, stride, 2 stride, ...
// (fpk_l0_sample_plan), exactly the sample k_l0_floor takes from a full table
void fpk_l0_sample_plan(const FpIndexDev& ix, int64_t* n_rows, int64_t* stride) {
  const int64_t want = L0_SAMPLE_PRE;
  *stride = ix.C > want ? ix.C / want : 1;
  *n_rows = (ix.C + *stride - 1) / *stride;
}
void fpk_l0_floors(const uint8_t* S8s, int64_t n_rows, const FpSearchShape& sh, uint8_t* floors, uint32_t* Fsum, uint32_t* esc, uint16_t* gfl,
                   hipStream_t st, float hot_tail, int gfl_rd) {
  const int nch = sh.Qp / 32;
  static const float tail0 = [] { const float v = (float)fp_test_opt("l0_tail", 0.025); return (v > 0.f && v < 0.5f) ? v : 0.025f; }();
  static const float tailh = [] { const float v = (float)fp_test_opt("l0h_tail", 0.0); return (v > 0.f && v < 0.5f) ? v : 0.f; }();   // (0: the caller's)
  const float tail = hot_tail > 0.f ? (tailh > 0.f ? tailh : hot_tail) : tail0;
  static std::atomic<uint64_t> ok1{0};
  fp_allow_big_lds((const void*)k_l0_floor, ok1, 136 * 1024);   // (four column chunks: 128 x 257 words)
  // n_rows <= L0_SAMPLE * 2: the kernel's own sampling stride over this table is 1 or (for 8192 < n_rows) still covers it
  hipLaunchKernelGGL(k_l0_floor, dim3((unsigned)sh.B), dim3(1024), (size_t)nch * 32 * 257 * 4, st, S8s, n_rows, sh.Q, nch, tail, floors, Fsum, esc,
                     reinterpret_cast<half_t*>(gfl), gfl_rd);
}

// S8 != nullptr: floors and the table from the full 8-bit table (two passes over it);  S8 == nullptr: both were produced
// around S1 (fpk_l0_floors on a sampled pre-pass, the table in S1's epilogue) and only the histogram is cleared here
size_t fpk_l0_hist_bytes(int B) { return (size_t)B * L0_HBINS * 4; }
void fpk_l0_prepare(const FpIndexDev& ix, const uint8_t* S8, const FpSearchShape& sh, FpL0Scratch& w, hipStream_t st, bool hist_prezeroed) {
  const int B = sh.B;
  const int nch = sh.Qp / 32;
  const int64_t Cpad = (ix.C + 15) & ~(int64_t)15;
  if (!hist_prezeroed) (void)hipMemsetAsync(w.hist, 0, fpk_l0_hist_bytes(B), st);
  if (!S8) return;
  static const float tail = [] { const float v = (float)fp_test_opt("l0_tail", 0.025); return (v > 0.f && v < 0.5f) ? v : 0.025f; }();
  static std::atomic<uint64_t> ok1{0};
  fp_allow_big_lds((const void*)k_l0_floor, ok1, 136 * 1024);   // (four column chunks: 128 x 257 words)
  hipLaunchKernelGGL(k_l0_floor, dim3((unsigned)B), dim3(1024), (size_t)nch * 32 * 257 * 4, st, S8, ix.C, sh.Q, nch, tail, w.floors, w.Fsum, w.esc,
                     (half_t*)nullptr, 0);
  hipLaunchKernelGGL(k_l0_table, dim3((unsigned)((Cpad + 255) / 256), (unsigned)B), dim3(256), 0, st, S8, ix.C, Cpad, sh.Q, nch, w.floors, w.e8,
                     w.esc);
}

void fpk_l0_scan(const FpIndexDev& ix, const FpSearchShape& sh, const int64_t* cand_off, const int32_t* cand_pid, int64_t M, FpL0Scratch& w,
                 hipStream_t st) {
  const int B = sh.B;
  const int64_t Cpad = (ix.C + 15) & ~(int64_t)15;
  static std::atomic<uint64_t> ok8{0}, ok4{0};
  fp_allow_big_lds((const void*)k_l0_scan<8>, ok8, 160 * 1024);
  fp_allow_big_lds((const void*)k_l0_scan<4>, ok4, 160 * 1024);
  const int64_t per_q = (M + B - 1) / B;
  // each workgroup copies the table (up to 128 KiB) into LDS first: candidates per workgroup (measured at cfg2: 256 / 512 / 1024 /
  // 2048 / 4096 -> 1.99 / 1.76 / 1.60 / 1.51 / 1.51 ms with the first version of the kernel)
  const int cpw = 4096;
  int64_t bx = (per_q + cpw - 1) / cpw;
  if (bx > 8192) bx = 8192;
  if (bx < 1) bx = 1;
  // measured at cfg2: XCD-affine 0.674 ms, plain (chunk, query) grid 0.565 ms -- the table copies are the smaller effect; with
  // the plain grid the workgroups of ~4 queries sweep the corpus together on all XCDs and share the code lines in the
  // memory-side cache.  Experiment only: FP_L0_XCD=1.
  const int affine_env = 0;
  // (also tried, round 3: query-fastest dispatch order, so that the 256 resident workgroups belong to all 64 queries and walk the
  // same stretch of the corpus together -- lines shared in L2 instead of read once per query: 0.98 ms against 0.56 ms, and
  // fewer, longer workgroups made it worse still; profiles/r03_l0_order_lab.txt)
  const int affine = (affine_env && B % 8 == 0 && bx * (int64_t)B < (1ll << 31)) ? 1 : 0;
  const dim3 grid = affine ? dim3((unsigned)(bx * B)) : dim3((unsigned)bx, (unsigned)B);
  FpL0Multi mr{};
  mr.nr = ix.n_ranges;
  mr.ub_stride = w.ub_stride;
  mr.seq_r = 0;
  for (int r = 0; r < ix.n_ranges && r < 8; ++r) { mr.x[r] = ix.pcodes_r[r]; mr.po[r] = ix.poff_r[r]; }
  // Tables of several ranges: ALL ranges in one launch.  A document's first lines of the nr ranges lie side by side (64-byte
  // lines: two ranges per 128-byte fabric request -- the fabric moves ~47 G requests/s whatever their size, and one launch per
  // range over the range's own lines made every 64-byte line a request of its own: 2.6 G requests = the 52 ms of cfg3's
  // scan).  The nr workgroups of a chunk are dealt to the SAME XCD (ids 8 apart) back to back, so they run together and the
  // second range's line is an L2 hit.  Each writes its partial sums to its own array; k_l0_combine adds F and the parts and
  // builds the histogram.  FP_L0_MULTI=0: one launch per range, accumulating in place (the form before round 3's end).
  const int multi_env = 1;
  if (ix.n_ranges > 1 && multi_env && w.ub_parts && ((bx + 7) & ~7ll) * ix.n_ranges * (int64_t)B < 0x7FFFFFFFll) {
    const dim3 g1((unsigned)(((bx + 7) & ~7ll) * ix.n_ranges * B));
    const int pair_env = 0;   // (lines read as 128-byte pairs of ranges: measured slower at cfg3, 61.0 vs 46.9 ms)
    mr.pair = (pair_env && ix.l0_ppl == 4 && ix.n_ranges % 2 == 0) ? 1 : 0;
    if (ix.l0_ppl == 4 && !mr.pair)
      hipLaunchKernelGGL((k_l0_scan<4>), g1, dim3(1024), fpk_l0_lds_bytes(ix), st, w.e8, Cpad, (int64_t)0, 0, w.esc, w.Fsum, cand_off, cand_pid,
                         (const int32_t*)nullptr, ix.pcodes, w.ub_parts, w.hist, 1, 0, (int)bx, 2, mr);
    else
      hipLaunchKernelGGL((k_l0_scan<8>), g1, dim3(1024), fpk_l0_lds_bytes(ix), st, w.e8, Cpad, (int64_t)0, 0, w.esc, w.Fsum, cand_off, cand_pid,
                         (const int32_t*)nullptr, ix.pcodes, w.ub_parts, w.hist, 1, 0, (int)bx, 2, mr);
    int64_t cb = (per_q + 16383) / 16384;
    if (cb > 256) cb = 256;
    if (cb < 1) cb = 1;
    hipLaunchKernelGGL(k_l0_combine, dim3((unsigned)cb, (unsigned)B), dim3(256), 0, st, w.ub_parts, ix.n_ranges, w.ub_stride, w.Fsum, cand_off, w.ub,
                       w.hist);
    return;
  }
  for (int r = 0; r < ix.n_ranges; ++r) {
    const int64_t off = (int64_t)r * FP_L0_RANGE;
    const int tb = (int)std::min<int64_t>(FP_L0_RANGE, Cpad - off);
    mr.seq_r = r;
#define L0_LAUNCH(L_) \
    hipLaunchKernelGGL((k_l0_scan<L_>), grid, dim3(1024), fpk_l0_lds_bytes(ix), st, w.e8, Cpad, off, tb, w.esc, w.Fsum, cand_off, \
                       cand_pid, ix.poff_r[r], ix.pcodes, w.ub, w.hist, r == 0 ? 1 : 0, r == ix.n_ranges - 1 ? 1 : 0, (int)bx, affine, mr)
    if (ix.l0_ppl == 4) L0_LAUNCH(4);
    else L0_LAUNCH(8);
#undef L0_LAUNCH
  }
}

void fpk_l0_pilot(const FpSearchShape& sh, const int64_t* cand_off, const int32_t* cand_pid, int64_t M, FpL0Scratch& w, hipStream_t st) {
  const int B = sh.B;
  static const int mult = [] { const int v = (int)fp_test_opt("l0_pilot", 4); return v >= 1 && v <= 16 ? v : 4; }();
  hipLaunchKernelGGL(k_l0_topcut, dim3((unsigned)B), dim3(1024), 0, st, w.hist, cand_off, sh.n_full, sh.R, mult, w.Fsum, w.cut, w.npilot);
  const int64_t per_q = (M + B - 1) / B;
  // candidates per workgroup: every workgroup ends with one atomic on its query's counter, and the B counters share a cache
  // line -- device-scope atomics on one line retire at ~6 ns each, whatever the address (8192 / 16384 / 32768 / 65536 per
  // workgroup at cfg2: 28.3 / 19.8 / 25.7 / 42.3 us; k_l0_count reads the same 42 MB without the atomics in 12.7 us)
  const int cpb = 16384;
  int64_t bx = (per_q + cpb - 1) / cpb;
  if (bx > 64) bx = 64;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(k_l0_pilot, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, st, w.ub, cand_off, cand_pid, w.cut, w.npilot, w.pilot_pid,
                     w.pilot_idx);
}

int64_t fpk_l0_pilot_cap() { return L0_PILOT_MAX; }

// level 0, second half (the pilot group's exact scores are in w.pilot_approx and, by candidate position, in w.cand_approx):
// threshold, ordered survivors with the pilot members' scores copied, the others on the extra list (w.xpid / w.xdst / w.nextra)
void fpk_l0_survivors(const FpSearchShape& sh, const int64_t* cand_off, const int32_t* cand_pid, FpL0Scratch& w, int32_t* nsurv,
                      int64_t* surv_off, int32_t* surv_pid, float* surv_approx, hipStream_t st, const FpLazyS1* lz) {
  const int B = sh.B;
  hipLaunchKernelGGL(k_l0_thr, dim3((unsigned)B), dim3(1024), 0, st, w.pilot_approx, w.npilot, cand_off, sh.n_full, sh.R, sh.Q, w.thr,
                     lz ? lz->tight : (const float*)nullptr, lz ? lz->loose : (const float*)nullptr, lz ? lz->negflag : (const uint32_t*)nullptr, sh.Qp);
  const bool one_launch = w.tickets && (int64_t)w.nblk * B <= FP_TICKET_MAX_WGS;
  hipLaunchKernelGGL(k_l0_count, dim3((unsigned)w.nblk, (unsigned)B), dim3(256), 0, st, w.ub, cand_off, w.thr, w.cut, w.npilot, w.blkcnt,
                     w.blkcntx, w.nblk, one_launch ? w.tickets : (uint32_t*)nullptr, nsurv, w.nextra, B, surv_off);
  if (!one_launch && w.tickets) {   // both scans and the offsets in one more launch
    hipLaunchKernelGGL(k_cand_scan, dim3((unsigned)B), dim3(256), 0, st, w.blkcnt, w.nblk, nsurv, w.blkcntx, w.nextra, w.tickets + B, B, surv_off,
                       (int64_t)0, (int32_t*)nullptr, (int64_t*)nullptr, (const int32_t*)nullptr);
  } else if (!one_launch) {
    hipLaunchKernelGGL(k_cand_scan, dim3((unsigned)B), dim3(256), 0, st, w.blkcnt, w.nblk, nsurv, w.blkcntx, w.nextra, (uint32_t*)nullptr, 0,
                       (int64_t*)nullptr, (int64_t)0, (int32_t*)nullptr, (int64_t*)nullptr, (const int32_t*)nullptr);
    hipLaunchKernelGGL(k_cand_offsets, dim3(1), dim3(256), 0, st, nsurv, B, surv_off, (int64_t)0, (int32_t*)nullptr, (int64_t*)nullptr,
                       (const int32_t*)nullptr);
  }
  hipLaunchKernelGGL(k_l0_compact, dim3((unsigned)w.nblk, (unsigned)B), dim3(256), 0, st, w.ub, cand_off, cand_pid, w.thr, w.cut, w.npilot,
                     w.cand_approx, w.blkcnt, w.blkcntx, w.nblk, surv_off, surv_pid, surv_approx, w.xpid, w.xdst);
}

// ============================================================================================
// S5  top-R selection by (approx desc, doc id asc): 3-pass radix select (11+11+10 bits) on
// the monotone key, then one ordered collect pass per query.
// selstate[b] = {need_select, keep, prefix, k_rem, n, -, -, -}
// ============================================================================================
// The radix select used to be init + 3 x (histogram, scan) + gather: eleven launches with the fill.  The one-workgroup-per-query
// steps now ride in the kernel that needs their result: every workgroup of histogram pass p+1 (and of the gather) repeats the
// scan of pass p's 2048 bins for itself (2 us of a 256-thread workgroup) and workgroup 0 records it for the launches after.
// State between launches: st[b] = {need_select, keep, prefix0 = 0, k_rem0 = keep, n, gt, eq, overflow}, ext[b] = {prefix, k_rem}
// after pass 0 and after pass 1 (two pairs: a launch never reads a word that one of its own workgroups writes).
__device__ __forceinline__ void sel_init_values(int64_t n, int64_t n_full, int64_t R, int64_t& keep) {
  keep = n;
  if (n_full < keep) keep = n_full;   // search.rs:605-611
  if (R < keep) keep = R;             // :614-619 (R = max(n_full/4, 1))
  if (keep < 0) keep = 0;
}
// scan of one pass's histogram `g` from the top by a 256-thread workgroup: (prefix, k_rem) -> those after the pass, in all threads
template <int PASS>
__device__ __forceinline__ void sel_scan_local(const uint32_t* __restrict__ g, uint32_t& prefix, uint32_t& k_rem) {
  constexpr int WIDTH = (PASS == 2) ? 10 : 11;
  constexpr int NB = 1 << WIDTH;
  constexpr int PER = NB / 256;  // bins per thread (8 or 4)
  // thread t owns bins [NB-1 - t*PER - (PER-1), NB-1 - t*PER] walking from the top
  uint32_t loc[PER];
  uint32_t sum = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) { loc[j] = g[NB - 1 - (threadIdx.x * PER + j)]; sum += loc[j]; }
  __shared__ uint32_t sc[16];
  __shared__ uint32_t s_out[2];
  if (threadIdx.x == 0) { s_out[0] = prefix << WIDTH; s_out[1] = k_rem; }
  const uint32_t incl = fp_block_scan_incl<uint32_t>(sum, sc), excl = incl - sum;   // (its barriers also order s_out)
  if (excl < k_rem && incl >= k_rem) {  // exactly one thread
    uint32_t cum = excl;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      if (cum + loc[j] >= k_rem) {
        const uint32_t bin = NB - 1 - (threadIdx.x * PER + j);
        s_out[0] = (prefix << WIDTH) | bin;
        s_out[1] = k_rem - cum;  // still needed from inside this bin
        break;
      }
      cum += loc[j];
    }
  }
  __syncthreads();
  prefix = s_out[0];
  k_rem = s_out[1];
}

template <int PASS>
__global__ __launch_bounds__(256) void k_sel_hist(const float* __restrict__ approx, const int64_t* __restrict__ cand_off,
                                                  uint32_t* __restrict__ st, uint32_t* __restrict__ hist, uint32_t* __restrict__ ext,
                                                  int64_t n_full, int64_t R, int32_t* __restrict__ sel_cnt) {
  constexpr int SHIFT = (PASS == 0) ? 21 : (PASS == 1) ? 10 : 0;
  constexpr int WIDTH = (PASS == 2) ? 10 : 11;
  const int b = blockIdx.y, B = gridDim.y;
  uint32_t* s = st + (int64_t)b * 8;
  uint32_t* e = ext + (int64_t)b * 4;
  const int64_t beg = cand_off[b], end = cand_off[b + 1];
  const int64_t i_pre = beg + (int64_t)blockIdx.x * 256 + threadIdx.x;   // (fetched ahead of the previous pass's scan: see k_sel_gather_lz)
  const float a_pre = i_pre < end ? approx[i_pre] : 0.f;
  uint32_t prefix = 0;
  if (PASS == 0) {
    const int64_t n = end - beg;
    int64_t keep;
    sel_init_values(n, n_full, R, keep);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      s[0] = (keep < n) ? 1u : 0u;
      s[1] = (uint32_t)keep;
      s[2] = 0u;
      s[3] = (uint32_t)keep;
      s[4] = (uint32_t)n;
      s[5] = 0u;  // elements strictly above the threshold gathered so far
      s[6] = 0u;  // elements equal to the threshold seen
      s[7] = 0u;  // tie buffer overflow -> ordered fallback
      sel_cnt[b] = (int32_t)keep;
    }
    if (!(keep < n)) return;
  } else {
    if (!s[0]) return;
    uint32_t k_rem;
    if (PASS == 1) { prefix = s[2]; k_rem = s[3]; } else { prefix = e[0]; k_rem = e[1]; }
    sel_scan_local<PASS - 1 < 0 ? 0 : PASS - 1>(hist + ((int64_t)(PASS - 1) * B + b) * FP_SEL_BINS, prefix, k_rem);
    if (blockIdx.x == 0 && threadIdx.x == 0) { e[2 * (PASS - 1)] = prefix; e[2 * (PASS - 1) + 1] = k_rem; }
  }
  __shared__ uint32_t h[FP_SEL_BINS];
  for (int i = threadIdx.x; i < FP_SEL_BINS; i += 256) h[i] = 0;
  __syncthreads();
  for (int64_t i = i_pre; i < end; i += (int64_t)gridDim.x * 256) {
    uint32_t k = mono32(i == i_pre ? a_pre : approx[i]);
    bool match = (PASS == 0) ? true : ((k >> (SHIFT + WIDTH)) == prefix);
    if (match) atomicAdd(&h[(k >> SHIFT) & ((1u << WIDTH) - 1)], 1u);
  }
  __syncthreads();
  uint32_t* g = hist + ((int64_t)PASS * B + b) * FP_SEL_BINS;
  for (int i = threadIdx.x; i < FP_SEL_BINS; i += 256)
    if (h[i]) atomicAdd(&g[i], h[i]);
}

// the last scan on its own (rerank lists beyond the LDS sort: no gather kernel follows the third histogram)
__global__ __launch_bounds__(256) void k_sel_scan_final(const uint32_t* __restrict__ hist, uint32_t* __restrict__ st,
                                                        const uint32_t* __restrict__ ext, int B) {
  const int b = blockIdx.x;
  uint32_t* s = st + (int64_t)b * 8;
  if (!s[0]) return;
  uint32_t prefix = ext[(int64_t)b * 4 + 2], k_rem = ext[(int64_t)b * 4 + 3];
  sel_scan_local<2>(hist + ((int64_t)2 * B + b) * FP_SEL_BINS, prefix, k_rem);
  if (threadIdx.x == 0) { s[2] = prefix; s[3] = k_rem; }
}

// ordered collect: one block per query
__device__ __forceinline__ void sel_collect_body(const float* __restrict__ approx, const int32_t* __restrict__ cand_pid,
                                                 const int64_t* __restrict__ cand_off, const uint32_t* __restrict__ st,
                                                 int64_t R, int32_t* __restrict__ sel_pid, float* __restrict__ sel_approx) {
  const int b = blockIdx.x;
  const uint32_t* s = st + (int64_t)b * 8;
  if (!s[7]) return;  // fallback only: the tie buffer of k_sel_gather overflowed for this query
  const bool need = s[0] != 0;
  const uint32_t kstar = s[2];
  const uint32_t need_eq = s[3];
  const uint32_t keep = s[1];
  const int64_t beg = cand_off[b], n = cand_off[b + 1] - beg;
  int32_t* op = sel_pid + (int64_t)b * R;
  float* oa = sel_approx + (int64_t)b * R;
  if (!need) {
    for (int64_t i = threadIdx.x; i < n && i < (int64_t)keep; i += 1024) { op[i] = cand_pid[beg + i]; oa[i] = approx[beg + i]; }
    return;
  }
  __shared__ uint32_t wg[16], we[16];
  __shared__ uint32_t base_g, base_e;
  if (threadIdx.x == 0) { base_g = 0; base_e = 0; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t start = 0; start < n; start += 1024) {
    const int64_t i = start + threadIdx.x;
    float a = 0.f;
    bool gt = false, eq = false;
    if (i < n) {
      a = approx[beg + i];
      uint32_t k = mono32(a);
      gt = k > kstar;
      eq = k == kstar;
    }
    const unsigned long long mg = __ballot(gt), me = __ballot(eq);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t rg = __popcll(mg & below), re = __popcll(me & below);
    if (lane == 0) { wg[wave] = __popcll(mg); we[wave] = __popcll(me); }
    __syncthreads();
    uint32_t pg = base_g, pe = base_e;
    for (int w = 0; w < wave; ++w) { pg += wg[w]; pe += we[w]; }
    rg += pg;
    re += pe;
    if (gt || (eq && re < need_eq)) {
      uint32_t pos = rg + (re < need_eq ? re : need_eq);
      if (pos < keep) { op[pos] = cand_pid[beg + i]; oa[pos] = a; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t tg = 0, te = 0;
      for (int w = 0; w < 16; ++w) { tg += wg[w]; te += we[w]; }
      base_g += tg;
      base_e += te;
    }
    __syncthreads();
  }
}

// The selection's last kernel also leaves the prefix of the per-query rerank counts that S6 maps its work items with (workgroup
// 0 writes pref[0..B]) -- one launch less in front of MaxSim.
// exclusive prefix of the queries' selection counts (what S6 maps its work with), by one workgroup of 1024 threads
__device__ __forceinline__ void sel_count_prefix(const int32_t* __restrict__ sel_cnt, int64_t* __restrict__ pref, int B) {
  __shared__ long long sp[16];
  long long base = 0;
  if (threadIdx.x == 0) pref[0] = 0;
  for (int start = 0; start < B; start += 1024) {
    const int i = start + (int)threadIdx.x;
    const long long x = (i < B) ? (long long)sel_cnt[i] : 0ll;
    long long tot = 0;
    const long long incl = fp_block_scan_incl<long long>(x, sp, &tot);
    if (i < B) pref[i + 1] = base + incl;
    base += tot;
  }
}
__global__ __launch_bounds__(1024) void k_sel_collect(const float* __restrict__ approx, const int32_t* __restrict__ cand_pid,
                                                      const int64_t* __restrict__ cand_off, const uint32_t* __restrict__ st,
                                                      int64_t R, int32_t* __restrict__ sel_pid, float* __restrict__ sel_approx,
                                                      const int32_t* __restrict__ sel_cnt, int64_t* __restrict__ pref /*nullable*/, int B) {
  sel_collect_body(approx, cand_pid, cand_off, st, R, sel_pid, sel_approx);
  if (!pref || blockIdx.x != 0) return;   // (sel_cnt is final before this kernel starts: k_sel_hist<0> / k_sel_front wrote it)
  __syncthreads();
  sel_count_prefix(sel_cnt, pref, B);
}

// parallel gather: elements above the threshold go straight to the output (any order), elements
// equal to it into a tie buffer; k_sel_finish then picks the need_eq smallest doc ids among the
// ties and sorts the selection by doc id (the order every later stage expects).
__global__ __launch_bounds__(256) void k_sel_gather(const float* __restrict__ approx, const int32_t* __restrict__ cand_pid,
                                                    const int64_t* __restrict__ cand_off, uint32_t* __restrict__ st, int64_t R,
                                                    int32_t* __restrict__ sel_pid, float* __restrict__ sel_approx,
                                                    int32_t* __restrict__ tie_pid, const uint32_t* __restrict__ hist,
                                                    const uint32_t* __restrict__ ext) {
  const int b = blockIdx.y;
  uint32_t* s = st + (int64_t)b * 8;
  const bool need = s[0] != 0;
  uint32_t kstar = 0;
  if (need) {   // the third pass's scan, by every workgroup for itself; workgroup 0 records {threshold key, ties still needed}
    uint32_t k_rem = ext[(int64_t)b * 4 + 3];
    kstar = ext[(int64_t)b * 4 + 2];
    sel_scan_local<2>(hist + ((int64_t)2 * gridDim.y + b) * FP_SEL_BINS, kstar, k_rem);
    if (blockIdx.x == 0 && threadIdx.x == 0) { s[2] = kstar; s[3] = k_rem; }
  }
  const uint32_t keep = s[1];
  const int64_t beg = cand_off[b], n = cand_off[b + 1] - beg;
  int32_t* op = sel_pid + (int64_t)b * R;
  float* oa = sel_approx + (int64_t)b * R;
  int32_t* tp = tie_pid + (int64_t)b * R;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float a = approx[beg + i];
    if (!need) {  // everything is kept, already in ascending doc-id order
      if (i < (int64_t)keep) { op[i] = cand_pid[beg + i]; oa[i] = a; }
      continue;
    }
    const uint32_t key = mono32(a);
    if (key > kstar) {
      const uint32_t pos = atomicAdd(&s[5], 1u);
      if (pos < keep) { op[pos] = cand_pid[beg + i]; oa[pos] = a; }
    } else if (key == kstar) {
      const uint32_t pos = atomicAdd(&s[6], 1u);
      if (pos < (uint32_t)R) tp[pos] = cand_pid[beg + i]; else s[7] = 1u;
    }
  }
}

__global__ __launch_bounds__(1024) void k_sel_finish(uint32_t* __restrict__ st, int64_t R, int32_t* __restrict__ sel_pid,
                                                     float* __restrict__ sel_approx, const int32_t* __restrict__ tie_pid) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x;
  uint32_t* s = st + (int64_t)b * 8;
  if (!s[0] || s[7]) return;  // nothing selected away, or fallback
  const uint32_t keep = s[1], need_eq = s[3], n_gt = s[5];
  const uint32_t n_eq = s[6] < (uint32_t)R ? s[6] : (uint32_t)R;
  int32_t* op = sel_pid + (int64_t)b * R;
  float* oa = sel_approx + (int64_t)b * R;
  const int32_t* tp = tie_pid + (int64_t)b * R;
  // 1) ties: ascending doc id, first need_eq of them complete the selection
  int tp2 = 2;
  while (tp2 < (int)n_eq) tp2 <<= 1;
  unsigned int* tv = reinterpret_cast<unsigned int*>(smem);
  for (int i = threadIdx.x; i < tp2; i += 1024) tv[i] = (i < (int)n_eq) ? (unsigned int)tp[i] : 0xFFFFFFFFu;
  __syncthreads();
  for (int k = 2; k <= tp2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < tp2; i += 1024) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned int a = tv[i], c = tv[ixj];
          const bool up = ((i & k) == 0);
          if ((a > c) == up) { tv[i] = c; tv[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
  const float kval = unmono32(s[2]);
  for (int i = threadIdx.x; i < (int)need_eq && n_gt + i < keep; i += 1024) { op[n_gt + i] = (int32_t)tv[i]; oa[n_gt + i] = kval; }
  __syncthreads();
  // 2) whole selection ascending by doc id (64-bit keys: id << 32 | approx bits)
  int kp2 = 2;
  while (kp2 < (int)keep) kp2 <<= 1;
  unsigned long long* kv = reinterpret_cast<unsigned long long*>(smem);
  __syncthreads();
  for (int i = threadIdx.x; i < kp2; i += 1024)
    kv[i] = (i < (int)keep) ? (((unsigned long long)(uint32_t)op[i] << 32) | (unsigned long long)__float_as_uint(oa[i])) : ~0ull;
  __syncthreads();
  if (kp2 == 1024) {   // one key per thread: ascending == descending on the complemented keys
    const unsigned long long sorted = ~fp_sort1024_desc(~kv[threadIdx.x], kv);
    kv[threadIdx.x] = sorted;
    __syncthreads();
  } else {
    for (int k = 2; k <= kp2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = threadIdx.x; i < kp2; i += 1024) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const unsigned long long a = kv[i], c = kv[ixj];
            const bool up = ((i & k) == 0);
            if ((a > c) == up) { kv[i] = c; kv[ixj] = a; }
          }
        }
        __syncthreads();
      }
    }
  }
  for (int i = threadIdx.x; i < (int)keep; i += 1024) {
    op[i] = (int32_t)(kv[i] >> 32);
    oa[i] = __uint_as_float((uint32_t)kv[i]);
  }
}

// ---- selection over S1's lazy form (FpLazyS1): approx[] holds UPPER bounds A_up ------------------------------------------------
// The radix select above ran on A_up: kstar = key of U, the keep-th largest A_up.  With D = the largest slack of any scored
// document (lz_delta), the keep-th largest TRUE score T lies in [U - D, U], so
//   * A_up <  U - D : certainly not selected (true score below T)            -> not even gathered;
//   * A_up >  U + D : certainly selected (true score >= A_up - D > U >= T)   -> straight to the output;
//   * the rest ("maybes", a few dozen per query): their TRUE scores are recomputed from scratch -- every code of the document
//     against every query column with the reference's ascending fp32 chain, fp16 rounding, fp16 max, ascending fp32 sum, i.e.
//     exactly the oracle's approximate score (k_lz_exact) -- and the best keep - #certain of them by (score desc, id asc)
//     complete the set (k_sel_finish_lz).
// The selected SET is then the reference's (its exact ties at the cut included); sel_approx holds A_up for the certain ones
// (nobody reads it on this path).  More maybes than the list holds (gcap = max(1024, R)): *flag is raised and the caller runs the batch again
// with the eager S1.
#define LZ_EX_BLOCKS 32   // k_lz_exact: workgroups (of four waves) per query: a wave per maybe up to 128 of them (FP_TEST lz_exb)
// gather + classification: the certain ones go straight to the head of the selection (any order), the maybes' ids to gpid
__global__ __launch_bounds__(256) void k_sel_gather_lz(const float* __restrict__ approx, const int32_t* __restrict__ cand_pid,
                                                       const int64_t* __restrict__ cand_off, uint32_t* __restrict__ st, int64_t R,
                                                       int32_t* __restrict__ sel_pid, float* __restrict__ sel_approx,
                                                       const uint32_t* __restrict__ hist, const uint32_t* __restrict__ ext,
                                                       const float* __restrict__ lz_tight, const float* __restrict__ lz_loose,
                                                       const uint32_t* __restrict__ lz_neg, uint32_t* __restrict__ gcount,
                                                       int32_t* __restrict__ gpid, int gcap, int Q, int Qp) {
  const int b = blockIdx.y;
  __shared__ float s_slack;
  // a thread's first element is fetched before the prologue (slack, threshold scan): the kernel is a chain of dependent first-touch
  // loads, and with one element per thread this takes the two at its end off the chain
  const int64_t beg = cand_off[b], n = cand_off[b + 1] - beg;
  const int64_t i_pre = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float a_pre = 0.f;
  int32_t p_pre = 0;
  if (i_pre < n) { a_pre = approx[beg + i_pre]; p_pre = cand_pid[beg + i_pre]; }
  const float qslack = lz_query_slack(lz_tight, lz_loose, lz_neg, b, Q, Qp, &s_slack);
  uint32_t* s = st + (int64_t)b * 8;
  const bool need = s[0] != 0;
  uint32_t kstar = 0;
  if (need) {
    uint32_t k_rem = ext[(int64_t)b * 4 + 3];
    kstar = ext[(int64_t)b * 4 + 2];
    sel_scan_local<2>(hist + ((int64_t)2 * gridDim.y + b) * FP_SEL_BINS, kstar, k_rem);
    if (blockIdx.x == 0 && threadIdx.x == 0) { s[2] = kstar; s[3] = k_rem; }
  }
  const uint32_t keep = s[1];
  int32_t* op = sel_pid + (int64_t)b * R;
  float* oa = sel_approx + (int64_t)b * R;
  uint32_t thr_lo = 0u, thr_hi = 0xFFFFFFFFu;
  if (need) {
    const float U = unmono32(kstar);
    const float d = lz_delta(qslack, U, Q);
    const float lo = U - d, hi = U + d;
    if (lo == lo && hi == hi) { thr_lo = mono32(lo); thr_hi = mono32(hi); }   // (NaN: everything is a maybe -> overflow -> the eager form)
  }
  // (one atomic per wave and class instead of one per element: the ~1000 certain documents of a query all draw their slots from
  // one address, and same-address device atomics retire at ~6 - 10 ns apiece)
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63); i0 < n; i0 += (int64_t)gridDim.x * 256) {
    const int64_t i = i0 + lane;
    const bool valid = i < n;
    const bool first = i == i_pre;
    const float a = first ? a_pre : (valid ? approx[beg + i] : 0.f);
    if (!need) {  // everything is kept, already in ascending doc-id order
      if (valid && i < (int64_t)keep) { op[i] = first ? p_pre : cand_pid[beg + i]; oa[i] = a; }
      continue;
    }
    const uint32_t key = mono32(a);
    const bool in = valid && key > thr_hi;
    const bool mb = valid && !in && key >= thr_lo;
    const unsigned long long m_in = __ballot(in), m_mb = __ballot(mb);
    if (m_in) {
      uint32_t base = 0u;
      if (lane == __builtin_ctzll(m_in)) base = atomicAdd(&s[5], (uint32_t)__builtin_popcountll(m_in));
      base = __shfl(base, __builtin_ctzll(m_in), 64);
      const uint32_t pos = base + (uint32_t)__builtin_popcountll(m_in & lt);
      if (in && pos < keep) { op[pos] = first ? p_pre : cand_pid[beg + i]; oa[pos] = a; }
    }
    if (m_mb) {
      uint32_t base = 0u;
      if (lane == __builtin_ctzll(m_mb)) base = atomicAdd(&gcount[b], (uint32_t)__builtin_popcountll(m_mb));
      base = __shfl(base, __builtin_ctzll(m_mb), 64);
      const uint32_t pos = base + (uint32_t)__builtin_popcountll(m_mb & lt);
      if (mb && pos < (uint32_t)gcap) gpid[(int64_t)b * gcap + pos] = first ? p_pre : cand_pid[beg + i];
    }
  }
}

// the oracle's approximate score of every maybe, one wave per document -- recomputing only what can matter.  For query column q
// the true maximum over the document's codes is attained by an entry whose STORED value lies within the column's slack of the
// stored maximum s (an entry stored below s - slack(s) is truly below s1_lower16(s) <= the true value of the stored argmax), and
// that is one entry per column, rarely two.  So:
//   1. the stored column maxima (the document's score rows of S: lane = (one of 16 codes, 8 columns), as in k_approx);
//   2. the (code, column) pairs within the slack of their column's maximum -> an LDS list (~35 of a document's ~1000 entries);
//   3. lane i re-evaluates pair i with the reference's ascending chain, both rows straight from L2 (32 loads in flight, then 128
//      dependent v_fma_mix: the only serial part), fp16 rounding;
//   4. LDS maxima per column, ascending fp32 sum over the real columns: exactly the oracle's value.
// (Earlier forms, cfg2, ~84 maybes per query, every entry re-evaluated: rows from global memory per code 168 us per batch; lane =
// code with the query row through the scalar cache 198 us; rows staged in LDS, two interleaved chains per lane 90 us -- LDS-bound.)
#define LZ_PAIRS 256   // pair list per wave (a document with more -- masses of tied entries -- raises the batch's overflow flag)
__global__ __launch_bounds__(256) void k_lz_exact(const uint32_t* __restrict__ st, const uint32_t* __restrict__ gcount, const int32_t* __restrict__ gpid,
                                                  float* __restrict__ gval, int gcap, const uint16_t* __restrict__ cent,
                                                  const uint16_t* __restrict__ qpad, int D, int Q, int Qp, const int64_t* __restrict__ uoff,
                                                  const int32_t* __restrict__ ucodes, const uint16_t* __restrict__ S, int64_t C,
                                                  const float* __restrict__ wcol, float kappa, int32_t* __restrict__ flag, uint32_t* __restrict__ stats) {
  __shared__ uint32_t s_pairs[4][LZ_PAIRS];      // code position << 8 | column
  __shared__ uint32_t s_colmax[4][128];          // mono32 keys of the columns' true maxima
  __shared__ uint32_t s_npair[4];
  // grid (queries, workgroups per query): the workgroups that have a maybe -- the first ~20 of a query's -- come FIRST in dispatch
  // order for all queries, the empty ones drain behind them (with the query as the slow index every round of resident
  // workgroups was two thirds empty ones: at 176 registers a CU holds two, and 4096 of them took eight rounds)
  const int b = blockIdx.x;
  if (!st[(int64_t)b * 8]) return;
  const uint32_t nm0 = gcount[b];
  const uint32_t nmay = nm0 < (uint32_t)gcap ? nm0 : (uint32_t)gcap;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t wid = blockIdx.y * 4u + (uint32_t)wave;
  const int ngrp = Qp / 32;
  const uint16_t* Sb = S + (int64_t)b * C * Qp;
  for (uint32_t m = wid; m < nmay; m += gridDim.y * 4u) {
    const int32_t pid = gpid[(int64_t)b * gcap + m];
    const int64_t u0 = uoff[pid];
    const int len = (int)(uoff[pid + 1] - u0);
    if (lane == 0) s_npair[wave] = 0u;
    for (int j = lane; j < Qp; j += 64) s_colmax[wave][j] = mono32(NEG_MASK_F);
    // steps 1 and 2 read the document's score rows the way k_approx does: lane = (one of 16 codes, 16-byte piece = 8 columns), so a
    // 33-code document is three independent loads per lane (a loop over the codes with one 2-byte load per lane and iteration was
    // 17 dependent round trips per pass: 92 us per cfg2 batch); documents of up to 64 codes keep the rows in registers for step 2
    const int cs = lane >> 2, pc = lane & 3;
    const int32_t code0 = lane < len ? ucodes[u0 + lane] : 0;
#pragma unroll 1
    for (int g = 0; g < ngrp; ++g) {
      const uint16_t* Sg = Sb + g * 32 + pc * 8;
      float wv[8];   // (the columns' windows: fetched with the score rows, used behind their reduction)
#pragma unroll
      for (int k = 0; k < 8; ++k) wv[k] = wcol[(int64_t)b * Qp + g * 32 + pc * 8 + k];
      uint32_t mxp[4] = {0xF0E2F0E2u, 0xF0E2F0E2u, 0xF0E2F0E2u, 0xF0E2F0E2u};   // packed fp16 -10000
      uint4 keepv[4];
      int32_t keepc[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        keepc[it] = __shfl(code0, it * 16 + cs, 64);
        keepv[it] = make_uint4(0xF0E2F0E2u, 0xF0E2F0E2u, 0xF0E2F0E2u, 0xF0E2F0E2u);
        if (it * 16 + cs < len) keepv[it] = *reinterpret_cast<const uint4*>(Sg + (int64_t)keepc[it] * Qp);
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        mxp[0] = pk_max_raw(mxp[0], keepv[it].x); mxp[1] = pk_max_raw(mxp[1], keepv[it].y);
        mxp[2] = pk_max_raw(mxp[2], keepv[it].z); mxp[3] = pk_max_raw(mxp[3], keepv[it].w);
      }
#pragma unroll 1
      for (int cb = 64; cb < len; cb += 64) {   // (documents of more than 64 codes)
        const int32_t mycode = cb + lane < len ? ucodes[u0 + cb + lane] : 0;
#pragma unroll 1
        for (int it = 0; it < 4; ++it) {
          const int32_t code = __shfl(mycode, it * 16 + cs, 64);
          if (cb + it * 16 + cs < len) {
            const uint4 v = *reinterpret_cast<const uint4*>(Sg + (int64_t)code * Qp);
            mxp[0] = pk_max_raw(mxp[0], v.x); mxp[1] = pk_max_raw(mxp[1], v.y); mxp[2] = pk_max_raw(mxp[2], v.z); mxp[3] = pk_max_raw(mxp[3], v.w);
          }
        }
      }
#pragma unroll
      for (int x = 4; x < 64; x <<= 1)
#pragma unroll
        for (int k = 0; k < 4; ++k) mxp[k] = pk_max_raw(mxp[k], shfl_xor_u32(mxp[k], x));
      // 2. the entries within their column's slack of the maximum -> (code << 8 | column) pairs
      float lo[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int col = g * 32 + pc * 8 + k;
        const uint16_t hb = (uint16_t)(mxp[k >> 1] >> (16 * (k & 1)));
        const float smax = (float)__builtin_bit_cast(half_t, hb);
        const float w = wv[k];
        uint32_t e = hb & 0x7C00u;
        e = (e < 0x2C00u ? 0x2C00u : e) - 0x2800u;
        lo[k] = (w > 0.f) ? smax - ((float)__builtin_bit_cast(half_t, (uint16_t)e) + 2.f * s1_u2(__builtin_fabsf(smax), w, kappa)) * 1.0001f : __builtin_inff();
        // a zero query row (w == 0: its window is empty, every stored score of the column is the reference's exact 0): the stored
        // maximum IS the true one -- no pairs (with lo = smax every code of the document tied at 0 and a handful of zero-padded
        // rows overflowed the pair list: two voided batches, then the scratch went eager)
        if (!(w > 0.f) && col < Q && cs == 0) atomicMax(&s_colmax[wave][col], mono32(smax));
        if (col >= Q) lo[k] = __builtin_inff();   // (pad columns: no pairs)
      }
      // (compact code on purpose: this kernel runs a handful of waves once through straight-line code, i.e. at the speed of its
      // instruction fetches -- 12 KB of code took 49 us whatever the work, ~260 ns per 64-byte line)
      auto emit = [&](const uint4 v, const int32_t code) {
        const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
        uint32_t mask = 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          mask |= ((float)__builtin_bit_cast(half_t, (uint16_t)(vw[k >> 1] >> (16 * (k & 1)))) >= lo[k]) ? (1u << k) : 0u;
        while (mask) {   // (rare: about one entry per column and document)
          const int k = __builtin_ctz(mask);
          mask &= mask - 1u;
          const uint32_t pos = atomicAdd(&s_npair[wave], 1u);
          if (pos < LZ_PAIRS) s_pairs[wave][pos] = ((uint32_t)code << 8) | (uint32_t)(g * 32 + pc * 8 + k);
        }
      };
#pragma unroll
      for (int it = 0; it < 4; ++it)
        if (it * 16 + cs < len) emit(keepv[it], keepc[it]);
#pragma unroll 1
      for (int cb = 64; cb < len; cb += 64) {
        const int32_t mycode = cb + lane < len ? ucodes[u0 + cb + lane] : 0;
#pragma unroll 1
        for (int it = 0; it < 4; ++it) {
          const int32_t code = __shfl(mycode, it * 16 + cs, 64);
          if (cb + it * 16 + cs < len) emit(*reinterpret_cast<const uint4*>(Sg + (int64_t)code * Qp), code);
        }
      }
    }
    // (one wave: its LDS operations complete in order; the fence keeps the compiler from moving the reads above the writes)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint32_t np = s_npair[wave];
    if (stats && lane == 0) atomicAdd(&stats[3], np);
    if (np > LZ_PAIRS) { np = LZ_PAIRS; if (lane == 0) *flag = 1; }
    // 3. one chain per lane and pair; up to 128 dims of both rows (sixteen 16-byte pieces each) in flight at a time
#pragma unroll 1
    for (uint32_t i0 = 0; i0 < np; i0 += 64) {
      const uint32_t i = i0 + (uint32_t)lane;
      if (i < np) {
        const uint32_t pr = s_pairs[wave][i];
        const int col = (int)(pr & 0xFFu);
        const uint16_t* crow = cent + (int64_t)(pr >> 8) * D;
        const uint16_t* qrow = qpad + ((int64_t)b * Qp + col) * D;
        float ch = 0.f;
#pragma unroll 1
        for (int k0 = 0; k0 < D; k0 += 64) {   // (64 dims of both rows in flight: 16 loads, ~110 registers, four waves per SIMD)
          uint4 cv[8], qv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const bool in = k0 + j * 8 < D;
            cv[j] = in ? *reinterpret_cast<const uint4*>(crow + k0 + j * 8) : make_uint4(0, 0, 0, 0);
            qv[j] = in ? *reinterpret_cast<const uint4*>(qrow + k0 + j * 8) : make_uint4(0, 0, 0, 0);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (k0 + j * 8 < D) s1_chain8(ch, cv[j], qv[j]);
        }
        atomicMax(&s_colmax[wave][col], mono32((float)(half_t)ch));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // 4. ascending fp32 sum over the real columns (search.rs:401 as the oracle fixes it), the same in every lane
    float total = 0.f;
#pragma unroll 1
    for (int j = 0; j < Q; ++j) total += unmono32(s_colmax[wave][j]);
    if (lane == 0) gval[(int64_t)b * gcap + m] = total;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

__global__ __launch_bounds__(1024) void k_sel_finish_lz(uint32_t* __restrict__ st, int64_t R, int32_t* __restrict__ sel_pid,
                                                        float* __restrict__ sel_approx, const uint32_t* __restrict__ gcount,
                                                        const int32_t* __restrict__ gpid, const float* __restrict__ gval, int gcap,
                                                        int32_t* __restrict__ flag, uint32_t* __restrict__ stats,
                                                        const int32_t* __restrict__ sel_cnt, int64_t* __restrict__ pref /*nullable*/, int B) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t s_bad;
  const int b = blockIdx.x, tid = threadIdx.x;
  // (the count prefix k_sel_collect writes in the eager form: sel_cnt is final since k_sel_hist<0>; the lazy form never takes the
  // ordered collection -- an overflow voids the batch -- so that launch is not enqueued here.  A workgroup of its own, behind the
  // queries': inside one of theirs its ten scan rounds were the kernel's critical path)
  if (b == B) {
    if (pref) sel_count_prefix(sel_cnt, pref, B);
    return;
  }
  uint32_t* s = st + (int64_t)b * 8;
  int32_t* op = sel_pid + (int64_t)b * R;
  float* oa = sel_approx + (int64_t)b * R;
  const int32_t* gp = gpid + (int64_t)b * gcap;
  const float* gv = gval + (int64_t)b * gcap;
  // every first-touch load up front, unconditionally (slots past the counts hold stale values and are masked below): one round
  // trip instead of counts -> lists
  const uint32_t need = s[0], keep = s[1], nin_all = s[5], nm_all = gcount[b];
  const int32_t p_op = (int64_t)tid < R ? op[tid] : 0;
  const float p_oa = (int64_t)tid < R ? oa[tid] : 0.f;
  const int32_t p_gp = tid < gcap ? gp[tid] : 0;
  const float p_gv = tid < gcap ? gv[tid] : 0.f;
  if (!need) return;  // nothing selected away
  const uint32_t nmay = nm_all < (uint32_t)gcap ? nm_all : (uint32_t)gcap;
  const uint32_t nin = nin_all < keep ? nin_all : keep;
  int kp2 = 2;
  while (kp2 < (int)keep) kp2 <<= 1;
  unsigned long long* kv = reinterpret_cast<unsigned long long*>(smem);                    // [kp2] the selection: id << 32 | score bits
  unsigned long long* mk = kv + kp2;                                                        // [max(1024, pow2(gcap))] maybes: mono32(score) << 32 | ~id
  // (more maybes than the lists hold, more certain ones than the selection -- impossible: fewer than keep scores exceed U --, or
  // too few entries altogether -- impossible unless a list overflowed: at least keep scores are >= U)
  if (tid == 0) s_bad = (nm_all > (uint32_t)gcap || nin_all >= keep + 1u || nin + nmay < keep) ? 1u : 0u;
  for (int i = tid; i < kp2; i += 1024) {
    const bool have = (uint32_t)i < nin;   // (slots past the row's R entries exist in kv only: kp2 is R rounded up to a power of two)
    const int32_t pi = i < 1024 ? p_op : (have ? op[i] : 0);
    const float ai = i < 1024 ? p_oa : (have ? oa[i] : 0.f);
    kv[i] = have ? (((unsigned long long)(uint32_t)pi << 32) | (unsigned long long)__float_as_uint(ai)) : ~0ull;
  }
  // best (keep - nin) maybes by (score desc, id asc)
  const uint32_t take = keep - nin;
  auto mkey = [&](uint32_t i) -> unsigned long long {
    const int32_t pi = i < 1024u ? p_gp : (i < nmay ? gp[i] : 0);
    const float vi = i < 1024u ? p_gv : (i < nmay ? gv[i] : 0.f);
    return i < nmay ? (((unsigned long long)mono32(vi) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)pi)) : 0ull;
  };
  if (nmay <= 256u) {    // (the usual case: a few dozen) rank by counting: every key against every other, broadcast reads of LDS
    const unsigned long long mine = mkey((uint32_t)tid);   // (tid < 1024: the prefetched slot is this thread's own)
    if ((uint32_t)tid < nmay) mk[tid] = mine;
    __syncthreads();
    if ((uint32_t)tid < nmay) {
      uint32_t rank = 0;
      for (uint32_t j = 0; j < nmay; ++j) rank += mk[j] > mine ? 1u : 0u;   // (keys are distinct: the ids are)
      if (rank < take) {
        const uint32_t pid = 0xFFFFFFFFu - (uint32_t)mine;
        kv[nin + rank] = ((unsigned long long)pid << 32) | (unsigned long long)__float_as_uint(unmono32((uint32_t)(mine >> 32)));
      }
    }
  } else
  if (nmay <= 1024u) {   // (the usual case: a few dozen) one key per thread, shuffle network
    const unsigned long long mine = mkey((uint32_t)tid);
    __syncthreads();
    const unsigned long long sorted = fp_sort1024_desc(mine, mk);
    __syncthreads();
    if ((uint32_t)tid < take && (uint32_t)tid < nmay) {
      const uint32_t pid = 0xFFFFFFFFu - (uint32_t)sorted;
      kv[nin + tid] = ((unsigned long long)pid << 32) | (unsigned long long)__float_as_uint(unmono32((uint32_t)(sorted >> 32)));
    }
  } else {               // large rerank lists: bitonic sort in LDS
    int mp2 = 2048;
    while (mp2 < (int)nmay) mp2 <<= 1;
    for (int i = tid; i < mp2; i += 1024) mk[i] = mkey((uint32_t)i);
    __syncthreads();
    for (int k = 2; k <= mp2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < mp2; i += 1024) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const unsigned long long x = mk[i], y = mk[ixj];
            const bool desc = ((i & k) == 0);
            if ((x < y) == desc) { mk[i] = y; mk[ixj] = x; }
          }
        }
        __syncthreads();
      }
    }
    for (uint32_t i = tid; i < take && i < nmay; i += 1024) {
      const unsigned long long sorted = mk[i];
      const uint32_t pid = 0xFFFFFFFFu - (uint32_t)sorted;
      kv[nin + i] = ((unsigned long long)pid << 32) | (unsigned long long)__float_as_uint(unmono32((uint32_t)(sorted >> 32)));
    }
  }
  __syncthreads();
  if (tid == 0) {
    if (s_bad) *flag = 1;
    if (stats) { atomicAdd(&stats[0], nin_all); atomicAdd(&stats[1], nm_all); atomicMax(&stats[2], nm_all); }
  }
  // whole selection ascending by doc id (the order every later stage expects)
  if (kp2 == 1024) {
    const unsigned long long srt = ~fp_sort1024_desc(~kv[tid], kv);
    kv[tid] = srt;
    __syncthreads();
  } else {
    for (int k = 2; k <= kp2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < kp2; i += 1024) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const unsigned long long a = kv[i], c = kv[ixj];
            const bool up = ((i & k) == 0);
            if ((a > c) == up) { kv[i] = c; kv[ixj] = a; }
          }
        }
        __syncthreads();
      }
    }
  }
  for (int i = tid; i < (int)keep; i += 1024) {
    // (a failed batch -- s_bad -- may hold filler keys: clamp the ids so that nothing downstream indexes out of range; its
    // results are discarded by the caller)
    const unsigned long long k = kv[i];
    op[i] = (k == ~0ull) ? 0 : (int32_t)(k >> 32);
    oa[i] = __uint_as_float((uint32_t)k);
  }
}

// k_sel_init + the three histogram / scan rounds + k_sel_gather in ONE workgroup per query, for lists of a few thousand entries
// (the survivors of S4's bound stages): nine launches of ~5 us each become one.  Same selstate / output contract as the
// separate kernels; k_sel_finish / k_sel_collect follow unchanged.
#define SEL_FRONT_REGS 16
__global__ __launch_bounds__(1024) void k_sel_front(const float* __restrict__ approx, const int32_t* __restrict__ cand_pid,
                                                    const int64_t* __restrict__ cand_off, int64_t n_full, int64_t R,
                                                    uint32_t* __restrict__ st, int32_t* __restrict__ sel_pid, float* __restrict__ sel_approx,
                                                    int32_t* __restrict__ tie_pid, int32_t* __restrict__ sel_cnt) {
  __shared__ uint32_t h[FP_SEL_BINS];
  __shared__ uint32_t sc[256];
  __shared__ uint32_t s_prefix, s_rem, s_gt, s_eq, s_ovf, s_lo, s_hi;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t beg = cand_off[b], n = cand_off[b + 1] - beg;
  int64_t keep = n;
  if (n_full < keep) keep = n_full;   // search.rs:605-611
  if (R < keep) keep = R;             // :614-619
  if (keep < 0) keep = 0;
  uint32_t* s = st + (int64_t)b * 8;
  int32_t* op = sel_pid + (int64_t)b * R;
  float* oa = sel_approx + (int64_t)b * R;
  int32_t* tp = tie_pid + (int64_t)b * R;
  const bool need = keep < n;
  const bool in_regs = n <= (int64_t)SEL_FRONT_REGS * 1024;   // (uniform) the list fits the workgroup's registers
  uint32_t kreg[SEL_FRONT_REGS];
  if (tid == 0) {
    s[0] = need ? 1u : 0u; s[1] = (uint32_t)keep; s[4] = (uint32_t)n;
    sel_cnt[b] = (int32_t)keep;
    s_prefix = 0u; s_rem = (uint32_t)keep; s_gt = 0u; s_eq = 0u; s_ovf = 0u; s_lo = 0xFFFFFFFFu; s_hi = 0u;
  }
  if (!need) {   // everything is kept, already in ascending doc-id order
    for (int64_t i = tid; i < keep; i += 1024) { op[i] = cand_pid[beg + i]; oa[i] = approx[beg + i]; }
    if (tid == 0) { s[2] = 0u; s[3] = (uint32_t)keep; s[5] = 0u; s[6] = 0u; s[7] = 0u; }
    return;
  }
  __syncthreads();
  // Radix select over the RANGE of the keys, 11 bits per round: bin = (key - lo) >> shift.  (Fixed digits of the raw key put
  // nearly every element into one or two bins in the first round -- the scores of a query share sign, exponent and leading
  // mantissa bits -- and same-address LDS atomics serialise: 33 us for 6 k elements.)
  {
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    if (in_regs) {   // all loads of the thread in flight together; the later rounds never touch memory
#pragma unroll
      for (int c = 0; c < SEL_FRONT_REGS; ++c) {
        const int64_t i = tid + (int64_t)c * 1024;
        kreg[c] = i < n ? mono32(approx[beg + i]) : 0u;
      }
#pragma unroll
      for (int c = 0; c < SEL_FRONT_REGS; ++c) {
        if (tid + (int64_t)c * 1024 < n) {
          kmin = kreg[c] < kmin ? kreg[c] : kmin;
          kmax = kreg[c] > kmax ? kreg[c] : kmax;
        }
      }
    } else {
      for (int64_t i = tid; i < n; i += 1024) {
        const uint32_t k = mono32(approx[beg + i]);
        kmin = k < kmin ? k : kmin;
        kmax = k > kmax ? k : kmax;
      }
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) {
      const uint32_t a = (uint32_t)__shfl_xor((int)kmin, sft, 64), c = (uint32_t)__shfl_xor((int)kmax, sft, 64);
      kmin = a < kmin ? a : kmin;
      kmax = c > kmax ? c : kmax;
    }
    if ((tid & 63) == 0) { atomicMin(&s_lo, kmin); atomicMax(&s_hi, kmax); }
  }
  __syncthreads();
  uint32_t lo = s_lo;
  int shift = 0;
  {
    const uint32_t span = s_hi - lo;   // largest offset
    const int bits = span ? 32 - __builtin_clz(span) : 1;
    shift = bits > 11 ? bits - 11 : 0;
  }
  uint32_t nb_valid = FP_SEL_BINS;
  for (;;) {   // at most three rounds
    for (int i = tid; i < FP_SEL_BINS; i += 1024) h[i] = 0u;
    __syncthreads();
    if (in_regs) {
#pragma unroll
      for (int c = 0; c < SEL_FRONT_REGS; ++c) {
        const uint32_t k = kreg[c];
        if (tid + (int64_t)c * 1024 < n && k >= lo && ((k - lo) >> shift) < nb_valid) atomicAdd(&h[(k - lo) >> shift], 1u);
      }
    } else {
      for (int64_t i = tid; i < n; i += 1024) {
        const uint32_t k = mono32(approx[beg + i]);
        if (k >= lo && ((k - lo) >> shift) < nb_valid) atomicAdd(&h[(k - lo) >> shift], 1u);
      }
    }
    __syncthreads();
    // 256 threads own 8 bins each from the top; inclusive scan of their sums; the thread whose bins hold the k_rem-th element
    uint32_t loc[8];
    uint32_t sum = 0;
    if (tid < 256) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        loc[j] = h[FP_SEL_BINS - 1 - (tid * 8 + j)];
        sum += loc[j];
      }
      sc[tid] = sum;
    }
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const uint32_t t = (tid < 256 && tid >= off) ? sc[tid - off] : 0u;
      __syncthreads();
      if (tid < 256) sc[tid] += t;
      __syncthreads();
    }
    if (tid < 256) {
      const uint32_t k_rem = s_rem;
      const uint32_t incl = sc[tid], excl = incl - sum;
      if (excl < k_rem && incl >= k_rem) {   // exactly one thread
        uint32_t cum = excl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (cum + loc[j] >= k_rem) {
            s_prefix = (uint32_t)(FP_SEL_BINS - 1 - (tid * 8 + j));
            s_rem = k_rem - cum;   // still needed from inside this bin
            break;
          }
          cum += loc[j];
        }
      }
    }
    __syncthreads();
    lo += s_prefix << shift;
    if (shift == 0) break;
    const int nshift = shift > 11 ? shift - 11 : 0;
    nb_valid = 1u << (shift - nshift);   // the chosen bin, split again
    shift = nshift;
    __syncthreads();
  }
  if (tid == 0) s_prefix = lo;
  __syncthreads();
  const uint32_t kstar = s_prefix;
  auto emit = [&](int64_t i, uint32_t key) {
    if (key > kstar) {
      const uint32_t pos = atomicAdd(&s_gt, 1u);
      if (pos < (uint32_t)keep) { op[pos] = cand_pid[beg + i]; oa[pos] = unmono32(key); }
    } else if (key == kstar) {
      const uint32_t pos = atomicAdd(&s_eq, 1u);
      if (pos < (uint32_t)R) tp[pos] = cand_pid[beg + i]; else s_ovf = 1u;
    }
  };
  if (in_regs) {
#pragma unroll
    for (int c = 0; c < SEL_FRONT_REGS; ++c) {
      const int64_t i = tid + (int64_t)c * 1024;
      if (i < n) emit(i, kreg[c]);
    }
  } else {
    for (int64_t i = tid; i < n; i += 1024) emit(i, mono32(approx[beg + i]));
  }
  __syncthreads();
  if (tid == 0) { s[2] = kstar; s[3] = s_rem; s[5] = s_gt; s[6] = s_eq; s[7] = s_ovf; }
}

// marks every query for the ordered collection (k_sel_collect): rerank lists too long for the LDS sort of k_sel_finish
__global__ void k_sel_force_collect(uint32_t* __restrict__ st, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) st[(int64_t)b * 8 + 7] = 1u;
}
// the selection and the maybes' sort share the LDS of one workgroup: (pow2(R) + max(1024, pow2(maybe capacity))) x 8 bytes
bool fpk_select_lazy_ok(const FpSearchShape& sh) { return sh.R <= 8192; }
int fpk_select_lazy_gcap(const FpSearchShape& sh) {   // room for the maybes of one query (their number grows with the density of scores at the cut, i.e. with R)
  static const int env = (int)fp_test_opt("lz_gcap", 0);   // tests: a small list forces the eager re-run
  if (env > 0) return env;
  return (int)std::max<int64_t>(1024, fp_next_pow2((int)sh.R));
}
size_t fpk_sel_hist_bytes(int B) { return (size_t)3 * B * FP_SEL_BINS * sizeof(uint32_t) + (size_t)B * 16; }   // three histograms per query + {prefix, k_rem} x 2
int fpk_select(const FpSearchShape& sh, const int64_t* cand_off, const int32_t* cand_pid, const float* approx, uint32_t* hist,
                uint32_t* selstate, int32_t* sel_pid, float* sel_approx, int32_t* sel_cnt, int32_t* tie_pid, hipStream_t st,
                bool short_lists, bool hist_prezeroed, int64_t* pref, const FpLazyS1* lz, const FpIndexDev* ixp, int64_t est_per_query) {
  const int B = sh.B;
  // workgroups per query of the histogram / gather passes: every one of them zeroes and flushes a 2048-bin histogram and repeats
  // the previous pass's scan, so a list of a few thousand survivors wants a handful, not 64 (grid-stride loops: any number works)
  unsigned gx = 64;
  if (est_per_query > 0) {
    const int64_t w = (est_per_query + 1023) / 1024;
    gx = (unsigned)(w < 4 ? 4 : w > 64 ? 64 : w);
  }
  gx = (unsigned)fp_test_opt("sel_gx", gx);
  if (gx < 1) gx = 1;
  // the gather: one element per thread where the chip has room (its prologue is the same in every workgroup, but its loop ends in
  // dependent loads and a slot atomic per wave)
  unsigned gg = 64;
  if (est_per_query > 0) {
    const int64_t w = (est_per_query + 255) / 256;
    gg = (unsigned)(w < 4 ? 4 : w > 64 ? 64 : w);
  }
  gg = (unsigned)fp_test_opt("sel_gg", gg);
  if (gg < 1) gg = 1;
  const uint16_t* ix_cent = ixp ? ixp->centroids : nullptr;
  const int ix_dim = ixp ? ixp->dim : 0;
  const int64_t* ix_uoff = ixp ? ixp->uoff : nullptr;
  const int32_t* ix_ucodes = ixp ? ixp->ucodes : nullptr;
  // the lazy form's approx[] / S hold UPPER candidates: only the general path settles them.  A caller that asks for it on a shape
  // that path does not serve (run_front gates on fpk_select_lazy_ok) would get a silently wrong selected set: refuse.
  if (lz && (short_lists || sh.R > FP_MAX_SORT || !ixp)) return 1;
  if (short_lists) {
    hipLaunchKernelGGL(k_sel_front, dim3((unsigned)B), dim3(1024), 0, st, approx, cand_pid, cand_off, sh.n_full, sh.R, selstate, sel_pid, sel_approx,
                       tie_pid, sel_cnt);
    static std::atomic<uint64_t> lds_ok2{0};
    fp_allow_big_lds((const void*)k_sel_finish, lds_ok2, 144 * 1024);
    int kp2 = 2;
    while (kp2 < (int)sh.R) kp2 <<= 1;
    hipLaunchKernelGGL(k_sel_finish, dim3((unsigned)B), dim3(1024), (size_t)kp2 * 8, st, selstate, sh.R, sel_pid, sel_approx, tie_pid);
    hipLaunchKernelGGL(k_sel_collect, dim3((unsigned)B), dim3(1024), 0, st, approx, cand_pid, cand_off, selstate, sh.R, sel_pid, sel_approx,
                       sel_cnt, pref, B);
    return 0;
  }
  const bool big = sh.R > FP_MAX_SORT;   // beyond k_sel_finish's LDS sort: the ordered single-workgroup collection does the whole job
  if (!hist_prezeroed) (void)hipMemsetAsync(hist, 0, fpk_sel_hist_bytes(B), st);
  if (lz) {   // (the caller asks for this only when !short_lists and !big)
    uint32_t* ext = hist + (size_t)3 * B * FP_SEL_BINS;
    dim3 gh(gx, (unsigned)B);
    hipLaunchKernelGGL(k_sel_hist<0>, gh, dim3(256), 0, st, approx, cand_off, selstate, hist, ext, sh.n_full, sh.R, sel_cnt);
    hipLaunchKernelGGL(k_sel_hist<1>, gh, dim3(256), 0, st, approx, cand_off, selstate, hist, ext, sh.n_full, sh.R, sel_cnt);
    hipLaunchKernelGGL(k_sel_hist<2>, gh, dim3(256), 0, st, approx, cand_off, selstate, hist, ext, sh.n_full, sh.R, sel_cnt);
    hipLaunchKernelGGL(k_sel_gather_lz, dim3(gg, (unsigned)B), dim3(256), 0, st, approx, cand_pid, cand_off, selstate, sh.R, sel_pid, sel_approx,
                       hist, ext, lz->tight, lz->loose, lz->negflag, lz->gcount, lz->gpid, lz->gcap, sh.Q, sh.Qp);
    static const int lz_exb = (int)fp_test_opt("lz_exb", LZ_EX_BLOCKS);
    hipLaunchKernelGGL(k_lz_exact, dim3((unsigned)B, (unsigned)(lz_exb < 1 ? 1 : lz_exb)), dim3(256), 0, st, selstate, lz->gcount, lz->gpid, lz->gval, lz->gcap, ix_cent,
                       lz->qpad, ix_dim, sh.Q, sh.Qp, ix_uoff, ix_ucodes, lz->S, ixp->C, lz->wcol, lz->kappa, lz->flag, lz->stats);
    static std::atomic<uint64_t> lds_ok3{0};
    fp_allow_big_lds((const void*)k_sel_finish_lz, lds_ok3, 152 * 1024);
    int kp2 = 2;
    while (kp2 < (int)sh.R) kp2 <<= 1;
    int mp2 = 1024;
    while (mp2 < lz->gcap) mp2 <<= 1;
    hipLaunchKernelGGL(k_sel_finish_lz, dim3((unsigned)B + (pref ? 1u : 0u)), dim3(1024), (size_t)kp2 * 8 + (size_t)mp2 * 8, st, selstate, sh.R, sel_pid,
                       sel_approx, lz->gcount, lz->gpid, lz->gval, lz->gcap, lz->flag, lz->stats, sel_cnt, pref, B);
    return 0;
  }
  uint32_t* ext = hist + (size_t)3 * B * FP_SEL_BINS;   // [B][4], behind the histograms (fpk_sel_hist_bytes)
  dim3 gh(gx, (unsigned)B);
  hipLaunchKernelGGL(k_sel_hist<0>, gh, dim3(256), 0, st, approx, cand_off, selstate, hist, ext, sh.n_full, sh.R, sel_cnt);
  hipLaunchKernelGGL(k_sel_hist<1>, gh, dim3(256), 0, st, approx, cand_off, selstate, hist, ext, sh.n_full, sh.R, sel_cnt);
  hipLaunchKernelGGL(k_sel_hist<2>, gh, dim3(256), 0, st, approx, cand_off, selstate, hist, ext, sh.n_full, sh.R, sel_cnt);
  if (big) {
    hipLaunchKernelGGL(k_sel_scan_final, dim3((unsigned)B), dim3(256), 0, st, hist, selstate, ext, B);
    hipLaunchKernelGGL(k_sel_force_collect, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, selstate, B);
  } else {
    hipLaunchKernelGGL(k_sel_gather, dim3(gg, (unsigned)B), dim3(256), 0, st, approx, cand_pid, cand_off, selstate, sh.R, sel_pid, sel_approx,
                       tie_pid, hist, ext);
  }
  if (!big) {
    static std::atomic<uint64_t> lds_ok{0};
    fp_allow_big_lds((const void*)k_sel_finish, lds_ok, 144 * 1024);
    int kp2 = 2;
    while (kp2 < (int)sh.R) kp2 <<= 1;
    hipLaunchKernelGGL(k_sel_finish, dim3((unsigned)B), dim3(1024), (size_t)kp2 * 8, st, selstate, sh.R, sel_pid, sel_approx, tie_pid);
  }
  // ordered single-block fallback, only for queries whose tie buffer overflowed (e.g. all scores equal)
  hipLaunchKernelGGL(k_sel_collect, dim3((unsigned)B), dim3(1024), 0, st, approx, cand_pid, cand_off, selstate, sh.R, sel_pid,
                     sel_approx, sel_cnt, pref, B);
  return 0;
}

// S6+S7 (fused decompress + exact MaxSim, per-token norms, exact-order repair): fp_maxsim.hip

// ============================================================================================
// S8  final ranking: LDS bitonic sort of 64-bit keys (mono32(score) << 32 | ~id), descending.
// ids must be < 2^32 (global doc ids of a <= 4.29e9-document corpus).
// ============================================================================================
__global__ __launch_bounds__(1024) void k_final_topk(const float* __restrict__ score, const int32_t* __restrict__ pid_local,
                                                     const int64_t* __restrict__ pid_global, const int32_t* __restrict__ cnt,
                                                     int64_t stride, int npow2, int64_t top_k, int64_t pid_offset,
                                                     int64_t* __restrict__ out_pid, float* __restrict__ out_score,
                                                     int32_t* __restrict__ out_cnt, const int64_t* __restrict__ stat_total,
                                                     const int32_t* __restrict__ stat_per_query, int64_t* __restrict__ stat_out,
                                                     const int32_t* __restrict__ stat_flag, const int64_t* __restrict__ stat_cand) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x;
  // a thread's first element before anything else, unconditionally (slots past the count are masked below): the count, the
  // statistics and the lists are then ONE round of first-touch loads instead of three dependent ones
  const int64_t o_pre = (int64_t)b * stride + threadIdx.x;
  const bool in_pre = (int64_t)threadIdx.x < stride;
  const int64_t idg_pre = (in_pre && !pid_local) ? pid_global[o_pre] : 0;
  const int32_t idl_pre = (in_pre && pid_local) ? pid_local[o_pre] : 0;
  const float s_pre = in_pre ? score[o_pre] : 0.f;
  const int n = cnt ? cnt[b] : (int)stride;
  // statistics of the search that travel with the results (one copy to the host instead of three): [0] = *stat_total,
  // [1 + b] = stat_per_query[b], [1 + B] = *stat_flag (the lazy S1's overflow flag)
  if (stat_out && threadIdx.x == 0) {
    stat_out[1 + blockIdx.x] = stat_per_query ? (int64_t)stat_per_query[blockIdx.x] : 0;
    if (blockIdx.x == 0) {
      stat_out[0] = stat_total ? *stat_total : 0;
      stat_out[1 + gridDim.x] = stat_flag ? (int64_t)*stat_flag : 0;
      // S3's block {candidate total, -, the probe's overflow flag (int32 at byte 16)}: saves the host-buffer search a copy node
      stat_out[2 + gridDim.x] = stat_cand ? stat_cand[0] : 0;
      stat_out[3 + gridDim.x] = stat_cand ? (int64_t) * reinterpret_cast<const int32_t*>(stat_cand + 2) : 0;
    }
  }
  // all LDS in the dynamic region (a static __shared__ in front would mis-align the 8-byte keys)
  int* s_validp = reinterpret_cast<int*>(smem);
  unsigned long long* v = reinterpret_cast<unsigned long long*>(smem + 16);
  if (threadIdx.x == 0) *s_validp = 0;
  __syncthreads();
  int myvalid = 0;
  for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
    unsigned long long key = 0ull;
    if (i < n) {
      const bool first = i == (int)threadIdx.x;
      int64_t id = pid_local ? (int64_t)(first ? idl_pre : pid_local[(int64_t)b * stride + i]) + pid_offset
                             : (first ? idg_pre : pid_global[(int64_t)b * stride + i]);
      float s = first ? s_pre : score[(int64_t)b * stride + i];
      if (id >= 0) {  // id < 0 = padding entry of a sharded buffer
        key = ((unsigned long long)mono32(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)id);
        ++myvalid;
      }
    }
    v[i] = key;
  }
  {   // (one LDS atomic per wave: 1024 of them on one address were microseconds)
    int wv = myvalid;
#pragma unroll
    for (int x = 32; x > 0; x >>= 1) wv += __shfl_xor(wv, x, 64);
    if ((threadIdx.x & 63) == 0 && wv) atomicAdd(s_validp, wv);
  }
  __syncthreads();
  if (npow2 == 1024 && blockDim.x == 1024) {   // one key per thread: shuffle-based network
    const unsigned long long sorted = fp_sort1024_desc(v[threadIdx.x], v);
    v[threadIdx.x] = sorted;
    __syncthreads();
  } else {
    for (int k = 2; k <= npow2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
          int ixj = i ^ j;
          if (ixj > i) {
            unsigned long long a = v[i], c = v[ixj];
            bool desc = ((i & k) == 0);
            if ((a < c) == desc) { v[i] = c; v[ixj] = a; }
          }
        }
        __syncthreads();
      }
    }
  }
  const int valid = *s_validp;
  const int m = (int)(top_k < valid ? top_k : valid);
  for (int i = threadIdx.x; i < m; i += blockDim.x) {
    unsigned long long key = v[i];
    out_pid[(int64_t)b * top_k + i] = (int64_t)(0xFFFFFFFFu - (uint32_t)key);
    out_score[(int64_t)b * top_k + i] = unmono32((uint32_t)(key >> 32));
  }
  for (int64_t i = m + threadIdx.x; i < top_k; i += blockDim.x) {   // padding contract (fastplaid.h): id -1, score 0 beyond the count
    out_pid[(int64_t)b * top_k + i] = -1;
    out_score[(int64_t)b * top_k + i] = 0.f;
  }
  if (threadIdx.x == 0) out_cnt[b] = m;
}

__global__ void k_final_stats(const int64_t* __restrict__ stat_total, const int32_t* __restrict__ stat_per_query, int B, int64_t* __restrict__ stat_out,
                              const int32_t* __restrict__ stat_flag, const int64_t* __restrict__ stat_cand) {
  for (int b = threadIdx.x; b < B; b += blockDim.x) stat_out[1 + b] = stat_per_query ? (int64_t)stat_per_query[b] : 0;
  if (threadIdx.x == 0) {
    stat_out[0] = stat_total ? *stat_total : 0;
    stat_out[1 + B] = stat_flag ? (int64_t)*stat_flag : 0;
    stat_out[2 + B] = stat_cand ? stat_cand[0] : 0;
    stat_out[3 + B] = stat_cand ? (int64_t) * reinterpret_cast<const int32_t*>(stat_cand + 2) : 0;
  }
}
int fpk_final_topk(const float* score, const int32_t* pid_local, const int64_t* pid_global, const int32_t* cnt, int64_t stride,
                   int B, int64_t top_k, int64_t pid_offset, int64_t* out_pid, float* out_score, int32_t* out_cnt,
                   hipStream_t st, const int64_t* stat_total, const int32_t* stat_per_query, int64_t* stat_out, const int32_t* stat_flag,
                   const int64_t* stat_cand) {
  int np2 = next_pow2((int)stride);
  if (np2 < 2) np2 = 2;
  if (stride > FP_MAX_SORT) {   // beyond the LDS sort: segmented device radix sort (rare: n_full_scores > 65536)
    if (stat_out) hipLaunchKernelGGL(k_final_stats, dim3(1), dim3(256), 0, st, stat_total, stat_per_query, B, stat_out, stat_flag, stat_cand);
    // (allocates 2 * B * stride keys and synchronises; -1: B * stride does not fit its 32-bit segment offsets -- nothing was written)
    return fps_final_topk_big(score, pid_local, pid_global, cnt, stride, B, top_k, pid_offset, out_pid, out_score, out_cnt, st);
  }
  static std::atomic<uint64_t> lds_ok{0};
  fp_allow_big_lds((const void*)k_final_topk, lds_ok, 144 * 1024);
  hipLaunchKernelGGL(k_final_topk, dim3((unsigned)B), dim3(1024), (size_t)np2 * 8 + 16, st, score, pid_local, pid_global, cnt, stride, np2,
                     top_k, pid_offset, out_pid, out_score, out_cnt, stat_total, stat_per_query, stat_out, stat_flag, stat_cand);
  return 0;
}

// ============================================================================================
// sharded-search helpers.  Both exchanges move ONE fixed-size record buffer per rank (include/fastplaid.h):
//   rec1 {i64 pid; f32 approx; i32 pad}                       local top-R candidates by approximate score
//   rec2 {i64 pid; f32 score; f32 unc_down; f32 unc; i32 pad} scores of the local survivors of the global cut: the MFMA score, its
//                                                             uncertainty budget (unc) and the part of it by which the reference's
//                                                             score may be LOWER (unc_down); record 0's pad carries the rank's status
// all_rec* = [G][B][R] as an all-gather lays them out.  The last stage marks the near-tied flagged documents on the UNION exactly
// as the unsharded search does on its rerank list and takes `exact` for them, `score` for the others: same result bit for bit.
// ============================================================================================
struct ShardRec1 { long long pid; float approx; int pad; };
struct ShardRec2 { long long pid; float score; float uncm; float unc; int pad; };
static_assert(sizeof(ShardRec1) == 16 && sizeof(ShardRec2) == 24, "record layout is part of the C ABI");

__global__ void k_shard_pack1(const float* __restrict__ approx, const int32_t* __restrict__ pid, const int32_t* __restrict__ cnt, int64_t R,
                              int64_t pid_offset, ShardRec1* __restrict__ out, int64_t total,
                              const int64_t* __restrict__ cand_total /*nullable*/, int64_t cand_cap, int status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int b = (int)(i / R), r = (int)(i % R);
  ShardRec1 o;
  // record 0 tells the other ranks whether this rank's batch outgrew its learnt candidate capacity (its lists are then empty and
  // every rank runs the batch again: fp_shard_search)
  // ... and (bit 1, `status`) whether it failed before this exchange: a failing rank keeps issuing every collective with empty
  // records instead of leaving its peers blocked in the next one, and every rank returns the error after its final sync
  o.pad = (i == 0) ? (((cand_total && cand_cap > 0 && *cand_total > cand_cap) ? 1 : 0) | status) : 0;
  if (r < cnt[b]) { o.pid = (long long)pid[i] + pid_offset; o.approx = approx[i]; }
  else { o.pid = -1; o.approx = -__builtin_inff(); }
  out[i] = o;
}
void fpk_shard_pack1(const float* sel_approx, const int32_t* sel_pid, const int32_t* sel_cnt, int B, int64_t R, int64_t pid_offset, void* rec1,
                     hipStream_t st, const int64_t* cand_total, int64_t cand_cap, int status) {
  const int64_t total = (int64_t)B * R;
  hipLaunchKernelGGL(k_shard_pack1, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, sel_approx, sel_pid, sel_cnt, R, pid_offset,
                     static_cast<ShardRec1*>(rec1), total, cand_total, cand_cap, status);
}

// flag |= OR over the ranks of the 32-bit status word at byte `word_off` of each rank's block of a gathered buffer (block stride
// `stride` bytes): bit 0 = that rank's candidate capacity overflowed, bit 1 = it failed
__global__ void k_shard_status(const unsigned char* __restrict__ all, int G, int64_t stride, int64_t word_off, int32_t* __restrict__ flag) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int f = *flag;
    for (int g = 0; g < G; ++g) f |= *reinterpret_cast<const int32_t*>(all + (int64_t)g * stride + word_off);
    *flag = f;
  }
}
void fpk_shard_status(const void* all, int G, int64_t stride_bytes, int64_t word_off, int32_t* flag, hipStream_t st) {
  hipLaunchKernelGGL(k_shard_status, dim3(1), dim3(64), 0, st, static_cast<const unsigned char*>(all), G, stride_bytes, word_off, flag);
}
void fpk_shard_any_overflow(const void* all_rec1, int G, int B, int64_t R, int32_t* flag, hipStream_t st) {
  (void)hipMemsetAsync(flag, 0, 4, st);
  fpk_shard_status(all_rec1, G, (int64_t)B * R * 16, 12, flag, st);
}

// pid / MFMA score / uncertainty (total and downward part) of the local survivors
__global__ void k_shard_pack2(const float* __restrict__ score, const float* __restrict__ unc, const float* __restrict__ uncm,
                              const int32_t* __restrict__ pid, const int32_t* __restrict__ cnt, int64_t R, int64_t pid_offset,
                              ShardRec2* __restrict__ out, int64_t total, int status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int b = (int)(i / R), r = (int)(i % R);
  ShardRec2 o;
  o.pad = i == 0 ? status : 0;
  if (r < cnt[b]) { o.pid = (long long)pid[i] + pid_offset; o.score = score[i]; o.uncm = uncm ? uncm[i] : 0.f; o.unc = unc ? unc[i] : 0.f; }
  else { o.pid = -1; o.score = -__builtin_inff(); o.uncm = 0.f; o.unc = 0.f; }
  out[i] = o;
}
void fpk_shard_pack2(const float* score, const float* unc, const float* uncm, const int32_t* sel_pid, const int32_t* sel_cnt, int B, int64_t R,
                     int64_t pid_offset, void* rec2, hipStream_t st, int status) {
  const int64_t total = (int64_t)B * R;
  hipLaunchKernelGGL(k_shard_pack2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, score, unc, uncm, sel_pid, sel_cnt, R, pid_offset,
                     static_cast<ShardRec2*>(rec2), total, status);
}

// ---- the same two steps for unions that do not fit the LDS sorts (n_ranks * R > 16384): no sort at all.  The ranks' lists are
// in ascending id order and the shards are contiguous id ranges in rank order, so (rank, slot) order IS id order: the cut is a
// radix select of the R-th largest approximate score (ties at the threshold taken in id order, like the sort's secondary key)
// followed by an ordered compaction, the union an ordered compaction.
__device__ __forceinline__ int shard_block_exscan(int v, int* lds /*[1025]*/) {   // exclusive prefix over the 1024 threads; lds[1024] = total
  const int t = threadIdx.x;
  lds[t] = v;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int add = t >= off ? lds[t - off] : 0;
    __syncthreads();
    lds[t] += add;
    __syncthreads();
  }
  const int incl = lds[t];
  if (t == 1023) lds[1024] = incl;
  __syncthreads();
  return incl - v;
}
__global__ __launch_bounds__(1024) void k_shard_cut_big(const ShardRec1* __restrict__ all, int G, int B, int64_t R, int64_t pid_lo, int64_t pid_hi,
                                                        int32_t* __restrict__ sel_pid, int32_t* __restrict__ sel_cnt) {
  __shared__ int hist[256];
  __shared__ int scan[1025];
  __shared__ uint32_t s_prefix;
  __shared__ int s_want;
  const int b = blockIdx.x, t = threadIdx.x;
  const int64_t n = (int64_t)G * R;
  auto rec = [&](int64_t i) -> ShardRec1 { return all[((i / R) * B + b) * R + (i % R)]; };
  auto key_of = [&](const ShardRec1& e) -> uint32_t { return e.pid >= 0 ? mono32(e.approx) : 0u; };   // 0 = padding (every valid key is > 0)
  // T = the R-th largest key (0 when fewer than R valid entries): 4 rounds of 8 bits from the top
  if (t == 0) { s_prefix = 0u; s_want = (int)R; }
  __syncthreads();
  for (int round = 0; round < 4; ++round) {
    const int shift = 24 - 8 * round;
    if (t < 256) hist[t] = 0;
    __syncthreads();
    const uint32_t prefix = s_prefix;
    const uint32_t himask = round == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int64_t i = t; i < n; i += 1024) {
      const uint32_t k = key_of(rec(i));
      if (k != 0u && (k & himask) == prefix) atomicAdd(&hist[(k >> shift) & 0xFFu], 1);
    }
    __syncthreads();
    if (t == 0) {
      int want = s_want, d = 255;
      for (; d > 0; --d) {
        if (hist[d] >= want) break;
        want -= hist[d];
      }
      // d == 0 also covers "fewer than `want` entries left": everything under the prefix is taken
      s_prefix = prefix | ((uint32_t)d << shift);
      s_want = want;
    }
    __syncthreads();
  }
  const uint32_t T = s_prefix;
  // ties at T are taken in (rank, slot) order until R entries are selected
  int64_t per = (n + 1023) / 1024;
  const int64_t i0 = (int64_t)t * per, i1 = i0 + per < n ? i0 + per : n;
  int gt = 0, eq = 0;
  for (int64_t i = i0; i < i1; ++i) {
    const uint32_t k = key_of(rec(i));
    gt += (k != 0u && k > T) ? 1 : 0;
    eq += (k != 0u && k == T) ? 1 : 0;
  }
  const int eq_before = shard_block_exscan(eq, scan);
  const int gt_before = shard_block_exscan(gt, scan);
  const int gt_total = scan[1024];
  __syncthreads();
  const int need = (int)R - gt_total;   // ties to take (>= 1 when R or more valid entries exist; all of them otherwise)
  (void)gt_before;
  int mine = 0, seen_eq = eq_before;
  for (int64_t i = i0; i < i1; ++i) {
    const ShardRec1 e = rec(i);
    const uint32_t k = key_of(e);
    bool take = k != 0u && k > T;
    if (k != 0u && k == T) { take = seen_eq < need; ++seen_eq; }
    if (take && e.pid >= pid_lo && e.pid < pid_hi) ++mine;
  }
  const int out0 = shard_block_exscan(mine, scan);
  const int total = scan[1024];
  int o = out0;
  seen_eq = eq_before;
  for (int64_t i = i0; i < i1; ++i) {
    const ShardRec1 e = rec(i);
    const uint32_t k = key_of(e);
    bool take = k != 0u && k > T;
    if (k != 0u && k == T) { take = seen_eq < need; ++seen_eq; }
    if (take && e.pid >= pid_lo && e.pid < pid_hi) sel_pid[(int64_t)b * R + o++] = (int32_t)(e.pid - pid_lo);
  }
  if (t == 0) sel_cnt[b] = total;
}
__global__ __launch_bounds__(1024) void k_shard_union_big(const ShardRec2* __restrict__ all, int G, int B, int64_t R, int64_t* __restrict__ u_pid,
                                                          float* __restrict__ u_score, int32_t* __restrict__ u_src, float* __restrict__ u_unc,
                                                          float* __restrict__ u_uncm, int32_t* __restrict__ u_cnt) {
  __shared__ int scan[1025];
  const int b = blockIdx.x, t = threadIdx.x;
  const int64_t n = (int64_t)G * R;
  const int64_t per = (n + 1023) / 1024;
  const int64_t i0 = (int64_t)t * per, i1 = i0 + per < n ? i0 + per : n;
  int mine = 0;
  for (int64_t i = i0; i < i1; ++i) mine += all[((i / R) * B + b) * R + (i % R)].pid >= 0 ? 1 : 0;
  int o = shard_block_exscan(mine, scan);
  const int total = scan[1024];
  for (int64_t i = i0; i < i1; ++i) {
    const ShardRec2 e = all[((i / R) * B + b) * R + (i % R)];
    if (e.pid < 0) continue;
    if (o < R) {
      const int64_t d = (int64_t)b * R + o;
      u_pid[d] = e.pid;
      u_score[d] = e.score;
      u_src[d] = (int32_t)i;
      u_unc[d] = e.unc;
      u_uncm[d] = e.uncm;
    }
    ++o;
  }
  if (t == 0) u_cnt[b] = total < (int)R ? total : (int)R;
}

// global cut: sort the union of the G ranks' local top-R by (approx desc, id asc), keep the first R valid, and
// of those the ones that live on this rank, written as local ids in ascending order.
__global__ __launch_bounds__(1024) void k_shard_cut(const ShardRec1* __restrict__ all, int G, int B, int npow2, int64_t R, int64_t pid_lo,
                                                    int64_t pid_hi, int32_t* __restrict__ sel_pid, int32_t* __restrict__ sel_cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* s_cntp = reinterpret_cast<int*>(smem);
  unsigned long long* v = reinterpret_cast<unsigned long long*>(smem + 16);
  const int b = blockIdx.x;
  const int n = (int)(G * R);
  for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
    unsigned long long key = 0ull;
    if (i < n) {
      const ShardRec1 e = all[((int64_t)(i / R) * B + b) * R + (i % R)];
      if (e.pid >= 0) key = ((unsigned long long)mono32(e.approx) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)e.pid);
    }
    v[i] = key;
  }
  __syncthreads();
  for (int k = 2; k <= npow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
        int ixj = i ^ j;
        if (ixj > i) {
          unsigned long long a = v[i], c = v[ixj];
          bool desc = ((i & k) == 0);
          if ((a < c) == desc) { v[i] = c; v[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
  // first R entries (valid ones have key != 0): keep local ids; then sort them ascending.
  const int lim = (int)(R < npow2 ? R : npow2);
  unsigned int* ids = reinterpret_cast<unsigned int*>(smem + 16 + (size_t)npow2 * 8);  // [rpow2]
  int rpow2 = 2;
  while (rpow2 < lim) rpow2 <<= 1;
  if (threadIdx.x == 0) *s_cntp = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < rpow2; i += blockDim.x) {
    unsigned int x = 0xFFFFFFFFu;
    if (i < lim) {
      unsigned long long key = v[i];
      if (key) {
        int64_t id = (int64_t)(0xFFFFFFFFu - (uint32_t)key);
        if (id >= pid_lo && id < pid_hi) { x = (unsigned int)(id - pid_lo); atomicAdd(s_cntp, 1); }
      }
    }
    ids[i] = x;
  }
  __syncthreads();
  for (int k = 2; k <= rpow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < rpow2; i += blockDim.x) {
        int ixj = i ^ j;
        if (ixj > i) {
          unsigned int a = ids[i], c = ids[ixj];
          bool up = ((i & k) == 0);
          if ((a > c) == up) { ids[i] = c; ids[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
  const int cnt = *s_cntp;
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) sel_pid[(int64_t)b * R + i] = (int32_t)ids[i];
  if (threadIdx.x == 0) sel_cnt[b] = cnt;
}

int fpk_shard_global_cut(const void* all_rec1, int G, int B, int64_t R, int64_t pid_lo, int64_t pid_hi, int32_t* sel_pid, int32_t* sel_cnt,
                         hipStream_t st) {
  const int n = (int)(G * R);
  const int np2 = next_pow2(n < 2 ? 2 : n);
  const int rp2 = next_pow2((int)(R < 2 ? 2 : R));
  const size_t lds = (size_t)np2 * 8 + (size_t)rp2 * 4 + 16;
  static const bool force_big = fp_test_opt("shard_big", 0) != 0;   // testing: the sort-free paths for every size
  if (lds > 160 * 1024 || force_big) {
    if ((int64_t)G * R >= 0x7FFFFFFFll) return -1;
    hipLaunchKernelGGL(k_shard_cut_big, dim3((unsigned)B), dim3(1024), 0, st, static_cast<const ShardRec1*>(all_rec1), G, B, R, pid_lo, pid_hi, sel_pid,
                       sel_cnt);
    return 0;
  }
  static std::atomic<uint64_t> lds_ok{0};
  fp_allow_big_lds((const void*)k_shard_cut, lds_ok, 160 * 1024);
  hipLaunchKernelGGL(k_shard_cut, dim3((unsigned)B), dim3(1024), lds, st, static_cast<const ShardRec1*>(all_rec1), G, B, np2, R, pid_lo, pid_hi,
                     sel_pid, sel_cnt);
  return 0;
}

// union of the ranks' exact-scored documents of one query, in ascending doc id order (== the unsharded rerank list):
// u_pid / u_score / u_exact / u_unc [B][R], u_cnt [B]
__global__ __launch_bounds__(1024) void k_shard_union(const ShardRec2* __restrict__ all, int G, int B, int npow2, int64_t R,
                                                      int64_t* __restrict__ u_pid, float* __restrict__ u_score, int32_t* __restrict__ u_src,
                                                      float* __restrict__ u_unc, float* __restrict__ u_uncm, int32_t* __restrict__ u_cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* v = reinterpret_cast<unsigned long long*>(smem);   // id << 32 | index into the [G*R] union
  __shared__ int s_n;
  const int b = blockIdx.x;
  const int n = (int)(G * R);
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  int mine = 0;
  for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
    unsigned long long key = ~0ull;
    if (i < n) {
      const long long pid = all[((int64_t)(i / R) * B + b) * R + (i % R)].pid;
      if (pid >= 0) { key = ((unsigned long long)(uint32_t)pid << 32) | (unsigned long long)(uint32_t)i; ++mine; }
    }
    v[i] = key;
  }
  if (mine) atomicAdd(&s_n, mine);
  __syncthreads();
  for (int k = 2; k <= npow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
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
  const int nv = s_n < (int)R ? s_n : (int)R;   // the global cut keeps at most R documents over all ranks
  for (int i = threadIdx.x; i < nv; i += blockDim.x) {
    const int idx = (int)(uint32_t)v[i];
    const ShardRec2 e = all[((int64_t)(idx / R) * B + b) * R + (idx % R)];
    const int64_t o = (int64_t)b * R + i;
    u_pid[o] = e.pid;
    u_score[o] = e.score;
    u_src[o] = idx;          // rank * R + slot in that rank's rerank list
    u_unc[o] = e.unc;
    u_uncm[o] = e.uncm;
  }
  if (threadIdx.x == 0) u_cnt[b] = nv;
}
int fpk_shard_union(const void* all_rec2, int G, int B, int64_t R, int64_t* u_pid, float* u_score, int32_t* u_src, float* u_unc, float* u_uncm,
                    int32_t* u_cnt, hipStream_t st) {
  const int n = (int)(G * R);
  const int np2 = next_pow2(n < 2 ? 2 : n);
  static const bool force_big = fp_test_opt("shard_big", 0) != 0;
  if ((size_t)np2 * 8 > 160 * 1024 || force_big) {
    if ((int64_t)G * R >= 0x7FFFFFFFll) return -1;
    hipLaunchKernelGGL(k_shard_union_big, dim3((unsigned)B), dim3(1024), 0, st, static_cast<const ShardRec2*>(all_rec2), G, B, R, u_pid, u_score, u_src,
                       u_unc, u_uncm, u_cnt);
    return 0;
  }
  static std::atomic<uint64_t> lds_ok{0};
  fp_allow_big_lds((const void*)k_shard_union, lds_ok, 160 * 1024);
  hipLaunchKernelGGL(k_shard_union, dim3((unsigned)B), dim3(1024), (size_t)np2 * 8, st, static_cast<const ShardRec2*>(all_rec2), G, B, np2, R, u_pid,
                     u_score, u_src, u_unc, u_uncm, u_cnt);
  return 0;
}

// ---- third exchange of the sharded search: only the near-tied documents are repaired, by the rank that holds them ----
// marks = union positions (k_final_mark on the union, identical on every rank; marks == nullptr: every flagged document).
// lmarks / lnmark: the marked documents of THIS rank as slots of its rerank list (the input of the repair kernel).
__global__ __launch_bounds__(256) void k_shard_local_marks(const int32_t* __restrict__ marks, const int32_t* __restrict__ nmark,
                                                           const float* __restrict__ u_unc, const int32_t* __restrict__ u_cnt,
                                                           const int32_t* __restrict__ u_src, int64_t R, int rank,
                                                           int32_t* __restrict__ lmarks, int32_t* __restrict__ lnmark) {
  __shared__ int s_n;
  const int b = blockIdx.x;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const int n = marks ? nmark[b] : u_cnt[b];
  for (int m = threadIdx.x; m < n; m += 256) {
    const int p = marks ? marks[(int64_t)b * R + m] : m;
    if (!marks && !(u_unc[(int64_t)b * R + p] > 0.f)) continue;
    const int src = u_src[(int64_t)b * R + p];
    if (src / (int)R == rank) lmarks[(int64_t)b * R + atomicAdd(&s_n, 1)] = src % (int)R;
  }
  __syncthreads();
  if (threadIdx.x == 0) lnmark[b] = s_n;
}
// x[b][p] = the repaired score of union position p, for the marked documents this rank holds (other entries are not read)
__global__ __launch_bounds__(256) void k_shard_pack3(const int32_t* __restrict__ marks, const int32_t* __restrict__ nmark,
                                                     const float* __restrict__ u_unc, const int32_t* __restrict__ u_cnt,
                                                     const int32_t* __restrict__ u_src, int64_t R, int rank,
                                                     const float* __restrict__ exact_local, float* __restrict__ x) {
  const int b = blockIdx.x;
  const int n = marks ? nmark[b] : u_cnt[b];
  for (int m = threadIdx.x; m < n; m += 256) {
    const int p = marks ? marks[(int64_t)b * R + m] : m;
    if (!marks && !(u_unc[(int64_t)b * R + p] > 0.f)) continue;
    const int src = u_src[(int64_t)b * R + p];
    if (src / (int)R == rank) x[(int64_t)b * R + p] = exact_local[(int64_t)b * R + src % (int)R];
  }
}
// marked documents take the repaired score from the rank that holds them
__global__ __launch_bounds__(256) void k_shard_apply3(const int32_t* __restrict__ marks, const int32_t* __restrict__ nmark,
                                                      const float* __restrict__ u_unc, const int32_t* __restrict__ u_cnt,
                                                      const int32_t* __restrict__ u_src, int64_t R, int64_t xstride /*floats per rank*/,
                                                      const float* __restrict__ xall /*[G][xstride]*/, float* __restrict__ u_score) {
  const int b = blockIdx.x;
  const int n = marks ? nmark[b] : u_cnt[b];
  for (int m = threadIdx.x; m < n; m += 256) {
    const int p = marks ? marks[(int64_t)b * R + m] : m;
    if (!marks && !(u_unc[(int64_t)b * R + p] > 0.f)) continue;
    const int g = u_src[(int64_t)b * R + p] / (int)R;
    u_score[(int64_t)b * R + p] = xall[(int64_t)g * xstride + (int64_t)b * R + p];
  }
}
void fpk_shard_local_marks(const int32_t* marks, const int32_t* nmark, const float* u_unc, const int32_t* u_cnt, const int32_t* u_src, int B,
                           int64_t R, int rank, int32_t* lmarks, int32_t* lnmark, hipStream_t st) {
  hipLaunchKernelGGL(k_shard_local_marks, dim3((unsigned)B), dim3(256), 0, st, marks, nmark, u_unc, u_cnt, u_src, R, rank, lmarks, lnmark);
}
void fpk_shard_pack3(const int32_t* marks, const int32_t* nmark, const float* u_unc, const int32_t* u_cnt, const int32_t* u_src, int B, int64_t R,
                     int rank, const float* exact_local, float* x, hipStream_t st) {
  hipLaunchKernelGGL(k_shard_pack3, dim3((unsigned)B), dim3(256), 0, st, marks, nmark, u_unc, u_cnt, u_src, R, rank, exact_local, x);
}
void fpk_shard_apply3(const int32_t* marks, const int32_t* nmark, const float* u_unc, const int32_t* u_cnt, const int32_t* u_src, int B, int64_t R,
                      const float* xall, float* u_score, hipStream_t st, int64_t xstride) {
  hipLaunchKernelGGL(k_shard_apply3, dim3((unsigned)B), dim3(256), 0, st, marks, nmark, u_unc, u_cnt, u_src, R, xstride > 0 ? xstride : (int64_t)B * R,
                     xall, u_score);
}

// ============================================================================================
// misc
// ============================================================================================
// bad (may be null): counts the entries outside [0, limit) -- a code that is no centroid, a list entry that is no document
__global__ void k_narrow(const int64_t* __restrict__ in, int32_t* __restrict__ out, int64_t n, int64_t add, int64_t limit, uint32_t* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = in[i];
    if (bad && (v < 0 || v >= limit)) atomicAdd(bad, 1u);
    out[i] = (int32_t)(v + add);
  }
}
void fpk_narrow_i64_i32(const int64_t* in, int32_t* out, int64_t n, int64_t add, hipStream_t st, int64_t limit, uint32_t* bad) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_narrow, dim3(fp_grid_cap((n + 255) / 256, 256)), dim3(256), 0, st, in, out, n, add, limit, bad);
}

// the IVF lists as construct_index receives them: *flag != 0 when some list is not strictly ascending (the reference sorts and
// de-duplicates the gathered ids per query, search.rs:538-541, so it takes any order; S3's range cut needs ascending lists).
// One wave per list.
__global__ __launch_bounds__(256) void k_ivf_check_sorted(const int64_t* __restrict__ off, const int32_t* __restrict__ pids, int64_t P,
                                                          uint32_t* __restrict__ flag) {
  const int lane = threadIdx.x & 63;
  for (int64_t list = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); list < P; list += (int64_t)gridDim.x * 4) {
    const int64_t beg = off[list], end = off[list + 1];
    bool bad = false;
    for (int64_t i = beg + lane; i + 1 < end; i += 64) bad |= pids[i] >= pids[i + 1];
    if (bad) atomicOr(flag, 1u);
  }
}
void fpk_ivf_check_sorted(const int64_t* off, const int32_t* pids, int64_t P, uint32_t* flag, hipStream_t st) {
  if (P <= 0) return;
  hipLaunchKernelGGL(k_ivf_check_sorted, dim3(fp_grid_cap((P + 3) / 4, 256)), dim3(256), 0, st, off, pids, P, flag);
}

// reconstruct_embeddings (embeddings.rs:12-69): decompress rows to fp32.  One 64-thread
// block per token; the fp32 sum of squares is taken in ascending-dim order by one lane so
// the result is bit-identical to the CPU reference order.
__global__ __launch_bounds__(64) void k_reconstruct(const uint16_t* __restrict__ cent, const uint16_t* __restrict__ lut,
                                                    const int32_t* __restrict__ codes, const uint8_t* __restrict__ resid, int D,
                                                    int nbits, const int64_t* __restrict__ tok_idx, float* __restrict__ out, int native) {
  __shared__ float e[512];
  __shared__ float nrm;
  const int64_t t = tok_idx[blockIdx.x];
  const int pb = 8 / nbits, pr = D * nbits / 8;
  const int32_t code = codes[t];
  for (int d = threadIdx.x; d < D; d += 64) {
    const int byte = resid[t * pr + fp_resid_pos(d / pb, nbits, D / 8, native)];
    const half_t w = __builtin_bit_cast(half_t, lut[byte * pb + d % pb]);
    const half_t c = __builtin_bit_cast(half_t, cent[(int64_t)code * D + d]);
    e[d] = (float)(half_t)((float)w + (float)c);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ss = 0.f;
    for (int d = 0; d < D; ++d) ss += e[d] * e[d];
    nrm = (float)(half_t)__builtin_sqrtf(ss);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 64) out[(int64_t)blockIdx.x * D + d] = (float)(half_t)(e[d] / nrm);
}

void fpk_reconstruct(const FpIndexDev& ix, const int64_t* tok_idx, int64_t n, float* out, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_reconstruct, dim3((unsigned)n), dim3(64), 0, st, ix.centroids, ix.lut, ix.codes, ix.residuals, ix.dim, ix.nbits,
                     tok_idx, out, ix.resid_native);
}

// token-score matrices (search.rs:651-653, :668-686): one 64-thread block per hit.  Per token: decompress exactly like
// k_reconstruct (ascending fp32 norm chain), then lane j takes query token j (loop for q_len > 64): ascending-k fp32
// chain, one rounding to fp16 -- the CPU reference's order, so the matrices are bit-identical to the oracle.
__global__ __launch_bounds__(64) void k_token_scores(const uint16_t* __restrict__ cent, const uint16_t* __restrict__ lut,
                                                     const int32_t* __restrict__ codes, const uint8_t* __restrict__ resid,
                                                     const int64_t* __restrict__ doc_off, const uint16_t* __restrict__ perm, int D, int nbits,
                                                     const uint16_t* __restrict__ queries /*[nq][Q][D]*/, int Q,
                                                     const int32_t* __restrict__ hit_query, const int32_t* __restrict__ hit_pid,
                                                     const int64_t* __restrict__ out_off, uint16_t* __restrict__ out, int native) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* qs = reinterpret_cast<float*>(smem);          // [Q][D] the hit's query as fp32
  float* e = qs + (size_t)Q * D;                       // [D]
  __shared__ float nrm;
  const int h = blockIdx.x;
  const int32_t pid = hit_pid[h];
  const uint16_t* q = queries + (int64_t)hit_query[h] * Q * D;
  for (int i = threadIdx.x; i < Q * D; i += 64) qs[i] = (float)__builtin_bit_cast(half_t, q[i]);
  const int64_t t0 = doc_off[pid];
  const int len = (int)(doc_off[pid + 1] - t0);
  const int pb = 8 / nbits, pr = D * nbits / 8;
  uint16_t* o = out + out_off[h];
  for (int s = 0; s < len; ++s) {
    const int64_t t = t0 + s;
    const int32_t code = codes[t];
    __syncthreads();  // e / nrm of the previous token are no longer read
    for (int d = threadIdx.x; d < D; d += 64) {
      const int byte = resid[t * pr + fp_resid_pos(d / pb, nbits, D / 8, native)];
      const half_t w = __builtin_bit_cast(half_t, lut[byte * pb + d % pb]);
      const half_t c = __builtin_bit_cast(half_t, cent[(int64_t)code * D + d]);
      e[d] = (float)(half_t)((float)w + (float)c);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float ss = 0.f;
      for (int d = 0; d < D; ++d) ss += e[d] * e[d];
      nrm = (float)(half_t)__builtin_sqrtf(ss);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 64) e[d] = (float)(half_t)(e[d] / nrm);
    __syncthreads();
    const int torig = perm ? (int)perm[t] : s;
    for (int j = threadIdx.x; j < Q; j += 64) {
      const float* qq = qs + (size_t)j * D;
      float acc = 0.f;
      for (int d = 0; d < D; ++d) acc += e[d] * qq[d];   // products of two fp16 values are exact in fp32: mul+add == fma
      o[(int64_t)j * len + torig] = __builtin_bit_cast(uint16_t, (half_t)acc);
    }
  }
}

int fpk_token_scores(const FpIndexDev& ix, const uint16_t* queries, int Q, const int32_t* hit_query, const int32_t* hit_pid, int64_t n_hits,
                     const int64_t* out_off, uint16_t* out, hipStream_t st) {
  if (n_hits <= 0) return 0;
  const size_t lds = ((size_t)Q * ix.dim + ix.dim) * sizeof(float);
  if (lds > 150 * 1024) return -1;
  static std::atomic<uint64_t> lds_ok{0};
  fp_allow_big_lds((const void*)k_token_scores, lds_ok, 152 * 1024);
  for (int64_t h0 = 0; h0 < n_hits; h0 += 0x7FFFFF00ll / 64) {   // grid.x * 64 threads must stay below 2^32
    const int64_t nh = std::min<int64_t>(n_hits - h0, 0x7FFFFF00ll / 64);
    hipLaunchKernelGGL(k_token_scores, dim3((unsigned)nh), dim3(64), lds, st, ix.centroids, ix.lut, ix.codes, ix.residuals, ix.doc_off, ix.perm,
                       ix.dim, ix.nbits, queries, Q, hit_query + h0, hit_pid + h0, out_off + h0, out, ix.resid_native);
  }
  return 0;
}

// ============================================================================================
// index creation, device part (create.rs:148-184, :404-428)
// ============================================================================================
// Nearest centroid, EXACT: the reference takes argmax over h(fp32 dot) of an ATen half matmul, whose fp32 sum runs in
// ascending k; an MFMA contraction sums in another order, which moves ~0.05 % of the fp16-rounded scores by an ulp and
// would flip the argmax between near-tied centroids.  Codes are integers and must be identical, so this kernel keeps
// one ascending-k fp32 chain per (token, centroid): 64 tokens x 64 centroids per tile, 4x4 outputs per thread, operands
// in LDS as fp32.  Ties go to the lowest centroid index (torch.argmax: first maximal value).
// L2 = true (k-means assignment): score = dot - half_sqnorm[c] compared in fp32 instead of the fp16-rounded dot.
template <bool L2>
__global__ __launch_bounds__(256) void k_assign_exact(const uint16_t* __restrict__ emb, int64_t T, const uint16_t* __restrict__ cent,
                                                      int64_t C, const float* __restrict__ half_sqnorm, int32_t* __restrict__ codes, int D) {
  // any dim (runtime): As / Bs [64][D + 1] fp32 and best_s [64][16] in dynamic LDS
  extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
  const int ld = D + 1;
  float* As_ = reinterpret_cast<float*>(xsm);
  float* Bs_ = As_ + 64 * ld;
  unsigned long long (*best_s)[16] = reinterpret_cast<unsigned long long (*)[16]>(xsm + (size_t)2 * 64 * ld * 4);
#define As(r, d) As_[(r) * ld + (d)]
#define Bs(r, d) Bs_[(r) * ld + (d)]
  const int tid = threadIdx.x;
  const int tr = tid >> 4, tc = tid & 15;       // 16 x 16 threads, 4 rows x 4 cols each
  const int64_t t0 = (int64_t)blockIdx.x * 64;
  for (int i = tid; i < 64 * D; i += 256) {
    const int r = i / D, d = i % D;
    As(r, d) = (t0 + r < T) ? (float)__builtin_bit_cast(half_t, emb[(t0 + r) * D + d]) : 0.f;
  }
  unsigned long long best[4] = {0ull, 0ull, 0ull, 0ull};
  for (int64_t c0 = 0; c0 < C; c0 += 64) {
    __syncthreads();
    for (int i = tid; i < 64 * D; i += 256) {
      const int r = i / D, d = i % D;
      Bs(r, d) = (c0 + r < C) ? (float)__builtin_bit_cast(half_t, cent[(c0 + r) * D + d]) : 0.f;
    }
    __syncthreads();
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int kk = 0; kk < D; ++kk) {   // ascending k: each accumulator is the reference's chain (products of fp16 pairs are exact in fp32)
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As(tr * 4 + i, kk);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs(tc + 16 * j, kk);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t c = c0 + tc + 16 * j;
        if (c < C) {
          uint32_t ord;
          if constexpr (L2) {
            const uint32_t fb = __float_as_uint(acc[i][j] - half_sqnorm[c]);
            ord = fb ^ ((fb >> 31) ? 0xFFFFFFFFu : 0x80000000u);   // order-preserving key of an fp32 value
          } else {
            ord = mono16(__builtin_bit_cast(uint16_t, (half_t)acc[i][j]));
          }
          const unsigned long long key = ((unsigned long long)ord << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)c);
          best[i] = key > best[i] ? key : best[i];   // higher score, then lower index
        }
      }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) best_s[tr * 4 + i][tc] = best[i];
  __syncthreads();
  if (tid < 64 && t0 + tid < T) {
    unsigned long long m = 0ull;
    for (int j = 0; j < 16; ++j) m = best_s[tid][j] > m ? best_s[tid][j] : m;
    codes[t0 + tid] = (int32_t)(0xFFFFFFFFu - (uint32_t)(m & 0xFFFFFFFFull));
  }
#undef As
#undef Bs
}
static size_t assign_exact_lds(int D) { return (size_t)2 * 64 * (D + 1) * 4 + 64 * 16 * 8; }
#define ASSIGN_MAX_DIM 256   // 2 x 64 x (D+1) fp32 operand tiles + 8 KiB must fit the 160 KiB of LDS

// ---- MFMA fast path of the nearest-centroid search, still exact -------------------------------------------------------
// The exact kernel above runs at 30 TFLOP/s.  MFMA sums in another order, so its scores cannot decide near-ties; but they
// can NARROW the search: pass 1 finds, per token, the maximum MFMA score m; pass 2 collects every centroid whose MFMA score
// is within one fp16 ulp (plus twice the worst-case summation error) of m -- the true argmax of the fp16-rounded exact
// scores is always among them (see k_assign_thr) -- and k_assign_recheck evaluates only those (1-2 per token, at most
// ASSIGN_CAP) with the ascending-k chain and the first-index tie rule.  A token that collects more than ASSIGN_CAP
// candidates (many near-duplicate centroids) sends the whole chunk back to the exact kernel.
#define ASSIGN_CAP 8
template <int D, int PASS>
__global__ __launch_bounds__(256) void k_assign_mfma(const uint16_t* __restrict__ emb, int64_t T, const uint16_t* __restrict__ cent, int64_t C,
                                                     float* __restrict__ tmax /*PASS 1 out*/, const float* __restrict__ thr /*PASS 2 in*/,
                                                     int32_t* __restrict__ ccnt, int32_t* __restrict__ ccand) {
  constexpr int CH = D / 8;
  constexpr int ROWB = D * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ts = smem;               // [128 tokens][ROWB]
  unsigned char* Cs = smem + 128 * ROWB;  // [128 centroids][ROWB]
  const int tid = threadIdx.x;
  const int64_t t0 = (int64_t)blockIdx.x * 128;
  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  for (int i = tid; i < 128 * CH; i += 256) {
    const int row = i / CH, j = i % CH;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t0 + row < T) v = *reinterpret_cast<const uint4*>(emb + (t0 + row) * D + j * 8);
    *reinterpret_cast<uint4*>(Ts + row * ROWB + ((j ^ (row & (CH - 1))) * 16)) = v;
  }
  // per-lane state over this lane's centroid columns: running maxima (pass 1) or the thresholds of its 32 token rows (pass 2)
  float st[2][16];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (PASS == 1) {
        st[a][r] = -INFINITY;
      } else {
        const int64_t t = t0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        st[a][r] = (t < T) ? thr[t] : INFINITY;
      }
    }
  for (int64_t c0 = 0; c0 < C; c0 += 128) {
    __syncthreads();
    for (int i = tid; i < 128 * CH; i += 256) {
      const int row = i / CH, j = i % CH;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (c0 + row < C) v = *reinterpret_cast<const uint4*>(cent + (c0 + row) * D + j * 8);
      *reinterpret_cast<uint4*>(Cs + row * ROWB + ((j ^ (row & (CH - 1))) * 16)) = v;
    }
    __syncthreads();
    f16v acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      h8 af[2], bf[2];
      const int j = ks * 2 + hi;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int rq = wr * 64 + t * 32 + l31;
        const int rc = wc * 64 + t * 32 + l31;
        af[t] = *reinterpret_cast<const h8*>(Ts + rq * ROWB + ((j ^ (rq & (CH - 1))) * 16));
        bf[t] = *reinterpret_cast<const h8*>(Cs + rc * ROWB + ((j ^ (rc & (CH - 1))) * 16));
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int64_t c = c0 + wc * 64 + b * 32 + l31;
      if (c >= C) continue;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[a][b][r];
          if (PASS == 1) {
            st[a][r] = fmaxf(st[a][r], v);
          } else if (v >= st[a][r]) {
            const int64_t t = t0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int pos = atomicAdd(&ccnt[t], 1);
            if (pos < ASSIGN_CAP) ccand[t * ASSIGN_CAP + pos] = (int32_t)c;
          }
        }
    }
  }
  if (PASS == 1) {
    __shared__ float red[2][128];
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = st[a][r];
#pragma unroll
        for (int s = 1; s < 32; s <<= 1) v = fmaxf(v, __shfl_xor(v, s, 64));
        if (l31 == 0) red[wc][wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = v;
      }
    __syncthreads();
    if (tid < 128 && t0 + tid < T) tmax[t0 + tid] = fmaxf(red[0][tid], red[1][tid]);
  }
}

// thr[t] = m - (two fp16 ulps at |m| + twice the worst-case fp32 summation error of a D-term dot product of this token).
// Why the true winner w is collected: h is monotone, so exact_w > exact_c - ulp16 for every c (else h(exact_c) > h(exact_w));
// with c* = argmax of the MFMA scores and |mfma - exact| <= eps:  mfma_w >= exact_w - eps > exact_c* - ulp16 - eps >= m - ulp16 - 2 eps.
__global__ void k_assign_thr(const uint16_t* __restrict__ emb, int64_t T, int D, float cmaxabs, const float* __restrict__ tmax,
                             float* __restrict__ thr, int32_t* __restrict__ ccnt) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (int64_t)gridDim.x * blockDim.x) {
    float l1 = 0.f;
    for (int d = 0; d < D; ++d) l1 += __builtin_fabsf((float)__builtin_bit_cast(half_t, emb[t * D + d]));
    const float m = tmax[t];
    const float ulp2 = __builtin_fmaxf(__builtin_fabsf(m), 6.103515625e-05f) * 0.001953125f;   // 2^-9 |m|  >= 2 ulp16
    const float eps2 = l1 * cmaxabs * (float)D * 1.1920929e-07f * 2.0f;                          // 2 * D * 2^-23 * sum|x_k| max|c|
    thr[t] = m - (ulp2 + eps2);
    ccnt[t] = 0;
  }
}

// exact evaluation of the collected candidates; codes[t] = -1 when the candidate list overflowed
__global__ void k_assign_recheck(const uint16_t* __restrict__ emb, int64_t T, const uint16_t* __restrict__ cent, int D,
                                 const int32_t* __restrict__ ccnt, const int32_t* __restrict__ ccand, int32_t* __restrict__ codes,
                                 int32_t* __restrict__ overflow) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (int64_t)gridDim.x * blockDim.x) {
    const int n = ccnt[t];
    if (n > ASSIGN_CAP || n < 1) {   // n < 1 cannot happen (the maximum itself passes its threshold) unless a score is NaN
      codes[t] = -1;
      atomicAdd(overflow, 1);
      continue;
    }
    unsigned long long best = 0ull;
    for (int i = 0; i < n; ++i) {
      const int32_t c = ccand[t * ASSIGN_CAP + i];
      float acc = 0.f;
      for (int d = 0; d < D; ++d)
        acc += (float)__builtin_bit_cast(half_t, emb[t * D + d]) * (float)__builtin_bit_cast(half_t, cent[(int64_t)c * D + d]);
      const uint16_t hv = __builtin_bit_cast(uint16_t, (half_t)acc);
      const unsigned long long key = ((unsigned long long)mono16(hv) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)c);
      best = key > best ? key : best;
    }
    codes[t] = (int32_t)(0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFull));
  }
}

// residual quantisation + packing: one thread per output byte
__global__ __launch_bounds__(256) void k_quantize_pack(const uint16_t* __restrict__ emb, const uint16_t* __restrict__ cent,
                                                       const int32_t* __restrict__ codes, const uint16_t* __restrict__ cutoffs, int D, int nbits,
                                                       int64_t T, uint8_t* __restrict__ out, int64_t* __restrict__ codes64) {
  const int pr = D * nbits / 8, per = 8 / nbits, ncut = (1 << nbits) - 1;
  const int64_t total = T * pr;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = g / pr;
    const int bi = (int)(g % pr);
    const int32_t code = codes[t];
    if (bi == 0) codes64[t] = code;
    unsigned byte = 0;
    for (int v = 0; v < per; ++v) {
      const int d = bi * per + v;
      const float e = (float)__builtin_bit_cast(half_t, emb[t * D + d]);
      const float c = (float)__builtin_bit_cast(half_t, cent[(int64_t)code * D + d]);
      const half_t r = (half_t)(e - c);   // fp16 subtraction == fp32 subtraction rounded once
      int bucket = 0;
      for (int i = 0; i < ncut; ++i) bucket += ((float)__builtin_bit_cast(half_t, cutoffs[i]) < (float)r) ? 1 : 0;
      // bits of `bucket` LSB first, appended to the big-endian bit stream
      for (int kbit = 0; kbit < nbits; ++kbit) byte = (byte << 1) | ((bucket >> kbit) & 1u);
    }
    out[g] = (uint8_t)byte;
  }
}

// scratch: tmax / thr f32 [T], ccnt i32 [T], ccand i32 [T * ASSIGN_CAP], overflow i32 [1]  (all inside `work`, see fpk_compress_work_bytes)
size_t fpk_compress_work_bytes(int64_t T) { return (size_t)T * (4 + 4 + 4 + 4 * ASSIGN_CAP) + 64; }

static int assign_codes(const uint16_t* emb, int64_t T, const uint16_t* cent, int64_t C, int D, float cmaxabs, int32_t* codes32, void* work,
                        hipStream_t st) {
  static const int impl_env = fp_test_opt("assign_exact", 0) != 0 ? 1 : 0;   // the all-VALU exact kernel only (tools/bench_compress.py)
  const unsigned eblocks = fp_grid_cap((T + 63) / 64, 256);
  if ((int64_t)eblocks * 64 < T) return -2;   // callers chunk far below this
  auto exact = [&]() {
    static std::atomic<uint64_t> ok{0};
    fp_allow_big_lds((const void*)k_assign_exact<false>, ok, 160 * 1024);
    hipLaunchKernelGGL((k_assign_exact<false>), dim3(eblocks), dim3(256), assign_exact_lds(D), st, emb, T, cent, C, (const float*)nullptr, codes32, D);
  };
  if (D < 1 || D > ASSIGN_MAX_DIM) return -1;
  if (impl_env || !work || C < 256 || (D != 128 && D != 64)) {   // tiny tables: the exact kernel is as fast; the MFMA narrowing is built for dim 64 / 128
    exact();
    return 0;
  }
  float* tmax = static_cast<float*>(work);
  float* thr = tmax + T;
  int32_t* ccnt = reinterpret_cast<int32_t*>(thr + T);
  int32_t* ccand = ccnt + T;
  int32_t* overflow = ccand + T * ASSIGN_CAP;
  const unsigned mblocks = (unsigned)((T + 127) / 128);
  const size_t lds = (size_t)2 * 128 * D * 2;
  (void)hipMemsetAsync(overflow, 0, 4, st);
  if (D == 128) {
    static std::atomic<uint64_t> ok1{0}, ok2{0};
    fp_allow_big_lds((const void*)k_assign_mfma<128, 1>, ok1, 80 * 1024);
    fp_allow_big_lds((const void*)k_assign_mfma<128, 2>, ok2, 80 * 1024);
    hipLaunchKernelGGL((k_assign_mfma<128, 1>), dim3(mblocks), dim3(256), lds, st, emb, T, cent, C, tmax, (const float*)nullptr, (int32_t*)nullptr,
                       (int32_t*)nullptr);
    hipLaunchKernelGGL(k_assign_thr, dim3(fp_grid_cap((T + 255) / 256, 256)), dim3(256), 0, st, emb, T, D, cmaxabs, tmax, thr, ccnt);
    hipLaunchKernelGGL((k_assign_mfma<128, 2>), dim3(mblocks), dim3(256), lds, st, emb, T, cent, C, (float*)nullptr, thr, ccnt, ccand);
  } else {
    hipLaunchKernelGGL((k_assign_mfma<64, 1>), dim3(mblocks), dim3(256), lds, st, emb, T, cent, C, tmax, (const float*)nullptr, (int32_t*)nullptr,
                       (int32_t*)nullptr);
    hipLaunchKernelGGL(k_assign_thr, dim3(fp_grid_cap((T + 255) / 256, 256)), dim3(256), 0, st, emb, T, D, cmaxabs, tmax, thr, ccnt);
    hipLaunchKernelGGL((k_assign_mfma<64, 2>), dim3(mblocks), dim3(256), lds, st, emb, T, cent, C, (float*)nullptr, thr, ccnt, ccand);
  }
  hipLaunchKernelGGL(k_assign_recheck, dim3(fp_grid_cap((T + 255) / 256, 256)), dim3(256), 0, st, emb, T, cent, D, ccnt, ccand, codes32, overflow);
  int32_t h_over = 0;
  if (hipMemcpyAsync(&h_over, overflow, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -3;
  if (h_over > 0) exact();   // a token met more than ASSIGN_CAP near-tied centroids: the whole chunk goes through the exact kernel
  return 0;
}

int fpk_compress(const uint16_t* emb, int64_t T, const uint16_t* cent, int64_t C, int D, int nbits, const uint16_t* cutoffs, float cmaxabs,
                 int32_t* codes32, int64_t* codes64, uint8_t* out, void* work, hipStream_t st) {
  if (T <= 0) return 0;
  if (int rc = assign_codes(emb, T, cent, C, D, cmaxabs, codes32, work, st)) return rc;
  const int64_t total = T * (D * nbits / 8);
  hipLaunchKernelGGL(k_quantize_pack, dim3(fp_grid_cap((total + 255) / 256, 256)), dim3(256), 0, st, emb, cent, codes32, cutoffs, D, nbits, T, out,
                     codes64);
  return 0;
}

__global__ void k_widen_i32(const int32_t* __restrict__ in, int64_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = in[i];
}

int fpk_assign_l2(const uint16_t* emb, int64_t T, const uint16_t* cent, const float* half_sqnorm, int64_t C, int D, int32_t* codes32,
                  int64_t* codes64, hipStream_t st) {
  if (T <= 0) return 0;
  const unsigned blocks = fp_grid_cap((T + 63) / 64, 256);
  if ((int64_t)blocks * 64 < T) return -2;
  if (D < 1 || D > ASSIGN_MAX_DIM) return -1;
  static std::atomic<uint64_t> ok{0};
  fp_allow_big_lds((const void*)k_assign_exact<true>, ok, 160 * 1024);
  hipLaunchKernelGGL((k_assign_exact<true>), dim3(blocks), dim3(256), assign_exact_lds(D), st, emb, T, cent, C, half_sqnorm, codes32, D);
  hipLaunchKernelGGL(k_widen_i32, dim3(fp_grid_cap((T + 255) / 256, 256)), dim3(256), 0, st, codes32, codes64, T);
  return 0;
}

// ============================================================================================
// arithmetic self-test: the two shortcuts the MaxSim kernel takes must equal the reference
// formulation (fp32 op + one rounding to fp16) for EVERY pair of fp16 bit patterns.
//   out[0]: REACHABLE pairs (n >= 0, |e| <= n(1+2^-9): a component never exceeds its vector's
//           norm) where h(quot2(e, 1/n as r_hi + r_lo)) != h(fl32(e / n))          -- must be 0
//   out[2]: (informational) the same for the single product h(fl32(e * fl32(1/n)))
//   out[3]: (informational) mismatches of quot2 over ALL pairs (unreachable specials such as
//           e = inf with finite n give inf - inf = NaN instead of inf)
//   out[1]: pairs where (packed fp16 add)(a, b)  != h(fl32(a + b))
// NaN results compare equal to NaN.
// ============================================================================================
__device__ __forceinline__ bool same_h(half_t a, half_t b) {
  uint16_t x = __builtin_bit_cast(uint16_t, a), y = __builtin_bit_cast(uint16_t, b);
  bool nx = (x & 0x7FFF) > 0x7C00, ny = (y & 0x7FFF) > 0x7C00;
  bool zx = (x & 0x7FFF) == 0, zy = (y & 0x7FFF) == 0;  // +0 == -0 (e = -0 gives +0: same dot products)
  return (nx && ny) || (zx && zy) || x == y;
}
__global__ __launch_bounds__(256) void k_selftest_arith(unsigned long long* __restrict__ out) {
  const uint16_t nb = (uint16_t)blockIdx.x;
  const half_t n = __builtin_bit_cast(half_t, nb);
  const float nf = (float)n;
  float r_hi, r_lo;
  recip2(nf, r_hi, r_lo);
  // reachable domain of the MaxSim kernel: n = h(sqrt(sum e_k^2)) >= 0 and no component can
  // exceed its own vector's norm by more than the fp16 rounding of n
  const bool n_ok = !(nb & 0x8000) || (nb & 0x7FFF) > 0x7C00;  // n >= +0, or NaN
  unsigned long long bad_dom = 0, bad_add = 0, bad_plain = 0, bad_all = 0;
  for (uint32_t eb = threadIdx.x * 2; eb < 65536u; eb += 512u) {
    const h2 e = u32_as_h2(eb | ((eb + 1u) << 16));
    uint32_t ea = h2_as_u32(e), eb2 = h2_as_u32(e);
    norm_pair2(ea, eb2, r_hi, r_lo);  // the exact code path of k_maxsim (both registers must agree)
    const h2 qq = u32_as_h2(ea);
    const half_t q0 = qq.x, q1 = (ea == eb2) ? qq.y : (half_t)__builtin_nanf("");
    const half_t d0 = (half_t)((float)e.x / nf), d1 = (half_t)((float)e.y / nf);
    const bool m0 = !same_h(q0, d0), m1 = !same_h(q1, d1);
    bad_all += m0 + m1;
    const bool dom0 = n_ok && !(__builtin_fabsf((float)e.x) > nf * 1.001953125f);
    const bool dom1 = n_ok && !(__builtin_fabsf((float)e.y) > nf * 1.001953125f);
    if (m0 && dom0) {
      ++bad_dom;
      unsigned long long slot = atomicAdd(&out[4], 1ull);
      if (slot < 11) out[5 + slot] = (unsigned long long)(eb & 0xFFFF) | ((unsigned long long)nb << 16) |
                                     ((unsigned long long)__builtin_bit_cast(uint16_t, q0) << 32) |
                                     ((unsigned long long)__builtin_bit_cast(uint16_t, d0) << 48);
    }
    bad_dom += (m1 && dom1);
    bad_plain += (dom0 && !same_h((half_t)((float)e.x * r_hi), d0));
    bad_plain += (dom1 && !same_h((half_t)((float)e.y * r_hi), d1));
    const h2 nn = {n, n};
    const h2 sum = e + nn;
    bad_add += !same_h(sum.x, (half_t)((float)e.x + nf));
    bad_add += !same_h(sum.y, (half_t)((float)e.y + nf));
  }
  if (bad_dom) atomicAdd(&out[0], bad_dom);
  if (bad_add) atomicAdd(&out[1], bad_add);
  if (bad_plain) atomicAdd(&out[2], bad_plain);
  if (bad_all) atomicAdd(&out[3], bad_all);
}
void fpk_selftest_arith(unsigned long long* out_dev, hipStream_t st) {
  hipLaunchKernelGGL(k_selftest_arith, dim3(65536), dim3(256), 0, st, out_dev);
}
