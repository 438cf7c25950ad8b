// Internal declarations shared by the HIP kernels (fp_kernels.hip, fp_synth.hip) and the
// host engine (fp_engine.cpp).  Not part of the C ABI (see include/fastplaid.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <stdint.h>

#define FP_WAVE 64
#define FP_MAX_PROBE 32          // n_ivf_probe supported by the register top-k kernel
#define FP_MAX_CELLS 8192        // q_len * n_ivf_probe cap (LDS sort of probed cells)
#define FP_MAX_SORT 16384        // entries the LDS bitonic sort handles (R, or G*R when sharded): 8 B keys, 128 KiB of LDS
#define FP_SEL_BINS 2048

// Device-resident index (all pointers are device memory).  Layout in HBM:
//   centroids  [C][D]      f16   row-major
//   lut        [256][8/nbits] f16  byte -> bucket weights (bit-reversal + index tables of
//                                  residual_codec.rs:83-140 folded into one table)
//   ivf_pids   [sum ivf_lengths] i32 local doc ids, per cell ascending
//   ivf_off    [P+1]       i64
//   codes      [T]         i32   (narrowed from the reference's i64: C < 2^31)
//   residuals  [T][PR]     u8    PR = D*nbits/8
//   doc_off    [N+1]       i64   (tensor.rs:221-224 cumulative lengths)
//   Tokens are stored SORTED BY CODE inside each document (MaxSim is order-independent over a
//   document's tokens): equal codes sit in adjacent lanes of one gather instruction and coalesce
//   into a single centroid-row fetch.  perm [T] u16 maps stored position -> original position
//   (used only by reconstruct / export / read_doc, which return the original order).
//   ucodes     [U] i32 + uoff [N+1] i64: per-document ascending UNIQUE codes.  The approximate
//              score only depends on the set of codes of a document, so S4 walks this list.
struct FpIndexDev {
  int nbits, dim, pr;
  int64_t C, P, N, T;
  int64_t pid_offset;
  const uint16_t* centroids;
  const uint16_t* lut;
  const int32_t* ivf_pids;
  const int64_t* ivf_off;
  const int32_t* codes;
  const uint8_t* residuals;
  const int64_t* doc_off;
  const uint16_t* perm;    // [T]   original in-document position of stored token i (tokens are stored
                           //       sorted by code inside each document; nullptr = stored in original order)
  const int32_t* ucodes;   // [U]   per-document ascending unique codes (dedup of `codes`)
  const int64_t* uoff;     // [N+1]
  int64_t U;
  int max_doc_len;
  const uint4* pcodes;     // [NL] lines of 128 B (64 B: l0_ppl 4): the unique codes again, packed into whole lines per document (fp_synth.hip,
                           //       "packed unique codes").  Line d < N is document d's first line (addressed by the id, no lookup); the extra
                           //       lines of documents with more codes than a line holds follow behind.  nullptr: no level 0 for this index
  const int32_t* poff;     // [N][2] {first extra line, extra line count} of each document in pcodes (read for flagged first lines only)
  // centroid ranges of 2^17 (FP_L0_RANGE) for tables beyond 2^17 centroids: range r has its own lines / offsets with codes
  // relative to r * 2^17; [0] aliases pcodes / poff
  int n_ranges;            // > 1: pcodes holds the ranges' first lines INTERLEAVED (line d * n_ranges + r = document d, range r: the lines of a
                           // document's ranges share 128-byte fabric requests) and pcodes_r[r] only range r's extra lines
  int l0_ppl;              // 16-byte pieces per code line: 8 (128-byte lines) or 4 (64-byte lines; tables of several ranges)
  int64_t n_lines;         // code lines over all ranges
  const uint4* pcodes_r[8];
  const int32_t* poff_r[8];
  const uint16_t* norms;   // [T]   fp16 bits of h(sqrt(sum_fp32 e_k^2)) per stored token (ascending-k sum; computed at index creation)
  int resid_native;           // 1: every token's residual row is stored in k_maxsim6's unit order (fp_resid_native_unit); the arrays
                              //    handed over by the caller and everything exported are in the reference's order
  const uint32_t* rinv;       // [T]   fp32 bits of a reciprocal r with h(fl32(e_k * r)) == h(fl32(e_k / n)) for every dim of the token
                              //         (bit 31 set: no such r within the search range -> the kernel takes the exact path for the step); nullable
  // S1's view of the centroid table (round 4): dims below 256 other than 64 / 128 are ZERO-PADDED to 64 / 128 / 256 in a second copy
  // [C][dim_s1], so that the centroid scores run through the streaming kernel -- whose exact mode re-evaluates the flagged scores
  // from LDS and registers; the one-tile kernel fetches both rows of every flagged score from L2 (dim 96: S1 0.94 ms against
  // 0.29 without the certification).  Zero products change neither the MFMA sum nor the ascending chain.  nullptr: S1 reads
  // `centroids` (dim 64 / 128 / 256, and dims above 256, which stay on the one-tile kernel).
  const uint16_t* cent_s1;
  int dim_s1;
};

// Native unit order of a token's residual bytes (k_maxsim6): a unit = 8 dims = nbits bytes; unit u = 4 s + g (k-step s of 32 dims,
// lane group g) is stored at position start(g) + s, so that the units one lane decodes are contiguous.  nu = dim / 8.
static inline __host__ __device__ constexpr int fp_resid_native_unit(int u, int nu) {
  const int s = u >> 2, g = u & 3;
  int start = 0;
  for (int i = 0; i < g; ++i) start += (nu - i + 3) >> 2;
  return start + s;
}
// stored position of the reference-order residual byte b of a token (identity when the index is not in native order)
static inline __host__ __device__ constexpr int fp_resid_pos(int b, int nbits, int nu, int native) {
  return native ? fp_resid_native_unit(b / nbits, nu) * nbits + b % nbits : b;
}
// the inverse: reference-order byte held at stored position p
static inline __host__ __device__ constexpr int fp_resid_logical(int p, int nbits, int nu, int native) {
  if (!native) return p;
  const int pu = p / nbits;
  int start = 0, g = 0;
  for (; g < 4; ++g) {
    const int cnt = (nu - g + 3) >> 2;
    if (pu < start + cnt) break;
    start += cnt;
  }
  return (4 * (pu - start) + g) * nbits + p % nbits;
}

// A HIP dispatch carries its grid size in work-items as a 32-bit number: blocks * block_size must
// stay below 2^32, a larger launch silently wraps (seen at 10 M documents: 5.1e9 work-items).
// Kernels whose natural grid scales with the index take this cap and walk a grid-stride loop.
// FP_GRID_CAP=<blocks> (testing) lowers the cap so that small corpora exercise the loops.
// FP_TEST="key=value,key=value,...": the knobs the TEST SUITE turns (forcing rare paths on small corpora, moving thresholds);
// every key is listed with the test that uses it in INTEGRATION.md.  Numbers only; an absent key gives `def`.
static inline double fp_test_opt(const char* key, double def) {
  const char* e = getenv("FP_TEST");
  if (!e) return def;
  const size_t kl = strlen(key);
  for (const char* p = e; *p;) {
    const char* c = strchr(p, ',');
    const size_t n = c ? (size_t)(c - p) : strlen(p);
    if (n > kl + 1 && strncmp(p, key, kl) == 0 && p[kl] == '=') return atof(p + kl + 1);
    if (!c) break;
    p = c + 1;
  }
  return def;
}
static inline unsigned fp_grid_cap(int64_t blocks, int block_size) {
  static const int64_t env_cap = (int64_t)fp_test_opt("grid_cap", 0);
  int64_t cap = 0xFFFFFFFFll / block_size;
  if (env_cap > 0 && env_cap < cap) cap = env_cap;
  return (unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

// ---- launch wrappers implemented in fp_kernels.hip -----------------------------------------
// All run on `st`; none synchronises.
// padded query length: Q rounded up to a multiple of 32 (zero rows appended) -- and to 128 for 64 < Q <= 128, so that a query's
// column groups fill whole 128-column S1 tiles and level 0's excess byte can be summed inside one tile (round 4: queries of 65 ..
// 128 tokens take the bound stage too; before, they scored every candidate exactly)
static inline int fp_padded_qlen(int Q) { return (Q > 64 && Q <= 128) ? 128 : ((Q + 31) & ~31); }
struct FpSearchShape {
  int B;       // queries in this sub-batch
  int Q;       // real query tokens
  int Qp;      // fp_padded_qlen(Q)
  int n_probe;
  int64_t R;   // exact-scoring budget per query = max(n_full/4, 1)
  int64_t n_full;
};

// regions (16-byte aligned, sizes in 16-byte units) that the first kernel of a batch clears on its way: the counters, flags and
// histograms later stages expect zeroed
#define FP_ZERO_REGIONS 6
struct FpZeroList {
  void* p[FP_ZERO_REGIONS];
  uint32_t n16[FP_ZERO_REGIONS];
  int n;
  bool add(void* ptr, size_t bytes) {
    if (n >= FP_ZERO_REGIONS || bytes > ((size_t)1 << 30)) return false;
    p[n] = ptr;
    n16[n] = (uint32_t)((bytes + 15) / 16);
    ++n;
    return true;
  }
};
void fpk_pack_queries(const uint16_t* q_dev_in /*[B,Q,D]*/, uint16_t* q_pad /*[B*Qp,D]*/, int B, int Q, int Qp, int D,
                      hipStream_t st, const FpZeroList* zero = nullptr,
                      float* wcol = nullptr /*[B*Qp]: w0 * |q_n| per packed row (0 for the zero rows), the absolute part of S1's certification window*/,
                      float w0 = 0.f, uint16_t* q_pad2 = nullptr /*[B*Qp, D2]: a second copy with the rows zero-padded to D2 >= D dims (S1's view)*/,
                      int D2 = 0);
// S8 (nullable; written only when Qp is 32, 64 or 128): 8-bit bins [B][Qp/32][C][32]; cmax (nullable): [B*Qp][ceil(C/128)] f16 column maxima per 128-centroid tile
// S1's optional by-product for S4's level 0: the excess table e8[b][Cpad] (one byte per query and centroid) computed in the
// epilogue from the column floors of a sampled pre-pass
struct FpS1Excess {
  const uint8_t* floors;   // [B][Qp]
  const uint16_t* gfl;     // [B][Qp] fp16: g = floor - 100 (2000 for the pad columns); rd != 0: 1024 - g instead
  uint8_t* e8;             // [B][Cpad] (nullptr: off)
  uint32_t* esc;           // [B][64]
  int64_t Cpad;
  int Q;                   // real query columns
  int rd;                  // the epilogue's one-fma form (s1_writeout): gfl holds 1024 - g
};
// Exact centroid scores (round 4).  The MFMA sums the 128 products in another order than the reference's ascending fp32 chain
// (search.rs:491 = ATen's half matmul), so ~0.05 % of the fp16 results differed by one ulp.  S1 now certifies every entry: the fp32
// accumulator x is flagged when a rounding boundary of fp16 lies within +-(wcol[n] + kappa |x|) of it -- only then can another
// summation order round differently -- and every flagged entry is re-evaluated with the ascending chain acc = fma(c_k, q_k, acc)
// inside the kernel (centroid rows from the LDS tile, query rows from the MFMA A fragments), before the tile and its by-products
// (excess bytes, 8-bit bins) leave.  S is then bit-identical to the reference's matmul.
struct FpS1Exact {
  int mode;                    // 0 off, 1 certify + repair, 2 (tests) re-evaluate EVERY entry and count the unflagged differences,
                               // 3 lazy: upper candidates only, the consumers repair what they use (FpLazyS1 below)
  const float* wcol;           // [B*Qp] absolute window per query column: w0 |q_n| (0 for zero rows: their scores are exact zeros)
  float kappa;                 // relative part of the window
  unsigned long long* stats;   // nullable [4]: flagged entries, repaired values that differ from the MFMA's, entries beyond the
                               // LDS list (slow path), mode 2: unflagged differences
};
// S1's LAZY form (round 5; FpS1Exact::mode 3): S1 stores the upper candidate h(x + u) of EVERY score and runs no chain at all; a
// stored value is the reference's or (rarely) one fp16 step above it.  The consumers settle only what they use:
//   * the probe lowers its collection threshold by the window (s1_lower16) and re-evaluates the <= 64 collected scores of a
//     column with the ascending chain before it ranks them (k_probe_merge);
//   * the bound stages read upper bounds anyway (excess bytes, 8-bit bins);
//   * k_approx sums the stored column maxima -- an UPPER bound A_up of the approximate score; how far below it the reference's
//     score can lie is bounded per query from the column maxima (tight / loose / negflag below);
//   * the selection gathers everything within the slack of the R-th largest A_up, keeps what lies above it by more than the slack
//     and recomputes the others ("maybes", a few dozen per query) from scratch with ascending chains (k_sel_finish_lz), so that the
//     selected SET equals the reference's; a list that overflows raises *flag and the caller runs the batch again eagerly.
struct FpLazyS1 {
  const float* wcol;        // [B*Qp] w0 |q_n| cmax (as FpS1Exact)
  float kappa;
  float inv_w0;             // 1 / (w0 x dim scale): wcol * inv_w0 = |q_n| cmax, the bound on any |score| of the column
  const uint16_t* qpad;     // [B*Qp][dim] packed queries (unpadded dim)
  const uint16_t* S;        // [B][C][Qp] the stored scores (upper candidates)
  float* tight;             // [B*Qp] per-column slack bound from the column's overall maximum (k_probe_tau); valid for non-negative maxima
  float* loose;             // [B*Qp] ... from the norm bound: valid for any value
  uint32_t* negflag;        // [B] zeroed; set by k_approx when a scored document has a negative column maximum
  uint32_t* gcount;         // [B] zeroed: maybes gathered by the selection
  int32_t* gpid;            // [B][gcap] their ids
  float* gval;              // [B][gcap] ... and recomputed scores
  int gcap;
  int32_t* flag;            // zeroed; != 0: a list overflowed -> results of the batch are not to be used
  uint32_t* stats;          // nullable [4]: certain entries, maybes, -, -
};
// the column maxima S1 emits (cmax) are taken over the UPPER candidates h(x + w): >= the exact maximum, at most one fp16 step above
int fpk_centroid_scores(const FpIndexDev& ix, const uint16_t* qpad, uint16_t* S, int B, int Qp, uint8_t* S8, uint16_t* cmax,
                        hipStream_t st, int64_t n_rows = 0, int64_t row_stride = 1, const FpS1Excess* exc = nullptr,
                        const FpS1Exact* exact = nullptr);
int fpk_probe(const FpIndexDev& ix, const uint16_t* S, const FpSearchShape& sh, const uint32_t* allow /*[B][Cw] or null*/,
              unsigned long long* partial, int nchunk, int32_t* cells /*[B][Q][n_probe]*/, int32_t* ucells /*[B][Q*n_probe]*/,
              int32_t* ncells /*[B]*/, const uint16_t* cmax128 /*nullable: S1's [B*Qp][ceil(C/128)] column maxima*/, hipStream_t st,
              bool prezeroed = false /*the region of fpk_probe_zero_region was cleared earlier in the stream*/,
              bool with_fallback = true /*false: a column with too many ties at its threshold leaves *fpk_probe_flag != 0 and NO cells;
                                          the caller reads the flag back and runs the batch again with the fallback*/,
              const FpS1Exact* cmax_upper = nullptr /*cmax128 came from S1's exact mode: maxima over the upper candidates; the threshold is lowered by the window (wcol, kappa)*/,
              const struct FpLazyS1* lz = nullptr /*S holds S1's lazy form (upper candidates everywhere): sound threshold + exact re-evaluation of the collected scores*/);
// does this shape take the threshold probe with S1's column maxima (the only probe path the lazy form of S1 serves)?
bool fpk_probe_lazy_ok(const FpIndexDev& ix, const FpSearchShape& sh, int nchunk);
size_t fpk_probe_scratch_bytes(const FpIndexDev& ix, const FpSearchShape& sh, int nchunk);
// the counters + overflow flag the threshold probe expects zeroed (inside `partial`); false = this shape takes the other path
bool fpk_probe_zero_region(const FpIndexDev& ix, const FpSearchShape& sh, int nchunk, unsigned long long* partial, void** p, size_t* bytes);
const int32_t* fpk_probe_flag(const FpIndexDev& ix, const FpSearchShape& sh, int nchunk, const unsigned long long* partial);
void fpk_subset_prepare(const FpIndexDev& ix, const int64_t* sub_ids, const int64_t* sub_off /*[B+1] dev*/, int B,
                        uint32_t* subbm, int64_t W, uint32_t* allow, int64_t Cw, int32_t* invalid, hipStream_t st,
                        int64_t max_len /*the longest list*/, int replicate_to = 0 /*B == 1: the list serves this many queries (rows 1.. = row 0)*/);
void fpk_ivf_mark(const FpIndexDev& ix, const int32_t* ucells, const int32_t* ncells, int maxcells, int B,
                  uint32_t* bitmap, int64_t W, hipStream_t st);
// candidate compaction: count -> scan -> offsets -> compact (ascending doc ids)
void fpk_cand_count(const uint32_t* bitmap, const uint32_t* subbm, const int32_t* invalid, int B, int64_t W,
                    int32_t* blkcnt, int nblk, int32_t* ncand, int64_t* cand_off /*[B+1]*/, hipStream_t st,
                    int64_t cap /*0: none*/, int32_t* invalid_rw, int64_t* total_out /*{total, -, int32 probe flag at byte 16}*/,
                    uint32_t* tickets = nullptr /*[B + 1] zeroed counters -> one launch instead of three*/,
                    const int32_t* probe_flag = nullptr);
int fpk_cand_words_per_block();   // bitmap words one workgroup of the candidate counting / compaction kernels covers
void fpk_cand_compact(const uint32_t* bitmap, const uint32_t* subbm, const int32_t* invalid, int B, int64_t W,
                      const int32_t* blkcnt /*exclusive-scanned*/, int nblk, const int64_t* cand_off, int32_t* cand_pid,
                      hipStream_t st);
// candidate rows start at cand_off[b] (CSR) or, without cand_off, at b * cap; a row holds min(cnt[b], cap) entries when cnt is
// given, else the whole CSR row.  Scores go to approx[row position] (if given) and/or scat[scat_off[b] + scat_idx[row position]].
void fpk_approx(const FpIndexDev& ix, const uint16_t* S, const FpSearchShape& sh, const int64_t* cand_off,
                const int32_t* cand_pid, int64_t M, float* approx, hipStream_t st, const int32_t* cnt = nullptr, int64_t cap = 0,
                float* scat = nullptr, const int32_t* scat_idx = nullptr, const int64_t* scat_off = nullptr,
                const FpLazyS1* lz = nullptr /*S is S1's lazy form: the scores are upper bounds; lz->negflag[b] is raised by a negative column maximum*/);
// top-R selection by (approx desc, doc id asc); output in ascending doc id order
// bound-and-refine front of S4 (see fp_kernels.hip): 8-bit bins of S, per-candidate bin sums, per-query cut, ordered survivors
#define FP_SURV_CHUNK 2048
#define FP_L0_CHUNK 8192   // candidates per workgroup of level 0's survivor count / compaction
void fpk_approx_q8_bounds(const FpIndexDev& ix, const uint8_t* S8, const FpSearchShape& sh, const int64_t* cand_off, const int32_t* cand_pid,
                          int64_t M, uint32_t* kq, hipStream_t st);
void fpk_approx_q8_cut(const FpSearchShape& sh, const int64_t* cand_off, const int32_t* cand_pid, int64_t M, uint32_t* q8hist, uint32_t* kq,
                       int32_t* cut, int32_t* blkcnt, int nblk, int32_t* nsurv, int64_t* surv_off, int32_t* surv_pid, hipStream_t st,
                       bool lazy_bins = false /*the bins came from S1's lazy form*/);
// level 0 of S4 (see fp_kernels.hip): scalar excess bound per centroid in LDS, pilot group, survivors
struct FpL0Scratch {
  uint8_t* floors;      // [B][Qp]
  uint32_t* Fsum;       // [B]
  uint8_t* e8;          // [B][Cpad]
  uint32_t* esc;        // [B][64]
  uint16_t* ub;         // [M]
  uint32_t* hist;       // [B][4096]
  int32_t* cut;         // [B]
  int32_t* blkcnt;      // [B][nblk] survivors per chunk
  int32_t* blkcntx;     // [B][nblk] survivors outside the pilot group per chunk
  int nblk;             // ceil(max candidates per query / FP_L0_CHUNK)
  int32_t* npilot;      // [B] (may exceed the capacity: then nothing is pruned)
  int32_t* pilot_pid;   // [B][fpk_l0_pilot_cap()]
  float* pilot_approx;  // [B][fpk_l0_pilot_cap()]
  int32_t* pilot_idx;   // [B][fpk_l0_pilot_cap()] position of the pilot document in the query's candidate list
  float* cand_approx;   // [M] exact scores by candidate position (only the pilot members' entries are written / read)
  int32_t* thr;         // [B] survivor threshold on ub (0 = keep everything)
  int32_t* nextra;      // [B] survivors outside the pilot group
  int32_t* xpid;        // [M] rows at surv_off[b]: their document ids ...
  int32_t* xdst;        // [M] ... and positions in the survivor list
  uint32_t* tickets;    // nullable: [B + 1] zeroed counters -> count + scans + offsets of the survivor lists in ONE launch
  uint16_t* ub_parts;   // nullable: [n_ranges][ub_stride] partial bounds of a multi-range table (all ranges in one launch)
  int64_t ub_stride;
};
struct FpL0Multi {         // the per-range arrays of a multi-range table, by value to the scan kernel
  int nr;
  int seq_r;               // one launch per range: the range of this launch
  int pair;                // 64-byte lines read as 128-byte pairs of ranges by 8 lanes (half of them count)
  int64_t ub_stride;       // entries between the ranges' partial-bound arrays
  const uint4* x[8];       // extra lines of range r
  const int32_t* po[8];    // {first extra line, count} of range r
};
bool fpk_l0_fits(const FpIndexDev& ix);
void fpk_l0_prepare(const FpIndexDev& ix, const uint8_t* S8 /*nullptr: floors and table came with S1*/, const FpSearchShape& sh, FpL0Scratch& w,
                    hipStream_t st, bool hist_prezeroed = false);
size_t fpk_l0_hist_bytes(int B);
size_t fpk_sel_hist_bytes(int B);
void fpk_l0_sample_plan(const FpIndexDev& ix, int64_t* n_rows, int64_t* stride);
void fpk_l0_floors(const uint8_t* S8_sample, int64_t n_rows, const FpSearchShape& sh, uint8_t* floors, uint32_t* Fsum, uint32_t* esc,
                   uint16_t* gfl /*[B][Qp] fp16 floors for S1's epilogue*/, hipStream_t st, float hot_tail = 0.f /*> 0: k_l0h_scan's floors, the (1 - hot_tail) quantile*/,
                   int gfl_rd = 0 /*FpS1Excess::rd*/);
// level 0 for documents with many distinct codes: per-column maxima over the HOT codes only (k_l0h_scan); same outputs as fpk_l0_scan
bool fpk_l0h_fits(const FpIndexDev& ix);
void fpk_l0h_scan(const FpIndexDev& ix, const uint16_t* S, const FpSearchShape& sh, const int64_t* cand_off, const int32_t* cand_pid, int64_t M,
                  FpL0Scratch& w, hipStream_t st);
void fpk_l0_scan(const FpIndexDev& ix, const FpSearchShape& sh, const int64_t* cand_off, const int32_t* cand_pid, int64_t M, FpL0Scratch& w,
                 hipStream_t st);
void fpk_l0_pilot(const FpSearchShape& sh, const int64_t* cand_off, const int32_t* cand_pid, int64_t M, FpL0Scratch& w, hipStream_t st);
int64_t fpk_l0_pilot_cap();
void fpk_l0_survivors(const FpSearchShape& sh, const int64_t* cand_off, const int32_t* cand_pid, FpL0Scratch& w, int32_t* nsurv,
                      int64_t* surv_off, int32_t* surv_pid, float* surv_approx, hipStream_t st,
                      const FpLazyS1* lz = nullptr /*S1's lazy form: the pilot scores are upper bounds*/);
int fpk_select(const FpSearchShape& sh, const int64_t* cand_off, const int32_t* cand_pid, const float* approx,
                uint32_t* hist /*[3][B][BINS]*/, uint32_t* selstate /*[B][8]*/, int32_t* sel_pid /*[B][R]*/,
                float* sel_approx /*[B][R]*/, int32_t* sel_cnt /*[B]*/, int32_t* tie_pid /*[B][R] scratch*/, hipStream_t st,
                bool short_lists = false /*lists of a few thousand entries: one workgroup per query does the whole radix select*/,
                bool hist_prezeroed = false,
                int64_t* pref = nullptr /*the last kernel also writes the [B + 1] prefix of sel_cnt here*/,
                const FpLazyS1* lz = nullptr /*approx[] are upper bounds from S1's lazy form (general path only: !short_lists, R <= FP_MAX_SORT)*/,
                const FpIndexDev* ix = nullptr /*needed with lz: the maybes are recomputed from the index*/,
                int64_t est_per_query = 0 /*expected list length per query (sizes the grids; 0 = unknown)*/);
// S1's lazy form: can the selection serve this shape, and how many gathered entries per query does it want room for?
bool fpk_select_lazy_ok(const FpSearchShape& sh);
int fpk_select_lazy_gcap(const FpSearchShape& sh);
// ---- fp_maxsim.hip ---------------------------------------------------------------------------
// per-token norms of a freshly laid-out index (centroids / lut / codes / residuals set in `ix`)
void fpk_token_norms(const FpIndexDev& ix, uint16_t* norms, hipStream_t st);
// by-products of the MaxSim kernel for the exact-order repair (all three or none)
struct FpMaxsimAux {
  uint16_t* cm16;    // [B][Rcap][Qp] per-column maxima (fp16 bits)
  float* unc;        // [B][Rcap]     uncertainty budget: sum of the fp16 ulps of the flagged columns (0 = certainly the reference's score)
  uint32_t* flags;   // [B][Rcap][Qp/32] flagged columns
  float* uncm;       // [B][Rcap]     the part of `unc` by which the reference's score may be lower (it may be higher by unc - uncm)
};
bool fpk_maxsim_fast_shape(int dim, int nbits);
bool fpk_maxsim6_shape(int dim, int nbits);   // shapes k_maxsim6 is instantiated for
// rows [t0, t0 + n) of ix.residuals rewritten in k_maxsim6's unit order through tmp (n * pr bytes)
void fpk_resid_native(const FpIndexDev& ix, int64_t t0, int64_t n, uint8_t* tmp, hipStream_t st);
// per-token reciprocals for k_maxsim6's one-multiply normalisation (needs ix.norms); n_hard_dev (nullable): device counter of the
// tokens that keep the exact path
void fpk_token_rinv(const FpIndexDev& ix, uint32_t* rinv, unsigned long long* n_hard_dev, hipStream_t st);
// exact scores of the rerank lists; pref = [B+1] int64 scratch
int fpk_maxsim(const FpIndexDev& ix, const uint16_t* q_pad, const FpSearchShape& sh, const int32_t* sel_pid, const int32_t* sel_cnt,
               int64_t Rcap, float* exact /*[B][Rcap]*/, int64_t* pref, const FpMaxsimAux& aux, hipStream_t st,
               bool pref_ready = false /*the selection already left the prefix of sel_cnt in pref (fpk_select's pref argument)*/);
// flagged documents that are near-tied in the final ranking -> marks [B][stride], nmark [B]; -1 when stride is too large for LDS
int fpk_final_mark(const float* score, const float* unc, const float* uncm /*nullable: symmetric*/, const int32_t* cnt, int64_t stride, int B,
                   int64_t top_k, int32_t* marks, int32_t* nmark,
                   hipStream_t st, uint32_t* flat_n = nullptr /*a zeroed counter ...*/, void* flat = nullptr /*... and the batch-wide {query, slot} list [B * stride] it counts*/);
// exact (ascending-k) re-evaluation of the flagged columns of the marked documents (marks == nullptr: of every flagged document)
void fpk_maxsim_repair(const FpIndexDev& ix, const uint16_t* q_pad, const FpSearchShape& sh, const int32_t* sel_pid, const int32_t* sel_cnt,
                       int64_t Rcap, const int32_t* marks, const int32_t* nmark, float* exact, const FpMaxsimAux& aux, hipStream_t st,
                       const uint32_t* flat_n = nullptr, const void* flat = nullptr /*fpk_final_mark's batch-wide list: one wave per entry*/);
// final ranking: sort (score desc, id asc), emit top_k with pid_offset applied; 0, or the error of the segmented device sort
// that takes over beyond FP_MAX_SORT entries per query (a hipError, -1 when B * stride >= 2^31): nothing was written then
int fpk_final_topk(const float* score /*[B][stride]*/, const int32_t* pid_local /*[B][stride] or null*/,
                    const int64_t* pid_global /*[B][stride] or null*/, const int32_t* cnt /*[B] or null -> stride*/,
                    int64_t stride, int B, int64_t top_k, int64_t pid_offset, int64_t* out_pid /*[B][top_k]*/,
                    float* out_score, int32_t* out_cnt, hipStream_t st, const int64_t* stat_total = nullptr, const int32_t* stat_per_query = nullptr,
                    int64_t* stat_out /*[4 + B]: statistics copied next to the results*/ = nullptr,
                    const int32_t* stat_flag = nullptr /*-> stat_out[1 + B]*/,
                    const int64_t* stat_cand = nullptr /*S3's {total, -, probe flag} block -> stat_out[2 + B], [3 + B]*/);
// sharded helpers (record layouts: include/fastplaid.h)
void fpk_shard_pack1(const float* sel_approx, const int32_t* sel_pid, const int32_t* sel_cnt, int B, int64_t R, int64_t pid_offset, void* rec1,
                     hipStream_t st, const int64_t* cand_total = nullptr, int64_t cand_cap = 0, int status = 0);
void fpk_shard_any_overflow(const void* all_rec1, int G, int B, int64_t R, int32_t* flag, hipStream_t st);
// *flag |= OR over the ranks' blocks (stride_bytes apart) of the 32-bit status word at byte word_off (bit 0 overflow, bit 1 failure)
void fpk_shard_status(const void* all, int G, int64_t stride_bytes, int64_t word_off, int32_t* flag, hipStream_t st);
void fpk_shard_pack2(const float* score, const float* unc /*nullable*/, const float* uncm /*nullable*/, const int32_t* sel_pid,
                     const int32_t* sel_cnt, int B, int64_t R, int64_t pid_offset, void* rec2, hipStream_t st, int status = 0);
int fpk_shard_global_cut(const void* all_rec1 /*[G][B][R]*/, int G, int B, int64_t R, int64_t pid_lo, int64_t pid_hi,
                         int32_t* sel_pid /*[B][R] local*/, int32_t* sel_cnt, hipStream_t st);
// union of the ranks' survivors in ascending id order; u_src = rank * R + slot in that rank's rerank list
int fpk_shard_union(const void* all_rec2 /*[G][B][R]*/, int G, int B, int64_t R, int64_t* u_pid, float* u_score, int32_t* u_src, float* u_unc,
                    float* u_uncm, int32_t* u_cnt, hipStream_t st);
// third exchange (marks = union positions of the near-tied documents, nullptr = every flagged document)
void fpk_shard_local_marks(const int32_t* marks, const int32_t* nmark, const float* u_unc, const int32_t* u_cnt, const int32_t* u_src, int B,
                           int64_t R, int rank, int32_t* lmarks, int32_t* lnmark, hipStream_t st);
void fpk_shard_pack3(const int32_t* marks, const int32_t* nmark, const float* u_unc, const int32_t* u_cnt, const int32_t* u_src, int B, int64_t R,
                     int rank, const float* exact_local, float* x /*[B][R]*/, hipStream_t st);
void fpk_shard_apply3(const int32_t* marks, const int32_t* nmark, const float* u_unc, const int32_t* u_cnt, const int32_t* u_src, int B, int64_t R,
                      const float* xall /*[G][xstride]; xstride 0 = B * R*/, float* u_score, hipStream_t st, int64_t xstride = 0);
// misc
void fpk_narrow_i64_i32(const int64_t* in, int32_t* out, int64_t n, int64_t add, hipStream_t st, int64_t limit = 0, uint32_t* bad = nullptr);
void fpk_ivf_check_sorted(const int64_t* off, const int32_t* pids, int64_t P, uint32_t* flag, hipStream_t st);
// create.rs:148-184, :404-428 on device pointers: nearest centroid (MFMA narrowing + exact re-check of the near-ties, or the exact
// kernel alone; codes32 scratch, codes64 output) + quantised packed residuals.  cmaxabs = max |centroid element| (bounds the MFMA
// summation error); work = fpk_compress_work_bytes(T) bytes of device scratch (nullptr: exact kernel only).
// -1 unsupported dim, -2 chunk too large, -3 HIP error
size_t fpk_compress_work_bytes(int64_t T);
int fpk_compress(const uint16_t* emb, int64_t T, const uint16_t* cent, int64_t C, int D, int nbits, const uint16_t* cutoffs, float cmaxabs,
                 int32_t* codes32, int64_t* codes64, uint8_t* out, void* work, hipStream_t st);
// k-means assignment step: argmax_c (dot - half_sqnorm[c]) in fp32, ties -> lowest index
int fpk_assign_l2(const uint16_t* emb, int64_t T, const uint16_t* cent, const float* half_sqnorm, int64_t C, int D, int32_t* codes32,
                  int64_t* codes64, hipStream_t st);
// [q_len, doc_len] fp16 similarity matrices of (query, doc) hits; -1 when q_len * dim does not fit LDS
int fpk_token_scores(const FpIndexDev& ix, const uint16_t* queries, int Q, const int32_t* hit_query, const int32_t* hit_pid, int64_t n_hits,
                     const int64_t* out_off, uint16_t* out, hipStream_t st);
void fpk_reconstruct(const FpIndexDev& ix, const int64_t* tok_idx /*[n] global token rows*/, int64_t n, float* out,
                     hipStream_t st);

void fpk_selftest_arith(unsigned long long* out_dev /*[2], zeroed*/, hipStream_t st);
// the fence-free "last workgroup finishes the job" pattern checked on this device: 0 = holds (synchronises `st`)
int fpk_ticket_selftest(hipStream_t st);

// ---- fp_synth.hip ----------------------------------------------------------------------------
struct FpSynthParams {
  int nbits, dim, pr;
  int64_t C;
  int lgC;
  int64_t n_docs_total, doc_begin, doc_end;
  int doc_len;
  int variable_len;
  uint64_t seed;
};
// Fills doc_off (N+1) on the HOST (caller uploads) -- lengths are a pure hash; returns T.
int64_t fps_doc_offsets_host(const FpSynthParams& p, int64_t* doc_off_host, int* max_len);
// Device generation of codes/residuals for the shard; tok_base = global index of the
// shard's first token.
void fps_generate(const FpSynthParams& p, const int64_t* doc_off_dev, int64_t n_docs, int64_t T, int64_t tok_base,
                  int32_t* codes, uint8_t* residuals, hipStream_t st);
// Sorts the tokens of every document by (code, original position) in place (documents longer
// than 4096 tokens are left as they are); allocates *perm [T] u16 and a temporary residual
// copy.  Returns 0 or a hipError.
int fps_sort_docs(int32_t* codes, uint8_t* residuals, const int64_t* doc_off_dev, int64_t n_docs, int64_t T, int max_len, int pr,
                  uint16_t** perm, hipStream_t st);
// Per-document sorted unique codes (allocates *ucodes, fills uoff_dev [N+1]).
int fps_build_ucodes(const int32_t* codes, const int64_t* doc_off_dev, int64_t n_docs, int max_len, int32_t** ucodes, int64_t* U,
                     int64_t* uoff_dev, hipStream_t st);
// The unique codes in [code_lo, code_hi) of every document packed into lines (S4 level 0): N first lines + the extra lines.  Allocates *lines and *poff [N][2].
int fps_build_pcodes(const int32_t* ucodes, const int64_t* uoff_dev, int64_t n_docs, int64_t code_lo, int64_t code_hi, void** lines,
                     int32_t** poff, int64_t* n_lines, hipStream_t st, int ppl /*16-byte pieces (6 codes each) per line: 8 or 4*/,
                     int64_t n_centroids /*of the whole table: fixes the pad slot of this range*/,
                     void* shared_first = nullptr /*multi-range tables: the interleaved first lines [n_docs * nr] -> *lines holds the extra lines only*/,
                     int nr = 1, int r = 0);
// final ranking of rerank lists beyond the LDS sort of k_final_topk (segmented device radix sort; synchronises); 0 or a hipError / -1
int fps_final_topk_big(const float* score, const int32_t* pid_local, const int64_t* pid_global, const int32_t* cnt, int64_t stride, int B,
                       int64_t top_k, int64_t pid_offset, int64_t* out_pid, float* out_score, int32_t* out_cnt, hipStream_t st);
// Builds the IVF (per-cell ascending unique local doc ids) from the per-document unique codes.
// Allocates *ivf_pids (hipMalloc) and fills ivf_off_dev [P+1].  Returns 0 or a hipError.
int fps_build_ivf(const int32_t* ucodes, const int64_t* uoff_dev, int64_t n_docs, int64_t U, int64_t P, int32_t** ivf_pids,
                  int64_t* ivf_total, int64_t* ivf_off_dev, hipStream_t st);
