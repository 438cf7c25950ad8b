// tools/probe/maxsim7_lab.hip -- the "distinct centroid rows through LDS" rebuild of the MaxSim kernel (round 5), kept OUT of the
// product: it is bit-identical to k_maxsim6 (37 parity tests, 64 / 64 id lists at cfg2) and SLOWER -- 351 .. 382 us against 223 us
// (profiles/r05_maxsim7_lab.txt).  Build: tools/maxsim7_lab.sh makes a variant library in which this translation unit replaces
// fp_maxsim.o (it includes the product source and hooks into fpk_maxsim); select it with FP_LIB_PATH.
#define FP_MAXSIM_LAB_HOOK 1
#include "../../fast-plaid_amd/csrc/fp_internal.h"
static int fpk_maxsim_lab(const FpIndexDev& ix, const uint16_t* qpad, const FpSearchShape& sh, const int32_t* sel_pid, const int32_t* sel_cnt,
                          int64_t Rcap, float* exact, int64_t* pref, const FpMaxsimAux& aux, hipStream_t st, bool pref_ready);
#include "../../fast-plaid_amd/csrc/fp_maxsim.hip"

// ======================================================================================================================
// k_maxsim7 -- every DISTINCT centroid row through LDS once (round 5; dim 128, one 32-column query chunk).
//
// k_maxsim6 fetches a token's centroid row with per-lane gathers: four dwordx4 per lane and 16-token step, 16 rows per
// instruction, although a document's tokens are stored sorted by code and 128 tokens carry only ~33 distinct codes.  The
// access-stream labs (tools/probe/gather_lab.hip M7, xcd_range_lab.hip; profiles/r04_*) put that pattern at 173 - 187 us for
// cfg2's rerank set and "distinct rows staged once, four waves per document" at 141 - 147 us.  This kernel is built on the
// latter:
//   * a workgroup of 16 waves (one per CU; the 64 KiB decode table, the query fragments and the row buffers share the LDS) is
//     four SUB-GROUPS of four waves; a sub-group walks its own documents 64 tokens (a GROUP) at a time, one 16-token step per
//     wave; its four waves meet once per group at an LDS counter (the other sub-groups run on);
//   * the group's distinct rows (tokens are sorted by code: a new row starts where the code changes; <= 24 kept per buffer, three
//     buffers per sub-group) are brought into LDS with global_load_lds_dwordx4 -- 16 lanes per 256-byte row, no VGPR staging,
//     XOR-swizzled by row so that the MFMA-order reads spread over the banks -- TWO groups ahead of their use; the residual
//     bytes and reciprocals of a group are fetched four, its codes five groups ahead;
//   * a lane's A operand comes from its token's row by four ds_read_b128 (slot = number of row starts in front of the token);
//     rows beyond a buffer's 24 (groups of many distinct codes) fall back to the per-lane gather;
//   * decode, normalisation, MFMA, fp32 column maxima, certification and sums are k_maxsim6's; a document's maxima are combined
//     over its four waves through LDS behind the group barrier.
// vmcnt retires in order and the DMA loads are inline assembly (the compiler would otherwise order every LDS access behind
// them, see k_centroid_scores_stream): per group a wave issues exactly 3 loads in front of the meeting point (residuals,
// reciprocal, codes) and 2 DMA instructions per group, so "all but the last two iterations' loads" is vmcnt(8).
#define MS7_WAVES 16
#define MS7_SG 4
#define MS7_ROWS 24
#define MS7_NBUF 3
struct Ms7Grp { long long off; int g0, len, r, valid; };   // one 64-token group of one document; wave-uniform
struct Ms7Stage {                 // what a wave holds of a group while it moves through the pipeline
  uint32_t rw[4];                 // the lane's residual bytes (16 at nbits 4)
  uint32_t nrm;                   // reciprocal bits / fp16 norm of the lane's token
  int32_t own;                    // code of the lane's own token (row gather fallback)
  int slot;                       // rows in front of the lane's token inside the group
  int nrows;                      // distinct rows of the group (wave-uniform)
};
#define MS7_LDS_QS 65536
#define MS7_LDS_QN (MS7_LDS_QS + 2 * 4 * 64 * 16)
#define MS7_LDS_ROWS (MS7_LDS_QN + 128)
#define MS7_LDS_RC (MS7_LDS_ROWS + MS7_SG * MS7_NBUF * MS7_ROWS * 256)
#define MS7_LDS_XCH (MS7_LDS_RC + MS7_WAVES * 256)
#define MS7_LDS_CTL (MS7_LDS_XCH + 2 * MS7_SG * 4 * 32 * 4)
#define MS7_LDS_DUMP (MS7_LDS_CTL + 64)        // 1 KiB nobody reads: where a DMA instruction without rows puts its 64 x 16 bytes
#define MS7_LDS_BYTES (MS7_LDS_DUMP + 1024)
template <int NBITS>
__global__ __launch_bounds__(MS7_WAVES * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_maxsim7(const MsArgs a) {
  using Cf = Ms6Cfg<4, NBITS>;
  constexpr int KS4 = 4, D = 128, NE = Cf::NE, RW = Cf::RW, PR = Cf::PR, LB = Cf::LB, NC16 = 2;
  static_assert(LB == 16, "one dwordx4 of residual bytes per lane (the wait counts below assume three loads per group)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* lut = smem;
  uint4* qs = reinterpret_cast<uint4*>(smem + MS7_LDS_QS);
  float* qn = reinterpret_cast<float*>(smem + MS7_LDS_QN);
  float* xch = reinterpret_cast<float*>(smem + MS7_LDS_XCH);
  uint32_t* ctl = reinterpret_cast<uint32_t*>(smem + MS7_LDS_CTL);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r16 = lane & 15, g = lane >> 4;
  const int sg = wave >> 2, wv = wave & 3;
  int32_t* rowcode = reinterpret_cast<int32_t*>(smem + MS7_LDS_RC) + wave * 64;
  ms_lds_base_is_zero(smem);
  ms6_fill_lut<NBITS>(lut, a.lut_g, tid, MS7_WAVES * 64);
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;   // (0: the table sits first)
  const uint32_t laneoff = (uint32_t)((lane & (Cf::COPIES - 1)) * Cf::EW * 4);
  const half_t negm = (half_t)NEG_MASK_F;
  int nq = a.Q - a.ch_begin * 32;
  nq = nq < 0 ? 0 : (nq > 32 ? 32 : nq);
  const int nflag = a.Qp / 32;
  const long long tot = a.pref[a.B];
  const long long lo = tot * (long long)blockIdx.x / (long long)gridDim.x;
  const long long hi_end = tot * (long long)(blockIdx.x + 1) / (long long)gridDim.x;
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
    for (int i = tid; i < NC16 * KS4 * 64; i += MS7_WAVES * 64) {
      const int ln = i & 63, s = (i >> 6) % KS4, c = (i >> 6) / KS4;
      const int col = a.ch_begin * 32 + c * 16 + (ln & 15);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (col < a.Qp) v = *reinterpret_cast<const uint4*>(a.qpad + ((int64_t)b * a.Qp + col) * D + 32 * s + 8 * (ln >> 4));
      qs[i] = v;
    }
    if (tid < 32) {
      const int col = a.ch_begin * 32 + tid;
      float ss = 0.f;
      if (col < a.Qp) {
        const uint16_t* qp = a.qpad + ((int64_t)b * a.Qp + col) * D;
        for (int k = 0; k < D; ++k) { const float x = (float)__builtin_bit_cast(half_t, qp[k]); ss = __builtin_fmaf(x, x, ss); }
      }
      qn[tid] = a.eps_rel * __builtin_sqrtf(ss);
    }
    if (tid < MS7_SG) ctl[tid] = 0u;   // the sub-groups' barrier counters
    __syncthreads();
    const int32_t* selp = a.sel_pid + (int64_t)b * a.Rcap;
    float* outp = a.exact + (int64_t)b * a.Rcap;
    typedef const __attribute__((address_space(4))) int32_t* ms_cptr_i32;
    typedef const __attribute__((address_space(4))) int64_t* ms_cptr_i64;
    const ms_cptr_i32 selc = (ms_cptr_i32)(uintptr_t)selp;
    const ms_cptr_i64 doffc = (ms_cptr_i64)(uintptr_t)a.doc_off;
    auto emit_empty = [&](int rr) {   // (wave wv == 0 of the sub-group) a document without tokens: the masked value in every column
      if (lane == 0) {
        const float v = (float)nq * NEG_MASK_F;
        outp[rr] = a.accumulate ? (outp[rr] + v) : v;
        if (a.unc && !a.accumulate) { a.unc[(int64_t)b * a.Rcap + rr] = 0.f; a.uncm[(int64_t)b * a.Rcap + rr] = 0.f; }
      }
      if (a.cm16 && lane < 32 && a.ch_begin * 32 < a.Qp) a.cm16[((int64_t)b * a.Rcap + rr) * a.Qp + a.ch_begin * 32 + lane] = __builtin_bit_cast(uint16_t, negm);
      if (a.flags && lane == 0 && a.ch_begin < nflag) a.flags[((int64_t)b * a.Rcap + rr) * nflag + a.ch_begin] = 0u;
    };
    // the sub-group's documents: ra + sg, ra + sg + 4, ...; its groups in order
    int it_r = ra + sg, it_g0 = 0, it_len = 0;
    long long it_off = 0;
    auto seek = [&]() {
      while (it_r < rb) {
        const int32_t pid = selc[it_r];
        it_off = doffc[pid];
        it_len = (int)(doffc[pid + 1] - it_off);
        if (it_len != 0) break;
        if (wv == 0) emit_empty(it_r);
        it_r += MS7_SG;
      }
      it_g0 = 0;
    };
    seek();
    auto next_grp = [&]() -> Ms7Grp {
      Ms7Grp gp{it_off, it_g0, it_len, it_r, it_r < rb ? 1 : 0};
      if (gp.valid) {
        it_g0 += 64;
        if (it_g0 >= it_len) { it_r += MS7_SG; seek(); }
      }
      return gp;
    };
    auto issue_loads = [&](Ms7Stage& S, const Ms7Grp& d) {   // two vector loads, always (an exhausted stream reads row 0)
      int tok = d.g0 + 16 * wv + r16;
      tok = d.valid ? (tok < d.len ? tok : d.len - 1) : 0;
      const long long row = (d.valid ? d.off : 0) + tok;
      const uint4 v = *reinterpret_cast<const uint4*>(a.resid + row * (long long)PR + g * LB);
      S.rw[0] = v.x; S.rw[1] = v.y; S.rw[2] = v.z; S.rw[3] = v.w;
      S.nrm = a.rinv ? a.rinv[row] : (uint32_t)a.norms[row];
    };
    auto load_codes = [&](const Ms7Grp& d) -> int32_t {   // code of group token `lane` (clamped); one vector load, always
      int ct = d.g0 + lane;
      ct = d.valid ? (ct < d.len ? ct : d.len - 1) : 0;
      return a.codes[(d.valid ? d.off : 0) + ct];
    };
    // row starts of a group (its codes have landed) -> slots, the row list, and the group's rows on their way into `buf`
    auto stage_rows = [&](Ms7Stage& S, const Ms7Grp& d, uint32_t buf, const int32_t c) {
      const int ntok = d.valid ? (d.len - d.g0 < 64 ? d.len - d.g0 : 64) : 0;
      const int32_t prev = __shfl_up(c, 1, 64);
      const bool isnew = lane < ntok && (lane == 0 || c != prev);
      const unsigned long long mask = __ballot(isnew);
      const int nrows = __popcll(mask);
      const int myslot = __popcll(mask & ((2ull << lane) - 1ull)) - 1;   // (lane < ntok: >= 0)
      if (isnew) rowcode[myslot] = c;
      S.nrows = nrows;
      S.slot = __shfl(myslot, 16 * wv + r16, 64);
      S.own = __shfl(c, 16 * wv + r16, 64);
      asm volatile("" ::: "memory");   // (the list is read back below by other lanes of this wave: LDS is in order within a wave)
      const int keep = nrows < MS7_ROWS ? nrows : MS7_ROWS;
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        const int R0 = 4 * (wv + 4 * j2);
        const int row = R0 + (lane >> 4);
        const bool act = row < keep && R0 + 4 <= MS7_ROWS;
        const int32_t code = keep > 0 ? rowcode[act ? row : keep - 1] : 0;
        // lane l's 16 bytes land at M0 + 16 l: position (row, l & 15) holds the row's chunk (l & 15) ^ (row & 15); lanes without a
        // row re-read the first chunk of the group's last row (one line, already on its way)
        const uint32_t voff = (uint32_t)code * 256u + (act ? (uint32_t)(((lane & 15) ^ (row & 15)) * 16) : 0u);
        // (every wave issues TWO instructions, so that the wait counts are the same for all; 24 rows are six of them: the second
        // one of waves 2 and 3 has no rows and writes to the dump)
        const uint32_t m0v = lds0 + (R0 + 4 <= MS7_ROWS ? buf + (uint32_t)R0 * 256u : (uint32_t)MS7_LDS_DUMP);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(a.cent), "s"(m0v) : "memory", "m0");
#pragma clang diagnostic pop
      }
    };

    float mx[NC16];
#pragma unroll
    for (int c = 0; c < NC16; ++c) mx[c] = NEG_MASK_F;
    int pend_r = -1;   // a document whose four waves left their column maxima in xch (previous iteration)

    // rounding, certification, sums and outputs of a finished document from its fp32 column maximum `am` (lane < 32: column lane)
    auto finish_doc = [&](int rr, float am) {
      const int q = a.ch_begin * 32 + (lane & 31);
      const bool mine = (lane < 32) && (q < a.Q);
      const half_t hm = (half_t)am;
      float sv = mine ? (float)hm : 0.f;
      uint32_t ef = (__float_as_uint(am) >> 23) & 0xFFu;
      ef = ef < 113u ? 113u : ef;
      const float halfulp = __uint_as_float((ef - 11u) << 23);
      const float dist = halfulp - __builtin_fabsf(am - (float)hm);
      const bool flag = mine && !(dist > qn[lane & 31]);
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
      if (a.cm16 && lane < 32 && a.ch_begin * 32 < a.Qp)
        a.cm16[((int64_t)b * a.Rcap + rr) * a.Qp + a.ch_begin * 32 + lane] = __builtin_bit_cast(uint16_t, hm);
      if (a.flags && lane == 0 && a.ch_begin < nflag) a.flags[((int64_t)b * a.Rcap + rr) * nflag + a.ch_begin] = (uint32_t)bal;
      if (lane == 0) {
        outp[rr] = a.accumulate ? (outp[rr] + sv) : sv;
        if (a.unc) {
          float* up = a.unc + (int64_t)b * a.Rcap + rr;
          *up = a.accumulate ? (*up + fu) : fu;
          float* um = a.uncm + (int64_t)b * a.Rcap + rr;
          *um = a.accumulate ? (*um + fm) : fm;
        }
      }
    };

    auto compute = [&](Ms7Stage& S, const Ms7Grp& d, uint32_t buf, int par) {
      const int ntok = d.valid ? (d.len - d.g0 < 64 ? d.len - d.g0 : 64) : 0;
      if (16 * wv < ntok) {   // (wave-uniform) this wave has tokens in the group
        uint32_t e[NE];
        const int slot = S.slot < MS7_ROWS ? S.slot : MS7_ROWS - 1;
        typedef __attribute__((address_space(3))) const ms_u32x4 ms7_lds_row;
#pragma unroll
        for (int s = 0; s < KS4; ++s) {
          const uint32_t addr = buf + (uint32_t)slot * 256u + (uint32_t)((((4 * s + g) ^ (slot & 15))) * 16);
          const ms_u32x4 v = *(ms7_lds_row*)(uintptr_t)addr;
          e[4 * s] = v.x; e[4 * s + 1] = v.y; e[4 * s + 2] = v.z; e[4 * s + 3] = v.w;
        }
        if (S.nrows > MS7_ROWS) {   // (wave-uniform, rare) rows beyond the buffer: the lane gathers its own
          const uint16_t* cp = a.cent + (long long)S.own * D + 8 * g;
          if (S.slot >= MS7_ROWS) {
#pragma unroll
            for (int s = 0; s < KS4; ++s) {
              const uint4 v = *reinterpret_cast<const uint4*>(cp + 32 * s);
              e[4 * s] = v.x; e[4 * s + 1] = v.y; e[4 * s + 2] = v.z; e[4 * s + 3] = v.w;
            }
          }
        }
        ms6_decode<NBITS, RW, NE>(laneoff, S.rw, e);
        if (a.rinv && !__any((int)(S.nrm >> 31))) {
          const float r = __uint_as_float(S.nrm);
#pragma unroll
          for (int i = 0; i < NE; i += 2) norm_mul2(e[i], e[i + 1], r);
        } else {
          uint16_t n16 = (uint16_t)S.nrm;
          if (a.rinv) {
            int tok = d.g0 + 16 * wv + r16;
            tok = tok < d.len ? tok : d.len - 1;
            n16 = a.norms[d.off + tok];
          }
          float r_hi, r_lo;
          recip2((float)__builtin_bit_cast(half_t, n16), r_hi, r_lo);
#pragma unroll
          for (int i = 0; i < NE; i += 2) norm_pair2(e[i], e[i + 1], r_hi, r_lo);
        }
        f4v acc[NC16];
#pragma unroll
        for (int c = 0; c < NC16; ++c) {
#pragma unroll
          for (int s = 0; s < KS4; ++s) {
            const h8 av = __builtin_bit_cast(h8, make_uint4(e[4 * s], e[4 * s + 1], e[4 * s + 2], e[4 * s + 3]));
            const h8 bq = __builtin_bit_cast(h8, qs[(c * KS4 + s) * 64 + lane]);
            if (s == 0) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bq, f4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bq, acc[c], 0, 0, 0);
          }
        }
        const int t0 = d.g0 + 16 * wv;
        const bool partial = t0 + 16 > d.len;
#pragma unroll
        for (int c = 0; c < NC16; ++c) {
          float v0 = acc[c][0], v1 = acc[c][1], v2 = acc[c][2], v3 = acc[c][3];
          if (partial) {
            const int rowb = t0 + 4 * g;
            if (rowb + 0 >= d.len) v0 = NEG_MASK_F;
            if (rowb + 1 >= d.len) v1 = NEG_MASK_F;
            if (rowb + 2 >= d.len) v2 = NEG_MASK_F;
            if (rowb + 3 >= d.len) v3 = NEG_MASK_F;
          }
          mx[c] = max3_raw(v2, v3, max3_raw(v0, v1, mx[c]));
        }
      }
      if (d.valid && d.g0 + 64 >= d.len) {   // the document's last group: this wave's column maxima -> xch (read behind the next barrier)
        const unsigned long long upper16 = 0xFFFF0000FFFF0000ull;
#pragma unroll
        for (int c = 0; c < NC16; ++c) {
          auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx[c]), __float_as_uint(mx[c]), false, false);
          float m = __builtin_fmaxf(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
          auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
          mx[c] = __builtin_fmaxf(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
        }
        float am;
        asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(am) : "v"(mx[0]), "v"(mx[1]), "s"(upper16));
        if (lane < 32) xch[((par * MS7_SG + sg) * 4 + wv) * 32 + lane] = am;
        pend_r = d.r;
#pragma unroll
        for (int c = 0; c < NC16; ++c) mx[c] = NEG_MASK_F;
      }
    };

    // ---- pipeline.  vmcnt retires in order, so what a wave waits for at the top of iteration k is "everything issued two
    // iterations ago or earlier": the rows of group k (DMA issued at k-2), the residuals / reciprocal of group k (issued at k-4)
    // and the codes of group k+2 (issued at k-3), while the 8 loads of the last two iterations -- rows of k+1, residuals of k+2
    // and k+3, codes of k+3 and k+4 -- stay in flight: an iteration has to cover HALF a memory round trip, not a whole one
    // (with the rows one group ahead and the waits draining the youngest prefetch, the first version ran 351 us).
    // Iteration k: wait, meet, [close the document finished at k-1], rows of k+2 -> buffer (k+2) % 3, compute k, prefetch
    // residuals / reciprocal of k+4 (into the stage group k just left) and codes of k+5.
    Ms7Stage S0, S1, S2, S3;
    Ms7Grp d0 = next_grp(), d1 = next_grp(), d2 = next_grp(), d3 = next_grp(), d4 = next_grp(), d5{};
    issue_loads(S0, d0);
    issue_loads(S1, d1);
    issue_loads(S2, d2);
    issue_loads(S3, d3);
    // (codes of group j live in C[j % 4]: rings of registers, not moves -- moving a register whose load is in flight waits for it)
    int32_t C0 = load_codes(d0), C1 = load_codes(d1), C2 = load_codes(d2), C3 = load_codes(d3);
    const uint32_t rows0 = (uint32_t)(MS7_LDS_ROWS + sg * MS7_NBUF * MS7_ROWS * 256);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    stage_rows(S0, d0, rows0, C0);
    stage_rows(S1, d1, rows0 + (uint32_t)(MS7_ROWS * 256), C1);
    C0 = load_codes(d4);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // (once per query: the steady-state count below assumes two full iterations behind it)
    int k = 0;
    // A sub-group's four waves meet at an LDS counter, not at s_barrier: the hardware barrier would walk all SIXTEEN waves of
    // the workgroup in lock step -- every sub-group waiting for the slowest row fetch of any, and all waves in the same phase
    // (LDS-bound decode, MFMA) at the same time: 358 us against k_maxsim6's 223.  With the counter the four sub-groups of a CU
    // drift apart like independent workgroups while sharing the decode table.
    auto sg_barrier = [&](uint32_t target) {
      __attribute__((address_space(3))) uint32_t* bp = (__attribute__((address_space(3))) uint32_t*)(uintptr_t)(MS7_LDS_CTL + 4 * sg);
      asm volatile("" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(bp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      for (;;) {
        const uint32_t v = __hip_atomic_load(bp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)v) >= target) break;
        __builtin_amdgcn_s_sleep(1);
      }
      asm volatile("" ::: "memory");
    };
    int kb = 0;   // k % 3
    auto body = [&](Ms7Stage& cur, Ms7Stage& nx2, const int32_t& c_use, int32_t& c_iss) -> bool {
      __builtin_amdgcn_s_waitcnt(0x0F78);   // vmcnt(8): all but the last two iterations' loads
      sg_barrier(4u * (uint32_t)(k + 1));   // the other three waves' parts of group k's rows have landed too; everybody is through with group k-1
      if (pend_r >= 0) {   // (sub-group-uniform) last iteration finished a document: one of its waves closes it
        if (wv == (pend_r & 3)) {
          const int par = (k + 1) & 1;   // (= the previous iteration's parity)
          float am = NEG_MASK_F;
          if (lane < 32) {
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) am = __builtin_fmaxf(am, xch[((par * MS7_SG + sg) * 4 + w2) * 32 + lane]);
          }
          finish_doc(pend_r, am);
        }
        pend_r = -1;
      }
      if (!d0.valid) return false;   // (sub-group-uniform) the stream is exhausted and nothing is pending
      const int kb2 = kb == 0 ? 2 : kb - 1;   // (k + 2) % 3
      stage_rows(nx2, d2, rows0 + (uint32_t)(kb2 * MS7_ROWS * 256), c_use);
      compute(cur, d0, rows0 + (uint32_t)(kb * MS7_ROWS * 256), k & 1);
      d5 = next_grp();
      issue_loads(cur, d4);        // group k + 4 takes the stage group k just left
      c_iss = load_codes(d5);      // group k + 5 takes the register of group k + 1 (consumed at k - 1)
      d0 = d1; d1 = d2; d2 = d3; d3 = d4; d4 = d5;
      ++k;
      kb = kb == 2 ? 0 : kb + 1;
      return true;
    };
    while (true) {
      if (!body(S0, S2, C2, C1)) break;
      if (!body(S1, S3, C3, C2)) break;
      if (!body(S2, S0, C0, C3)) break;
      if (!body(S3, S1, C1, C0)) break;
    }
  }
}

template <int NBITS>
static void launch_maxsim7(const FpIndexDev& ix, const uint16_t* qpad, const FpSearchShape& sh, const int32_t* sel_pid, const int64_t* pref,
                           int64_t Rcap, float* exact, const FpMaxsimAux& aux, hipStream_t st) {
  const int64_t tot_max = (int64_t)sh.B * Rcap;
  int grid = ms_num_cus();
  if ((int64_t)grid * MS7_WAVES > tot_max) grid = (int)std::max<int64_t>(1, (tot_max + MS7_WAVES - 1) / MS7_WAVES);
  MsArgs a{ix.centroids, ix.lut, ix.codes, ix.norms, ix.residuals, ix.doc_off, qpad, sel_pid, pref, exact, aux.cm16, aux.unc, aux.uncm, aux.flags,
           Rcap, sh.B, sh.Q, sh.Qp, 0, 0, 1.9073486e-06f /*2^-19*/, ix.rinv, 0};
  static std::atomic<uint64_t> ok7{0};
  fp_allow_big_lds((const void*)k_maxsim7<NBITS>, ok7, 160 * 1024);
  hipLaunchKernelGGL((k_maxsim7<NBITS>), dim3((unsigned)grid), dim3(MS7_WAVES * 64), (size_t)MS7_LDS_BYTES, st, a);
}


static int fpk_maxsim_lab(const FpIndexDev& ix, const uint16_t* qpad, const FpSearchShape& sh, const int32_t* sel_pid, const int32_t* sel_cnt,
                          int64_t Rcap, float* exact, int64_t* pref, const FpMaxsimAux& aux, hipStream_t st, bool pref_ready) {
  // dim 128 x nbits 4 with one 32-column query chunk (everything else: the product's kernels)
  if (!(ix.resid_native && ix.dim == 128 && ix.nbits == 4 && sh.Qp == 32 && ix.C < (1ll << 24))) return -1;
  if (!pref_ready) hipLaunchKernelGGL(k_cnt_prefix, dim3(1), dim3(256), 0, st, sel_cnt, sh.B, pref);
  launch_maxsim7<4>(ix, qpad, sh, sel_pid, pref, Rcap, exact, aux, st);
  return 0;
}
