// Host engine + C ABI (include/fastplaid.h) of the MI355X PLAID search library.
// Replaces, for the search path, rust/search/{search,load,tensor,padding}.rs and
// rust/utils/residual_codec.rs of the reference: index construction = upload + re-layout,
// search_many = one batched, stream-ordered pipeline of HIP kernels (fp_kernels.hip).
// No torch / tch; only the HIP runtime.
#include <atomic>
#include "../../include/fastplaid.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "fp_internal.h"

// ------------------------------------------------------------------------------------------
// errors (rust/utils/errors.rs:5-7: every failure becomes one message string)
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(x)                                                                                      \
  do {                                                                                                 \
    hipError_t e_ = (x);                                                                               \
    if (e_ != hipSuccess)                                                                              \
      return fail(FP_EHIP, std::string("HIP error: ") + hipGetErrorString(e_) + " at " #x);            \
  } while (0)

// a kernel launch reports a bad configuration only through hipGetLastError: every stage boundary checks it, so a failed launch
// stops the search with the stage named instead of surfacing later in somebody else's HIP call
#define STAGE_DONE(X)                                                                                  \
  do {                                                                                                 \
    hipError_t l_ = hipGetLastError();                                                                 \
    if (l_ != hipSuccess)                                                                              \
      return fail(FP_EHIP, std::string("HIP launch error: ") + hipGetErrorString(l_) + " before the end of stage " #X); \
    if (!s->capturing) HIPCHK(hipEventRecord(s->ev[X], st));                                           \
  } while (0)
#define LAUNCHCHK(where)                                                                               \
  do {                                                                                                 \
    hipError_t l_ = hipGetLastError();                                                                 \
    if (l_ != hipSuccess) return fail(FP_EHIP, std::string("HIP launch error: ") + hipGetErrorString(l_) + " in " where); \
  } while (0)

extern "C" const char* fp_last_error(void) { return g_err.c_str(); }
extern "C" const char* fp_version(void) { return "fastplaid-hip 0.1 (gfx950)"; }
extern "C" int fp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ------------------------------------------------------------------------------------------
// grow-only device buffer
// ------------------------------------------------------------------------------------------
// A captured graph (FP_GRAPH) holds raw pointers and is only replayed while nothing it points to has moved: every (re)allocation
// of a buffer bumps the allocation generation of the scratch that OWNS the buffer, and the generation is part of the graph's key.
// The owner is fixed when the buffer is constructed (a buffer that is a member of a Scratch picks up that scratch's counter
// through t_ctor_gen, see ScratchGen), not looked up through the calling thread: the staged shard API holds a scratch across
// calls, possibly on several threads and interleaved with other searches.
static std::atomic<uint64_t> g_alloc_gen{0};   // buffers outside any scratch
static thread_local uint64_t* t_ctor_gen = nullptr;
static std::atomic<uint64_t> g_graph_replays{0};   // fp_graph_replay_count(): calls served by one hipGraphLaunch
// fp_set_graph_replay / FP_GRAPH: on unless the environment says 0
static std::atomic<int> g_graph_replay{[] { const char* e = getenv("FP_GRAPH"); return (e && atoi(e) == 0) ? 0 : 1; }()};
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  uint64_t* gen = t_ctor_gen;   // the owning scratch's allocation generation (nullptr: the process-wide one)
  void bump() {
    if (gen) ++*gen;
    else g_alloc_gen.fetch_add(1, std::memory_order_relaxed);
  }
  hipError_t ensure(size_t need) {
    if (need <= cap) return hipSuccess;
    bump();
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = need + need / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) return e;
    cap = want;
    return hipSuccess;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

// grow-only pinned host buffer (results come down in ONE async copy instead of three staged ones)
// Every entry point selects its index's device through this.  First call on a thread: the thread's stream-capture interaction mode
// becomes "relaxed".  fp_search captures its batch pipeline into a HIP graph on its own non-blocking stream; with the default
// (global) mode the runtime refuses potentially unsafe calls -- allocations, synchronous copies -- on EVERY thread while any
// capture is open, and one refused call invalidates the capture: two threads searching one index failed each other's calls
// (found by tests/fuzz_worker.py threads, round 6).  Relaxed means what the library needs: its captures concern the capturing
// stream only.
static hipError_t fp_set_device(int device) {
  static thread_local bool relaxed = false;
  if (!relaxed) {
    hipStreamCaptureMode m = hipStreamCaptureModeRelaxed;
    (void)hipThreadExchangeStreamCaptureMode(&m);
    relaxed = true;
  }
  return hipSetDevice(device);
}

// The library never touches the legacy (null) stream (one exception: prepay_legacy_stream): a synchronous hipMemcpy there is refused -- and invalidates the capture --
// while another thread of the process captures a graph (fp_search does, on its scratch's stream).  Blocking copies go through
// an explicit stream instead: the caller's own (index construction, a scratch) or this per-device utility stream.
static hipStream_t util_stream(int device) {
  static std::mutex mu;
  static hipStream_t streams[64] = {};
  std::lock_guard<std::mutex> g(mu);
  hipStream_t& st = streams[device & 63];
  if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;   // (nullptr: the legacy stream after all)
  return st;
}
// captures this library has open right now, over all threads (see prepay_legacy_stream)
static std::atomic<int> g_open_captures{0};
// The ONE deliberate legacy-stream call of the library, at the end of an index construction: the runtime creates the device's legacy
// stream (a hardware queue: 3 - 4 ms) the first time something needs it, and the first hipGraphInstantiate of a process does -- ~9 ms
// inside somebody's third fp_search call (profiles/r06_first_graph_cost.txt).  Index construction is where a few milliseconds do not
// matter.  Skipped while one of the library's own captures is open (the runtime would refuse the call and invalidate that capture).
static void prepay_legacy_stream() {
  if (g_open_captures.load(std::memory_order_acquire) != 0) return;
  void* p = nullptr;
  if (hipMalloc(&p, 64) == hipSuccess) {
    (void)hipMemset(p, 0, 64);
    (void)hipFree(p);
  }
  (void)hipGetLastError();
}
static hipError_t copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
  return e != hipSuccess ? e : hipStreamSynchronize(st);
}

struct HostBuf {
  void* p = nullptr;
  size_t cap = 0;
  uint64_t* gen = t_ctor_gen;
  hipError_t ensure(size_t need) {
    if (need <= cap) return hipSuccess;
    if (gen) ++*gen;
    else g_alloc_gen.fetch_add(1, std::memory_order_relaxed);
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    size_t want = need + need / 8 + 256;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) return e;
    cap = want;
    return hipSuccess;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

enum { ST_UPLOAD = 0, ST_CENTROID, ST_S1MAIN, ST_PROBE, ST_IVF, ST_COMPACT, ST_APREP, ST_APPROX, ST_REFINE, ST_SELECT, ST_MAXSIM, ST_REPAIR, ST_TOPK, ST_N };
// "S1 centroid_gemm" is exactly ONE kernel, the main pass of the centroid GEMM (round 5; until then the stage included the
// sampled pre-pass and the floors, now "S1 prepass+floors": zero when level 0 is not prepared)
static const char* kStageNames[ST_N] = {"upload+pack", "S1 prepass+floors", "S1 centroid_gemm", "S2 probe_topk", "S3 ivf_mark+count",
                                        "S3 compact",  "S4 prepare",       "S4 approx",     "S4 refine",
                                        "S5 select",   "S6+S7 maxsim",     "S7 order repair", "S8 topk+download"};
// "S4 approx" is exactly ONE kernel -- k_l0_scan (level-0 bound of every candidate), k_approx_q8 (8-bit bounds of every
// candidate) or k_approx (exact score of every candidate), whichever form of S4 runs -- so that it can be compared with a
// profiler's per-kernel time; "S4 prepare" = level 0's floors + excess table (zero otherwise); "S4 refine" = the cut, the
// survivor compaction and their exact rescoring (zero when every candidate is scored exactly).

// base of Scratch: constructed before the buffer members, so that each of them records &alloc_gen as its owner
struct ScratchGen {
  uint64_t alloc_gen = 0;   // bumped by every (re)allocation of one of this scratch's buffers
  ScratchGen() { t_ctor_gen = &alloc_gen; }
};
struct Scratch : ScratchGen {
  Scratch() { t_ctor_gen = nullptr; }   // (runs after the members' initialisers)
  hipStream_t st = nullptr;
  hipEvent_t ev[ST_N + 1] = {};
  DevBuf qin, qpad, S, partial, cells, ucells, ncells, allow, subbm, invalid, sub_ids, sub_off, bitmap, blkcnt, ncand,
      cand_off, cand_pid, approx, hist, selstate, sel_pid, sel_approx, sel_cnt, tie_pid, exact, out_pid, out_score, out_cnt, tmpf, tmpp,
      tok_idx, recon, out_all, S8, cmax128, kq, q8hist, cut, blkcnt2, nsurv, surv_off, surv_pid, l0_floors, l0_F, l0_e8, l0_esc, l0_ub,
      l0_hist, l0_npilot, l0_pilot_pid, l0_pilot_approx, l0_pilot_idx, l0_capprox, l0_thr, l0_nextra, l0_xpid, l0_xdst, l0_blkx, Ssample, l0_gfl, u_cnt, spec_total, sh_lmarks, sh_lnmark, sh_x, sh_xall, ms_uncm, ms_cm16, ms_unc, ms_flags, ms_pref, ms_marks, ms_nmark, sh_rec, sh_all, tickets, ms_flat, l0_ubp, wcol, s1stats, qpad_s1, lz_state, lz_gpid, lz_gval, lz_slackq;
  HostBuf h_out, h_small;
  int lazy_fails = 0;           // batches whose lazy-S1 selection lists overflowed (run again eagerly); two of them switch the lazy form off for this scratch (post_batch: for a stretch of batches, not for good)
  int lazy_clean = 0, lazy_off_batches = 0;
  bool ms_repairable = false;   // the last run_maxsim produced budgets / flags (fast-path shape, repair enabled)
  bool ms_have_marks = false;   // h_small + 64 holds the per-query marked counts of the last batch
  // Candidate-buffer capacity learnt from earlier batches of the same shape (0: none): with it, fp_search does not wait for the
  // candidate total in the middle of the pipeline (see run_front)
  int64_t spec_cap = 0;
  int64_t spec_last = 0;   // the previous batch's total: sizes the grids (the capacity would over-provision them by a quarter)
  int64_t spec_key[4] = {0, 0, 0, 0};   // {B, Q, n_ivf_probe, subset?}
  // Graph replay (fp_set_graph_replay; FP_GRAPH=0 switches it off): once fp_search runs on the learnt capacity there is no host decision left inside the pipeline, so the
  // whole call -- query upload from a pinned staging buffer, ~55 launches and fills, result download -- is captured once per
  // {shape, parameters, capacity, allocation generation} and replayed with one hipGraphLaunch
  struct GraphCache {
    hipGraphExec_t exec = nullptr;
    int64_t key[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t last[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // key of the last speculative call (a capture needs one such call before it: every buffer sized)
    int warm = 0, fails = 0;
    bool valid = false, used_q8 = false, marked = false;
    bool no_fb = false;   // the captured batch ran the threshold probe WITHOUT its fallback kernels: a raised overflow flag voids a replayed batch too
    bool fb_thr = false;  // ... WITH them (Pipe::probe_fb_thr)
  } graph;
  bool capturing = false;
  // The learnt capacity and the captured graph belong to a SHAPE {B, Q, n_ivf_probe, n_full_scores, top_k, subset?}.  The fields
  // above are the state of the shape in use; the states of the other recently used shapes wait here (least recently used one
  // evicted), so that a caller alternating between a few shapes keeps replaying each of them.
  struct ShapeState {
    int64_t key[6] = {-1, -1, -1, -1, -1, -1};
    int64_t spec_cap = 0, spec_last = 0;
    GraphCache graph;
    uint64_t stamp = 0;
    bool used = false;
  };
  static constexpr int kShapes = 8;
  ShapeState shapes[kShapes];
  int cur_shape = -1;
  uint64_t shape_stamp = 0;
  void select_shape(const int64_t (&key)[6]) {
    if (cur_shape >= 0 && std::equal(key, key + 6, shapes[cur_shape].key)) { shapes[cur_shape].stamp = ++shape_stamp; return; }
    if (cur_shape >= 0) {   // park the live state
      ShapeState& o = shapes[cur_shape];
      o.spec_cap = spec_cap; o.spec_last = spec_last; o.graph = graph;
    }
    int slot = -1;
    for (int i = 0; i < kShapes; ++i)
      if (shapes[i].used && std::equal(key, key + 6, shapes[i].key)) slot = i;
    if (slot < 0) {
      for (int i = 0; i < kShapes && slot < 0; ++i)
        if (!shapes[i].used) slot = i;
      if (slot < 0) {   // evict the least recently used shape
        slot = 0;
        for (int i = 1; i < kShapes; ++i)
          if (shapes[i].stamp < shapes[slot].stamp) slot = i;
        if (shapes[slot].graph.exec) (void)hipGraphExecDestroy(shapes[slot].graph.exec);
      }
      shapes[slot] = ShapeState{};
      std::copy(key, key + 6, shapes[slot].key);
      shapes[slot].used = true;
    }
    ShapeState& n = shapes[slot];
    n.stamp = ++shape_stamp;
    spec_cap = n.spec_cap; spec_last = n.spec_last; graph = n.graph;
    n.graph.exec = nullptr;   // (owned by the live fields while the shape is in use)
    spec_key[0] = key[0]; spec_key[1] = key[1]; spec_key[2] = key[2]; spec_key[3] = key[5];
    cur_shape = slot;
  }
  HostBuf h_qin;
  bool fold_stats = false;      // host-buffer search: survivor / marked counts travel with the result copy instead of two small copies
  bool ms_marked_now = false;   // the last run_maxsim marked near-tied documents (ms_nmark is valid)
  bool pref_ready = false;      // ms_pref holds the prefix of sel_cnt (left by S5's last kernel)
  bool probe_fb = false;        // a batch overflowed the threshold probe's tie room: the fallback kernels are enqueued (post_batch: until 64 batches in a row raise no flag)
  int probe_fb_clean = 0;
  int l0_poor = 0;              // consecutive batches in which level 0 let more than a quarter of the candidates through
  int l0h_poor = 0;             // ... and its hot-code form
  int sh_marks_mode = 0;        // sharded search, third exchange: 0 none, 1 near-tied documents, 2 every flagged document
  void destroy() {
    out_all.release();
    for (DevBuf* b : {&S8, &cmax128, &kq, &q8hist, &cut, &blkcnt2, &nsurv, &surv_off, &surv_pid, &l0_floors, &l0_F, &l0_e8, &l0_esc, &l0_ub,
                      &l0_hist, &l0_npilot, &l0_pilot_pid, &l0_pilot_approx, &l0_pilot_idx, &l0_capprox, &l0_thr, &l0_nextra, &l0_xpid, &l0_xdst, &l0_blkx, &Ssample, &l0_gfl, &u_cnt, &spec_total, &sh_lmarks, &sh_lnmark, &sh_x, &sh_xall, &ms_uncm, &ms_cm16, &ms_unc, &ms_flags, &ms_pref, &ms_marks,
                      &ms_nmark, &sh_rec, &sh_all, &tickets, &ms_flat, &l0_ubp, &wcol, &s1stats, &qpad_s1, &lz_state, &lz_gpid, &lz_gval, &lz_slackq})
      b->release();
    h_out.release();
    h_small.release();
    h_qin.release();
    if (graph.exec) { (void)hipGraphExecDestroy(graph.exec); graph.exec = nullptr; graph.valid = false; }
    for (auto& sh2 : shapes)
      if (sh2.graph.exec) { (void)hipGraphExecDestroy(sh2.graph.exec); sh2.graph.exec = nullptr; }
    DevBuf* all[] = {&qin,      &qpad,     &S,        &partial, &cells,   &ucells,     &ncells,  &allow,    &subbm,  &invalid, &sub_ids,
                     &sub_off,  &bitmap,   &blkcnt,   &ncand,   &cand_off, &cand_pid,  &approx,  &hist,     &selstate, &sel_pid, &tie_pid,
                     &sel_approx, &sel_cnt, &exact,   &out_pid, &out_score, &out_cnt,  &tmpf,    &tmpp,     &tok_idx, &recon};
    for (DevBuf* b : all) b->release();
    for (auto& e : ev)
      if (e) (void)hipEventDestroy(e);
    if (st) (void)hipStreamDestroy(st);
  }
};

struct fp_index {
  FpIndexDev d{};
  int device = 0;
  bool has_ivf = false;
  std::vector<void*> owned;
  std::vector<int64_t> h_doc_off;
  int64_t bytes = 0;
  std::mutex mu;
  std::vector<Scratch*> pool;
  bool synthetic = false;
  int64_t n_hard_tokens = 0;   // tokens without a one-multiply reciprocal (k_token_rinv)
  float cent_norm_max = 1.f;   // largest centroid norm (>= 1): scales S1's certification window (FpS1Exact)
  bool tickets_ok = true;      // the fence-free ticket chains passed their self-test on this index's device (finish_layout)
};

static thread_local float g_last_ms[ST_N];
static thread_local bool g_have_ms = false;
static thread_local int g_last_lazy = 0;      // the last fp_search batch ran S1's lazy form (1), the eager one (0), or was a replayed graph (-1)
static thread_local uint64_t g_last_s1[4];   // FP_S1_STATS: S1's certification counters of the last call (flagged, changed, slow path, mode-2 unflagged differences)
static bool s1_stats_enabled() { static const bool on = getenv("FP_S1_STATS") != nullptr; return on; }
static thread_local int64_t g_last_counts[6];  // candidates, exact-scored docs, repaired docs, sub-batches, form of S4 (0 exact / 1 8-bit bounds / 2 level 0; -1: replayed graph), lazy-S1 overflows the scratch remembers (>= 2: lazy form off)

// results of one (sub-)batch in one device block: ids | scores | counts
struct OutLayout {
  size_t score_off, cnt_off, stat_off, total, nk, n;
  OutLayout(int B, int64_t K) {
    n = (size_t)B;
    nk = (size_t)B * (size_t)K;
    score_off = (nk * 8 + 255) & ~(size_t)255;
    cnt_off = (score_off + nk * 4 + 255) & ~(size_t)255;
    stat_off = (cnt_off + n * 4 + 255) & ~(size_t)255;   // [4 + B] int64 statistics (survivor total, marked documents per query, lazy-S1 overflow flag, candidate total, probe overflow flag)
    total = stat_off + (n + 4) * 8;
  }
  void scatter(const void* host, int64_t* pids, float* scores, int32_t* counts) const {
    const char* h = static_cast<const char*>(host);
    memcpy(pids, h, nk * 8);
    memcpy(scores, h + score_off, nk * 4);
    memcpy(counts, h + cnt_off, n * 4);
  }
};

static Scratch* new_scratch() {
  Scratch* s = new Scratch();
  if (hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking) != hipSuccess) {
    delete s;
    return nullptr;
  }
  for (auto& e : s->ev) (void)hipEventCreate(&e);
  return s;
}
static void fresh_stream(Scratch* s) {
  hipStream_t fresh = nullptr;
  if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) == hipSuccess) {
    (void)hipStreamDestroy(s->st);   // (nothing is pending on it: a capture runs nothing)
    s->st = fresh;
  }
  (void)hipGetLastError();
}
// Closes a capture that cannot be used (an error inside it, or the runtime invalidated it) and gives the scratch a FRESH stream: on
// this runtime a stream whose capture was invalidated from outside keeps returning "previous error during capture" after
// hipStreamEndCapture (found by tests/fuzz_worker.py hostile).  Nothing is pending on the old stream: a capture runs nothing.
static void abandon_capture(Scratch* s) {
  hipGraph_t g = nullptr;
  (void)hipStreamEndCapture(s->st, &g);
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
  if (s->capturing) g_open_captures.fetch_sub(1, std::memory_order_acq_rel);
  s->capturing = false;
  s->graph.fails = 1000;
  s->graph.warm = 0;
  fresh_stream(s);
}
static Scratch* acquire(fp_index* ix) {
  {
    std::lock_guard<std::mutex> g(ix->mu);
    if (!ix->pool.empty()) {
      Scratch* s = ix->pool.back();
      ix->pool.pop_back();
      return s;
    }
  }
  return new_scratch();
}
static void release(fp_index* ix, Scratch* s) {
  std::lock_guard<std::mutex> g(ix->mu);
  ix->pool.push_back(s);
}

// ------------------------------------------------------------------------------------------
// index construction (load.rs:124-186)
// ------------------------------------------------------------------------------------------
static void build_lut_host(int nbits, const uint16_t* weights, std::vector<uint16_t>& lut) {
  // residual_codec.rs:83-140 folded: byte -> rev_map -> idx_lookup[rev][j] -> weights
  const int per = 8 / nbits, mask = (1 << nbits) - 1;
  lut.assign((size_t)256 * per, 0);
  for (int v = 0; v < 256; ++v) {
    unsigned out = 0;
    int pos = 8;
    while (pos >= nbits) {
      unsigned seg = ((unsigned)v >> (pos - nbits)) & (unsigned)mask, rev = 0;
      for (int k = 0; k < nbits; ++k)
        if (seg & (1u << k)) rev |= 1u << (nbits - 1 - k);
      out |= rev;
      if (pos > nbits) out <<= nbits;
      pos -= nbits;
    }
    out &= 0xFF;
    int j = 0;
    for (int k = per - 1; k >= 0; --k) lut[(size_t)v * per + j++] = weights[(out >> (k * nbits)) & mask];
  }
}

// largest row norm of an fp16 matrix (never below 1): the rounding noise of a score's fp32 chain scales with |c| |q|
static float max_row_norm_f16(const uint16_t* m, int64_t rows, int dim) {
  double best = 1.0;
  for (int64_t r = 0; r < rows; ++r) {
    double ss = 0.0;
    const uint16_t* row = m + r * dim;
    for (int k = 0; k < dim; ++k) {
      const double x = (double)(float)__builtin_bit_cast(_Float16, row[k]);
      ss += x * x;
    }
    if (ss > best * best) best = std::sqrt(ss);
  }
  return (float)best;
}

template <typename T>
static hipError_t dev_alloc(fp_index* ix, T** p, size_t n) {
  void* q = nullptr;
  size_t bytes = (n ? n : 1) * sizeof(T) + 64;  // slack: vector loads may touch past the end
  hipError_t e = hipMalloc(&q, bytes);
  if (e != hipSuccess) return e;
  ix->owned.push_back(q);
  ix->bytes += (int64_t)bytes;
  *p = reinterpret_cast<T*>(q);
  return hipSuccess;
}

// limit > 0: every entry must lie in [0, limit) -- `what` names the array in the error (the reference fails on such an index
// inside index_select, per query; a code that is no centroid or a list entry that is no document would be an out-of-bounds
// read on the device here, so the index is refused when it is built)
static int upload_narrow(const int64_t* host, int32_t* dev, int64_t n, hipStream_t st, int64_t limit = 0, const char* what = "") {
  // chunked i64 -> i32 through a device staging buffer
  if (n <= 0) return FP_OK;
  const int64_t chunk = 32ll << 20;  // 32M entries = 256 MiB
  int64_t* stage = nullptr;
  const size_t stage_bytes = (size_t)std::min(chunk, n) * 8;
  HIPCHK(hipMalloc((void**)&stage, stage_bytes + 8));
  uint32_t* bad = limit > 0 ? reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(stage) + stage_bytes) : nullptr;
  uint32_t h_bad = 0;
  hipError_t e = bad ? hipMemsetAsync(bad, 0, 4, st) : hipSuccess;
  for (int64_t s = 0; s < n && e == hipSuccess; s += chunk) {
    int64_t m = std::min(chunk, n - s);
    e = hipMemcpyAsync(stage, host + s, (size_t)m * 8, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) break;
    fpk_narrow_i64_i32(stage, dev + s, m, 0, st, limit, bad);
    e = hipStreamSynchronize(st);
  }
  if (e == hipSuccess && bad) e = copy_sync(&h_bad, bad, 4, hipMemcpyDeviceToHost, st);
  (void)hipFree(stage);
  if (e != hipSuccess) return fail(FP_EHIP, hipGetErrorString(e));
  if (h_bad) return fail(FP_EINVAL, std::string(what) + ": " + std::to_string(h_bad) + " entries outside [0, " + std::to_string(limit) + ")");
  return FP_OK;
}

// residual_codec.rs:83-140 / search.rs:53-107 are generic in nbits (a divisor of 8) and dim; the centroid GEMM stages K in
// 8-dim (16-byte) pieces, hence dim % 8.  MFMA MaxSim kernels exist for dim 48/64/96/128 x nbits 2/4; every other shape runs
// the generic (exact, slower) MaxSim kernel.
static int check_shape(int nbits, int dim) {
  if (!(nbits == 1 || nbits == 2 || nbits == 4 || nbits == 8)) return fail(FP_EINVAL, "nbits must divide 8 (1, 2, 4 or 8)");
  if (dim < 8 || dim % 8 != 0 || dim > 1024) return fail(FP_EUNSUPPORTED, "dim must be a multiple of 8 in [8, 1024]");
  return FP_OK;
}

struct StreamGuard {   // destroys the construction stream on every exit path
  hipStream_t st = nullptr;
  ~StreamGuard() { if (st) (void)hipStreamDestroy(st); }
};

// common tail of index construction: tokens sorted by code inside each document, per-token norms, per-document unique codes.
// ix->d.{centroids, lut, codes, residuals, doc_off} are set.  Every device buffer is registered in ix->owned before its
// producer's return code is looked at, so a failing build frees it with the index.
// the self-test of the "last workgroup finishes the job" chains, once per device and process (fpk_ticket_selftest)
static bool device_tickets_ok(int device, hipStream_t st) {
  static std::mutex mu;
  static int state[64];   // 0 unknown, 1 ok, 2 failed
  std::lock_guard<std::mutex> g(mu);
  int& s = state[device & 63];
  if (s == 0) s = fpk_ticket_selftest(st) == 0 ? 1 : 2;
  return s == 1;
}

static int finish_layout(fp_index* ix, int maxlen, hipStream_t st) {
  FpIndexDev& D = ix->d;
  ix->tickets_ok = device_tickets_ok(ix->device, st);
  {
    uint16_t* perm = nullptr;
    const int src = fps_sort_docs(const_cast<int32_t*>(D.codes), const_cast<uint8_t*>(D.residuals), D.doc_off, D.N, D.T, maxlen, D.pr, &perm, st);
    if (perm) { ix->owned.push_back(perm); ix->bytes += D.T * 2; }
    if (src != 0) return fail(FP_EHIP, "in-document token sort failed (hip error " + std::to_string(src) + ")");
    D.perm = perm;
  }
  {
    uint16_t* norms = nullptr;
    hipError_t e = hipMalloc((void**)&norms, (size_t)std::max<int64_t>(D.T, 1) * 2 + 64);
    if (e != hipSuccess) return fail(FP_EHIP, std::string("HIP error: ") + hipGetErrorString(e) + " (token norms)");
    ix->owned.push_back(norms);
    ix->bytes += D.T * 2;
    fpk_token_norms(D, norms, st);
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(FP_EHIP, std::string("HIP error: ") + hipGetErrorString(e) + " (token norms)");
    D.norms = norms;
  }
  // S1's zero-padded view of the centroid table (FpIndexDev::cent_s1)
  D.cent_s1 = nullptr;
  D.dim_s1 = D.dim;
  const bool pad_env = true;
  if (pad_env && D.dim < 256 && D.dim != 64 && D.dim != 128) {
    const int dp = D.dim < 64 ? 64 : (D.dim < 128 ? 128 : 256);
    uint16_t* cp = nullptr;
    hipError_t e = hipMalloc((void**)&cp, (size_t)D.C * dp * 2 + 64);
    if (e != hipSuccess) return fail(FP_EHIP, std::string("HIP error: ") + hipGetErrorString(e) + " (padded centroid table)");
    ix->owned.push_back(cp);
    ix->bytes += D.C * dp * 2;
    e = hipMemsetAsync(cp, 0, (size_t)D.C * dp * 2, st);
    if (e == hipSuccess) e = hipMemcpy2DAsync(cp, (size_t)dp * 2, D.centroids, (size_t)D.dim * 2, (size_t)D.dim * 2, (size_t)D.C, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(FP_EHIP, std::string("HIP error: ") + hipGetErrorString(e) + " (padded centroid table)");
    D.cent_s1 = cp;
    D.dim_s1 = dp;
  }
  D.resid_native = 0;
  D.rinv = nullptr;
  if (fpk_maxsim6_shape(D.dim, D.nbits)) {
    // per-token reciprocals (from the rows still in the reference's byte order), then every row rewritten in the MaxSim kernel's
    // unit order IN PLACE, a slab of rows at a time through a bounded temporary
    static const bool use_rinv = fp_test_opt("ms_rinv", 1) != 0;
    hipError_t e = hipSuccess;
    if (use_rinv) {
      uint32_t* rv = nullptr;
      unsigned long long* nh = nullptr;
      e = hipMalloc((void**)&rv, (size_t)std::max<int64_t>(D.T, 1) * 4 + 64);
      if (e != hipSuccess) return fail(FP_EHIP, std::string("HIP error: ") + hipGetErrorString(e) + " (token reciprocals)");
      ix->owned.push_back(rv);
      ix->bytes += D.T * 4;
      e = hipMalloc((void**)&nh, 8);
      if (e != hipSuccess) return fail(FP_EHIP, std::string("HIP error: ") + hipGetErrorString(e) + " (token reciprocals)");
      ix->owned.push_back(nh);
      (void)hipMemsetAsync(nh, 0, 8, st);
      fpk_token_rinv(D, rv, nh, st);
      unsigned long long hard = 0;
      e = hipMemcpyAsync(&hard, nh, 8, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e != hipSuccess) return fail(FP_EHIP, std::string("HIP error: ") + hipGetErrorString(e) + " (token reciprocals)");
      D.rinv = rv;
      ix->n_hard_tokens = (int64_t)hard;
    }
    if (D.T > 0) {
      const int64_t slab = std::max<int64_t>(1, std::min<int64_t>(D.T, (512ll << 20) / D.pr));
      uint8_t* tmp = nullptr;
      e = hipMalloc((void**)&tmp, (size_t)slab * D.pr);
      if (e != hipSuccess) return fail(FP_EHIP, std::string("HIP error: ") + hipGetErrorString(e) + " (native residual order)");
      for (int64_t t0 = 0; t0 < D.T; t0 += slab) fpk_resid_native(D, t0, std::min<int64_t>(slab, D.T - t0), tmp, st);
      e = hipStreamSynchronize(st);
      (void)hipFree(tmp);
      if (e != hipSuccess) return fail(FP_EHIP, std::string("HIP error: ") + hipGetErrorString(e) + " (native residual order)");
    }
    D.resid_native = 1;
  }
  {
    int64_t* uoff = nullptr; int32_t* ucodes = nullptr; int64_t U = 0;
    hipError_t e = hipMalloc((void**)&uoff, ((size_t)D.N + 1) * 8 + 64);
    if (e != hipSuccess) return fail(FP_EHIP, std::string("HIP error: ") + hipGetErrorString(e) + " (unique-code offsets)");
    ix->owned.push_back(uoff);
    ix->bytes += (D.N + 1) * 8;
    const int urc = fps_build_ucodes(D.codes, D.doc_off, D.N, maxlen, &ucodes, &U, uoff, st);
    if (ucodes) { ix->owned.push_back(ucodes); ix->bytes += U * 4; }
    if (urc != 0) return fail(FP_EHIP, "unique-code build failed (hip error " + std::to_string(urc) + ")");
    D.ucodes = ucodes; D.uoff = uoff; D.U = U;
  }
  // packed code lines for S4's level 0: one set per range of 2^17 centroids (codes relative to the range), up to 2^20
  D.n_ranges = 0;
  D.n_lines = 0;
  D.l0_ppl = 8;
  // the static part of run_front's level-0 predicate (table of >= 32768 centroids, documents of <= 64 distinct codes on average;
  // FP_APPROX_IMPL=l0 forces level 0 in tests): an index that can never take level 0 does not carry the packed lines
  static const bool l0_forced = [] { const char* e2 = getenv("FP_APPROX_IMPL"); return e2 && e2[0] == 'l'; }();
  if (D.C <= 8 * (1ll << 17) && (l0_forced || (D.C * 64 >= (2ll << 20) && D.U <= 64 * D.N))) {
    const int nr = (int)((D.C + (1ll << 17) - 1) >> 17);
    // lines of 4 pieces (24 codes) when a document has few codes per range: half the lanes, half the bytes per candidate and range
    static const int ppl_env = (int)fp_test_opt("l0_ppl", 0);
    if (ppl_env == 4 || ppl_env == 8) D.l0_ppl = ppl_env;
    else if (nr > 1 && D.N > 0 && (double)D.U / ((double)D.N * nr) <= 14.0) D.l0_ppl = 4;
    // several ranges: ONE array of first lines, a document's ranges side by side (the scan then runs all ranges in one launch
    // and the lines of a document's ranges share fabric requests); the per-range arrays hold the extra lines only
    void* shared = nullptr;
    if (nr > 1) {
      const size_t sb = (size_t)D.N * nr * 16 * D.l0_ppl + 256;
      if (hipMalloc(&shared, sb) != hipSuccess) return fail(FP_EHIP, "packed-code build failed (out of memory)");
      ix->owned.push_back(shared);
      ix->bytes += (int64_t)sb;
      D.n_lines += D.N * nr;
    }
    for (int r = 0; r < nr; ++r) {
      void* lines = nullptr; int32_t* poff = nullptr; int64_t nl = 0;
      const int64_t lo = (int64_t)r << 17, hi = std::min<int64_t>(D.C, lo + (1ll << 17));
      const int prc = fps_build_pcodes(D.ucodes, D.uoff, D.N, lo, hi, &lines, &poff, &nl, st, D.l0_ppl, D.C, shared, nr, r);
      if (lines) { ix->owned.push_back(lines); ix->bytes += nl * 16 * D.l0_ppl; D.n_lines += nl; }
      if (poff) { ix->owned.push_back(poff); ix->bytes += (D.N + 1) * 8; }
      if (prc != 0) return fail(FP_EHIP, "packed-code build failed (hip error " + std::to_string(prc) + ")");
      D.pcodes_r[r] = static_cast<const uint4*>(lines); D.poff_r[r] = poff;
    }
    D.n_ranges = nr;
    D.pcodes = nr > 1 ? static_cast<const uint4*>(shared) : D.pcodes_r[0]; D.poff = D.poff_r[0];
  }
  return FP_OK;
}

extern "C" int fp_index_create(const fp_index_desc* d, int device_id, fp_index** out) {
  if (!d || !out) return fail(FP_EINVAL, "null argument");
  *out = nullptr;
  if (int rc = check_shape(d->nbits, d->dim)) return rc;
  if (d->n_centroids <= 0 || !d->centroids || !d->bucket_weights) return fail(FP_EINVAL, "centroids / bucket_weights missing");
  if (d->n_docs < 0 || (d->n_docs > 0 && (!d->doc_lengths || !d->doc_codes || !d->doc_residuals)))
    return fail(FP_EINVAL, "document arrays missing");
  if (d->n_docs >= 0x7FFFFFFFll) return fail(FP_EUNSUPPORTED, "more than 2^31-1 documents per index shard");
  HIPCHK(fp_set_device(device_id));
  fp_index* ix = new fp_index();
  ix->device = device_id;
  auto bail = [&](int rc) {
    fp_index_destroy(ix);
    return rc;
  };
  StreamGuard sg;
  hipStream_t& st = sg.st;
  FpIndexDev& D = ix->d;
  D.nbits = d->nbits; D.dim = d->dim; D.pr = d->dim * d->nbits / 8;
  D.C = d->n_centroids; D.N = d->n_docs; D.pid_offset = d->pid_offset;
  // doc offsets (tensor.rs:221-224)
  ix->h_doc_off.assign((size_t)D.N + 1, 0);
  int maxlen = 0;
  for (int64_t i = 0; i < D.N; ++i) {
    int64_t l = d->doc_lengths[i];
    if (l < 0) return bail(fail(FP_EINVAL, "negative document length"));
    ix->h_doc_off[i + 1] = ix->h_doc_off[i] + l;
    if (l > maxlen) maxlen = (int)l;
  }
  D.T = ix->h_doc_off[D.N];
  D.max_doc_len = maxlen;
  uint16_t* cent = nullptr; uint16_t* lut = nullptr; int64_t* doc_off = nullptr; int32_t* codes = nullptr; uint8_t* res = nullptr;
  int64_t* ivf_off = nullptr; int32_t* ivf_pids = nullptr;
#define ICHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return bail(fail(FP_EHIP, std::string("HIP error: ") + hipGetErrorString(e_) + " at " #x)); } while (0)
  ICHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  ICHK(dev_alloc(ix, &cent, (size_t)D.C * D.dim));
  ICHK(copy_sync(cent, d->centroids, (size_t)D.C * D.dim * 2, hipMemcpyHostToDevice, st));
  ix->cent_norm_max = max_row_norm_f16(d->centroids, D.C, D.dim);
  std::vector<uint16_t> hlut;
  build_lut_host(D.nbits, d->bucket_weights, hlut);
  ICHK(dev_alloc(ix, &lut, hlut.size()));
  ICHK(copy_sync(lut, hlut.data(), hlut.size() * 2, hipMemcpyHostToDevice, st));
  ICHK(dev_alloc(ix, &doc_off, (size_t)D.N + 1));
  ICHK(copy_sync(doc_off, ix->h_doc_off.data(), ((size_t)D.N + 1) * 8, hipMemcpyHostToDevice, st));
  ICHK(dev_alloc(ix, &codes, (size_t)D.T));
  if (int rc = upload_narrow(d->doc_codes, codes, D.T, st, D.C, "doc_codes (centroid ids)")) return bail(rc);
  ICHK(dev_alloc(ix, &res, (size_t)D.T * D.pr));
  if (D.T > 0) ICHK(copy_sync(res, d->doc_residuals, (size_t)D.T * D.pr, hipMemcpyHostToDevice, st));
  D.centroids = cent; D.lut = lut; D.doc_off = doc_off; D.codes = codes; D.residuals = res;
  if (int rc = finish_layout(ix, maxlen, st)) return bail(rc);
  ix->has_ivf = d->ivf != nullptr && d->ivf_lengths != nullptr;
  D.P = ix->has_ivf ? d->n_ivf_lists : 0;
  {
    std::vector<int64_t> hoff((size_t)D.P + 1, 0);
    for (int64_t i = 0; i < D.P; ++i) {
      if (d->ivf_lengths[i] < 0) return bail(fail(FP_EINVAL, "negative ivf length"));
      hoff[i + 1] = hoff[i] + d->ivf_lengths[i];
    }
    ICHK(dev_alloc(ix, &ivf_off, hoff.size()));
    ICHK(copy_sync(ivf_off, hoff.data(), hoff.size() * 8, hipMemcpyHostToDevice, st));
    const int64_t tot = hoff[D.P];
    ICHK(dev_alloc(ix, &ivf_pids, (size_t)tot));
    if (tot > 0) {
      if (int rc = upload_narrow(d->ivf, ivf_pids, tot, st, std::max<int64_t>(D.N, 1), "ivf (document ids)")) return bail(rc);
      // The reference gathers the probed lists and sorts + de-duplicates the ids per query (search.rs:538-541): any order inside a
      // list, repeated ids included, gives the same candidates there.  S3's per-tile range cut needs strictly ascending lists --
      // what create.rs / update.rs / delete.rs write (optimize_ivf).  Anything else is put in that order here, once.
      uint32_t* flag = nullptr;
      ICHK(hipMalloc((void**)&flag, 4));
      uint32_t h_flag = 0;
      hipError_t fe = hipMemsetAsync(flag, 0, 4, st);
      if (fe == hipSuccess) {
        fpk_ivf_check_sorted(ivf_off, ivf_pids, D.P, flag, st);
        fe = hipMemcpyAsync(&h_flag, flag, 4, hipMemcpyDeviceToHost, st);
      }
      if (fe == hipSuccess) fe = hipStreamSynchronize(st);
      (void)hipFree(flag);
      ICHK(fe);
      if (h_flag) {
        std::vector<int64_t> fixed;
        fixed.reserve((size_t)tot);
        std::vector<int64_t> noff((size_t)D.P + 1, 0);
        for (int64_t i = 0; i < D.P; ++i) {
          const size_t b = fixed.size();
          fixed.insert(fixed.end(), d->ivf + hoff[i], d->ivf + hoff[i + 1]);
          std::sort(fixed.begin() + b, fixed.end());
          fixed.erase(std::unique(fixed.begin() + b, fixed.end()), fixed.end());
          noff[i + 1] = (int64_t)fixed.size();
        }
        ICHK(copy_sync(ivf_off, noff.data(), noff.size() * 8, hipMemcpyHostToDevice, st));
        if (int rc = upload_narrow(fixed.data(), ivf_pids, (int64_t)fixed.size(), st)) return bail(rc);
      }
    }
  }
  ICHK(hipStreamSynchronize(st));
#undef ICHK
  D.ivf_off = ivf_off; D.ivf_pids = ivf_pids;
  prepay_legacy_stream();
  *out = ix;
  return FP_OK;
}

extern "C" void fp_index_destroy(fp_index* ix) {
  if (!ix) return;
  (void)fp_set_device(ix->device);
  for (Scratch* s : ix->pool) {
    s->destroy();
    delete s;
  }
  for (void* p : ix->owned) (void)hipFree(p);
  delete ix;
}

extern "C" int64_t fp_index_num_docs(const fp_index* ix) { return ix ? ix->d.N : 0; }
extern "C" int64_t fp_index_num_tokens(const fp_index* ix) { return ix ? ix->d.T : 0; }
extern "C" int64_t fp_index_num_centroids(const fp_index* ix) { return ix ? ix->d.C : 0; }
extern "C" int32_t fp_index_dim(const fp_index* ix) { return ix ? ix->d.dim : 0; }
extern "C" int32_t fp_index_nbits(const fp_index* ix) { return ix ? ix->d.nbits : 0; }
extern "C" int64_t fp_index_device_bytes(const fp_index* ix) { return ix ? ix->bytes : 0; }
extern "C" int64_t fp_index_num_unique_codes(const fp_index* ix) { return ix ? ix->d.U : 0; }
extern "C" int64_t fp_index_num_code_lines(const fp_index* ix) { return ix ? ix->d.n_lines * ix->d.l0_ppl / 8 : 0; }   // in 128-byte units
extern "C" int64_t fp_index_num_hard_tokens(const fp_index* ix) { return ix ? ix->n_hard_tokens : 0; }
extern "C" int32_t fp_index_tickets_ok(const fp_index* ix) { return ix && ix->tickets_ok ? 1 : 0; }

// ------------------------------------------------------------------------------------------
// synthetic device-resident corpora
// ------------------------------------------------------------------------------------------
extern "C" int fp_index_create_synthetic(const fp_synth_desc* d, int device_id, fp_index** out) {
  if (!d || !out) return fail(FP_EINVAL, "null argument");
  *out = nullptr;
  if (int rc = check_shape(d->nbits, d->dim)) return rc;
  if (d->n_centroids <= 1 || (d->n_centroids & (d->n_centroids - 1))) return fail(FP_EINVAL, "n_centroids must be a power of two");
  if (d->doc_begin < 0 || d->doc_end < d->doc_begin || d->doc_end > d->n_docs_total) return fail(FP_EINVAL, "bad document range");
  if (d->doc_len < 1) return fail(FP_EINVAL, "doc_len < 1");
  HIPCHK(fp_set_device(device_id));
  fp_index* ix = new fp_index();
  ix->device = device_id;
  ix->synthetic = true;
  auto bail = [&](int rc) {
    fp_index_destroy(ix);
    return rc;
  };
  FpSynthParams p{};
  p.nbits = d->nbits; p.dim = d->dim; p.pr = d->dim * d->nbits / 8; p.C = d->n_centroids;
  p.lgC = 0;
  while ((1ll << p.lgC) < p.C) ++p.lgC;
  p.n_docs_total = d->n_docs_total; p.doc_begin = d->doc_begin; p.doc_end = d->doc_end;
  p.doc_len = d->doc_len; p.variable_len = d->variable_len; p.seed = d->seed;
  FpIndexDev& D = ix->d;
  D.nbits = p.nbits; D.dim = p.dim; D.pr = p.pr; D.C = p.C; D.P = p.C; D.N = p.doc_end - p.doc_begin; D.pid_offset = p.doc_begin;
  std::vector<int64_t> hoff((size_t)D.N + 2, 0);
  int maxlen = 0;
  D.T = fps_doc_offsets_host(p, hoff.data(), &maxlen);
  const int64_t tok_base = hoff[(size_t)D.N + 1];
  D.max_doc_len = maxlen;
  ix->h_doc_off.assign(hoff.begin(), hoff.begin() + D.N + 1);
#define ICHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return bail(fail(FP_EHIP, std::string("HIP error: ") + hipGetErrorString(e_) + " at " #x)); } while (0)
  uint16_t* cent = nullptr; uint16_t* lut = nullptr; int64_t* doc_off = nullptr; int32_t* codes = nullptr; uint8_t* res = nullptr;
  int64_t* ivf_off = nullptr; int32_t* ivf_pids = nullptr;
  StreamGuard sg;
  hipStream_t& st = sg.st;
  ICHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  ICHK(dev_alloc(ix, &cent, (size_t)D.C * D.dim));
  ICHK(copy_sync(cent, d->centroids, (size_t)D.C * D.dim * 2, hipMemcpyHostToDevice, st));
  ix->cent_norm_max = max_row_norm_f16(d->centroids, D.C, D.dim);
  std::vector<uint16_t> hlut;
  build_lut_host(D.nbits, d->bucket_weights, hlut);
  ICHK(dev_alloc(ix, &lut, hlut.size()));
  ICHK(copy_sync(lut, hlut.data(), hlut.size() * 2, hipMemcpyHostToDevice, st));
  ICHK(dev_alloc(ix, &doc_off, (size_t)D.N + 1));
  ICHK(copy_sync(doc_off, ix->h_doc_off.data(), ((size_t)D.N + 1) * 8, hipMemcpyHostToDevice, st));
  ICHK(dev_alloc(ix, &codes, (size_t)D.T));
  ICHK(dev_alloc(ix, &res, (size_t)D.T * D.pr));
  fps_generate(p, doc_off, D.N, D.T, tok_base, codes, res, st);
  ICHK(hipStreamSynchronize(st));
  D.centroids = cent; D.lut = lut; D.doc_off = doc_off; D.codes = codes; D.residuals = res;
  if (int rc = finish_layout(ix, maxlen, st)) return bail(rc);
  ICHK(dev_alloc(ix, &ivf_off, (size_t)D.P + 1));
  int64_t tot = 0;
  int rc = fps_build_ivf(D.ucodes, D.uoff, D.N, D.U, D.P, &ivf_pids, &tot, ivf_off, st);
  if (ivf_pids) { ix->owned.push_back(ivf_pids); ix->bytes += tot * 4; }
  if (rc != 0) return bail(fail(FP_EHIP, "IVF build failed (hip/hipcub error " + std::to_string(rc) + ")"));
  ICHK(hipStreamSynchronize(st));
#undef ICHK
  ix->has_ivf = true;
  D.ivf_off = ivf_off; D.ivf_pids = ivf_pids;
  prepay_legacy_stream();
  *out = ix;
  return FP_OK;
}

// one stored residual row -> the reference's byte order (identity unless the index keeps k_maxsim6's unit order)
static inline void row_to_reference_order(const fp_index* ix, const uint8_t* stored, uint8_t* out) {
  const int pr = ix->d.pr;
  if (!ix->d.resid_native) { memcpy(out, stored, (size_t)pr); return; }
  const int nbits = ix->d.nbits, nu = ix->d.dim / 8;
  for (int b = 0; b < pr; ++b) out[b] = stored[fp_resid_pos(b, nbits, nu, 1)];
}

extern "C" int64_t fp_index_read_doc(const fp_index* ix, int64_t doc, int64_t* codes, uint8_t* residuals, int64_t cap) {
  if (!ix || doc < 0 || doc >= ix->d.N) return fail(FP_EINVAL, "bad document id");
  (void)fp_set_device(ix->device);
  const int64_t off = ix->h_doc_off[doc], len = ix->h_doc_off[doc + 1] - off;
  if (len > cap) return fail(FP_EINVAL, "capacity too small");
  const int pr = ix->d.pr;
  std::vector<int32_t> tmp((size_t)len);
  std::vector<uint8_t> rtmp((size_t)len * pr);
  std::vector<uint16_t> perm((size_t)len);
  if (len > 0) {
    HIPCHK(copy_sync(tmp.data(), ix->d.codes + off, (size_t)len * 4, hipMemcpyDeviceToHost, util_stream(ix->device)));
    HIPCHK(copy_sync(rtmp.data(), ix->d.residuals + off * pr, (size_t)len * pr, hipMemcpyDeviceToHost, util_stream(ix->device)));
    if (ix->d.perm) HIPCHK(copy_sync(perm.data(), ix->d.perm + off, (size_t)len * 2, hipMemcpyDeviceToHost, util_stream(ix->device)));
  }
  for (int64_t i = 0; i < len; ++i) {  // stored position i holds original token perm[i]
    const int64_t p = ix->d.perm ? perm[i] : i;
    codes[p] = tmp[i];
    row_to_reference_order(ix, rtmp.data() + i * pr, residuals + p * pr);
  }
  return len;
}

extern "C" int64_t fp_index_read_ivf(const fp_index* ix, int64_t cell, int64_t* pids, int64_t cap) {
  if (!ix || !ix->has_ivf || cell < 0 || cell >= ix->d.P) return fail(FP_EINVAL, "bad cell");
  (void)fp_set_device(ix->device);
  int64_t be[2];
  HIPCHK(copy_sync(be, ix->d.ivf_off + cell, 16, hipMemcpyDeviceToHost, util_stream(ix->device)));
  const int64_t len = be[1] - be[0];
  if (len > cap) return fail(FP_EINVAL, "capacity too small");
  std::vector<int32_t> tmp((size_t)len);
  if (len > 0) HIPCHK(copy_sync(tmp.data(), ix->d.ivf_pids + be[0], (size_t)len * 4, hipMemcpyDeviceToHost, util_stream(ix->device)));
  for (int64_t i = 0; i < len; ++i) pids[i] = tmp[i];
  return len;
}

// ------------------------------------------------------------------------------------------
// search pipeline
// ------------------------------------------------------------------------------------------
struct TraceOut {
  uint16_t* S; int64_t* cells; int64_t* cand; float* approx; int64_t* rerank; float* exact; int64_t* counts;
};

static int validate_search(const fp_index* ix, int32_t nq, int32_t q_len, int32_t dim, const fp_search_params* p) {
  if (!ix || !p) return fail(FP_EINVAL, "null argument");
  if (!ix->has_ivf)  // search.rs:227-232
    return fail(FP_ECOMPRESS_ONLY,
                "This index was built with compress_only=True and does not support search. Rebuild with compress_only=False "
                "to enable search.");
  if (nq < 0 || q_len < 1) return fail(FP_EINVAL, "Expected a 3D tensor for queries with at least one token per query");
  if (dim != ix->d.dim) return fail(FP_EINVAL, "query dim does not match the index dim");
  if (p->top_k < 0 || p->n_full_scores < 0 || p->n_ivf_probe < 0) return fail(FP_EINVAL, "negative search parameter");
  if ((int64_t)q_len * std::max<int64_t>(p->n_ivf_probe, 1) > (1ll << 28)) return fail(FP_EUNSUPPORTED, "q_len * n_ivf_probe > 2^28");
  if (p->n_full_scores / 4 > (1ll << 24)) return fail(FP_EUNSUPPORTED, "n_full_scores > 2^26");
  return FP_OK;
}

struct Pipe {
  fp_index* ix;
  Scratch* s;
  FpSearchShape sh;
  int64_t W, Cw;
  int nblk, nchunk;
  bool degenerate;  // reference errs for every query (topk k out of range) -> all results empty
  int64_t M;
  bool exact_all = false;  // trace mode: exact approximate score of EVERY candidate (no bound-and-refine)
  bool used_q8 = false;   // a bound stage pruned the candidates (8-bit bounds or level 0)
  int approx_impl = 0;    // 0 exact for all, 1 8-bit bounds, 2 level 0
  bool l0_ready = false;  // S1 produced level 0's floors and excess table (instead of the 8-bit table)
  bool l0_hot = false;    // ... for the hot-code scan (k_l0h_scan) instead of the code-line scan
  bool sub_shared = false;   // ONE subset list for every query of the batch (fp_search_shared_subset): h_sub_off is {0, n}
  bool allow_spec = false;   // fp_search / fp_search_device: M may be the learnt capacity instead of this batch's total (no mid-pipeline sync)
  bool spec = false;         // ... and it was
  bool probe_no_fb = false;  // the probe's tie-overflow fallback is NOT enqueued: the caller checks the flag after its sync and re-runs
  bool probe_fb_thr = false; // the threshold probe ran WITH its fallback behind it (the flag then reports an overflow, not a routing decision)
  bool probe_prezeroed = false, selhist_prezeroed = false, l0hist_prezeroed = false;   // cleared by the batch's first kernel
  int s1_mode = 0;           // FpS1Exact::mode of this batch's S1
  bool total_folded = false; // the candidate total / probe flag come down with the result block (no copy of their own)
  bool want_lazy = false;    // the caller allows S1's lazy form (fp_search / fp_search_device without subset, trace or probe fallback)
  bool lazy = false;         // ... and this batch runs it (FpLazyS1)
};

// after a batch: the candidate capacity the next batch of this shape runs on (see run_front)
static void learn_capacity(Scratch* s, int64_t M_true) {
  static const int pct = [] { const int v = (int)fp_test_opt("spec_cap_pct", 125); return v > 0 ? v : 125; }();   // tests: < 100 forces the re-run
  const int64_t want = M_true * pct / 100 + 1024;
  // The capacity is part of the graph's key: raising it costs two calls outside the replay (one plain, one capturing).  Until
  // round 6 it followed every new maximum of the totals (125 % of it) although the capacity in force -- 125 % of an EARLIER
  // maximum -- still held the batch: with a distinct query batch per call the first dozens of calls kept re-capturing (the timed
  // region of the bench ran 2 % slower than its own repeats).  Now it moves when a total comes within 8 % of it.
  if (pct < 100 || s->spec_cap == 0 || M_true * 100 > s->spec_cap * 92) s->spec_cap = std::max(want, pct < 100 ? (int64_t)0 : s->spec_cap);
  s->spec_last = M_true;
}

// stages S1..S5 for one sub-batch whose fp16 queries are already in s->qin
static int run_front(Pipe& P, const int64_t* h_sub_ids, const int64_t* h_sub_off /*B+1 rebased*/, bool has_subset) {
  fp_index* ix = P.ix;
  Scratch* s = P.s;
  const FpIndexDev& D = ix->d;
  const FpSearchShape& sh = P.sh;
  hipStream_t st = s->st;
  const int B = sh.B;
  HIPCHK(s->qpad.ensure((size_t)B * sh.Qp * D.dim * 2));
  // The counters, flags and histograms the later stages expect zeroed are cleared by the batch's first kernel (every
  // hipMemsetAsync is a launch of its own: five of them were ~20 us of a 400 us one-query search).
  FpZeroList zl{};
  P.nchunk = (int)std::min<int64_t>(16, std::max<int64_t>(1, D.C / 2048));
  HIPCHK(s->partial.ensure(fpk_probe_scratch_bytes(D, sh, P.nchunk)));
  HIPCHK(s->invalid.ensure((size_t)B * 4));
  HIPCHK(s->tickets.ensure((size_t)(2 * B + 4) * 4));   // [B + 1] candidate lists, [B + 1] survivor lists
  // "The last workgroup finishes the job" (fp_kernels.hip, fp_publish / fp_read_published) publishes counts with device-scope
  // atomic exchanges and reads them back with device-scope atomic loads -- no fence, which is what makes it cheap, and which
  // relies on gfx950 keeping device-scope atomics coherent at L2 (checked by the tests on this part, outside the letter of the
  // HIP memory model).  FP_TICKETS=0 takes the plain count -> scan -> offsets launches instead.
  static const bool tickets_env = [] { const char* e = getenv("FP_TICKETS"); return !(e && atoi(e) == 0); }();
  const bool tickets_on = tickets_env && ix->tickets_ok;   // (the self-test of the index's device, fp_index_create)
  HIPCHK(s->hist.ensure(fpk_sel_hist_bytes(B)));
  {
    void* zp = nullptr;
    size_t zb = 0;
    // (a shape the threshold probe does not serve -- n_ivf_probe > 32, FP_PROBE_FALLBACK -- always runs the other kernels: their
    // raised flag is no overflow and must not send the batch round again)
    const bool thr_ok = sh.n_probe >= 1 && fpk_probe_zero_region(D, sh, P.nchunk, s->partial.as<unsigned long long>(), &zp, &zb);
    if (!thr_ok) P.probe_no_fb = false;
    P.probe_fb_thr = thr_ok && !P.probe_no_fb;
    P.probe_prezeroed = thr_ok && zl.add(zp, zb);
    (void)zl.add(s->invalid.p, (size_t)B * 4);
    (void)zl.add(s->tickets.p, (size_t)(2 * B + 4) * 4);
    P.selhist_prezeroed = zl.add(s->hist.p, fpk_sel_hist_bytes(B));
    P.l0hist_prezeroed = false;
    if (fpk_l0_fits(D)) {
      HIPCHK(s->l0_hist.ensure(fpk_l0_hist_bytes(B)));
      P.l0hist_prezeroed = zl.add(s->l0_hist.p, fpk_l0_hist_bytes(B));
    }
  }
  // S1's lazy form (FpLazyS1): where the threshold probe and the general selection serve the shape
  static const int s1x_env = [] { const char* e = getenv("FP_S1_EXACT"); return e ? atoi(e) : 3; }();
  P.lazy = P.want_lazy && s1x_env == 3 && sh.Qp <= 128 && D.C < (1ll << 24) && fpk_probe_lazy_ok(D, sh, P.nchunk) && fpk_select_lazy_ok(sh) &&
           D.ucodes != nullptr;   // (C < 2^24: k_lz_exact's pair list packs code << 8 | column)
  const size_t lz_bytes = ((size_t)(2 * B + 8) * 4 + 15) & ~(size_t)15;   // zeroed: [B] negative-maximum flags | [B] maybes | flag, pad x 3 | stats x 4
  const int lz_gcap = fpk_select_lazy_gcap(sh);
  if (P.lazy) {
    HIPCHK(s->lz_state.ensure(lz_bytes));
    HIPCHK(s->lz_gpid.ensure((size_t)B * lz_gcap * 4));
    HIPCHK(s->lz_gval.ensure((size_t)B * lz_gcap * 4));
    HIPCHK(s->lz_slackq.ensure((size_t)2 * B * sh.Qp * 4));
    if (!zl.add(s->lz_state.p, lz_bytes)) HIPCHK(hipMemsetAsync(s->lz_state.p, 0, lz_bytes, st));
  }
  // S1 exact mode (FpS1Exact): certification window w0 |q_n| + kappa |x| around the MFMA result; FP_S1_EXACT=0 switches the
  // certification off (S then differs from the reference's matmul by one fp16 ulp in ~0.05 % of its entries), 2 re-evaluates
  // every entry (tests); FP_S1_W0_LOG2 / FP_S1_KAPPA_LOG2 move the window (defaults 2^-21.5 and 2^-20, x dim / 128 above 128)
  static const float s1x_w0 = std::exp2((float)fp_test_opt("s1_w0_log2", -21.5));
  static const float s1x_kappa = std::exp2((float)fp_test_opt("s1_kappa_log2", -20.0));
  const bool s1x_stats = s1_stats_enabled() && !s->capturing;
  P.s1_mode = P.lazy ? 3 : (s1x_env <= 0 ? 0 : (s1x_env == 2 ? 2 : 1));   // (FP_S1_EXACT=3, the default: lazy where it applies, eager elsewhere)
  const float dim_scale = D.dim > 128 ? (float)D.dim / 128.0f : 1.0f;
  FpS1Exact xe{P.s1_mode, nullptr, s1x_kappa * dim_scale, nullptr};
  if (P.s1_mode) {
    HIPCHK(s->wcol.ensure((size_t)B * sh.Qp * 4 + 16));
    xe.wcol = s->wcol.as<float>();
    if (s1x_stats) {
      HIPCHK(s->s1stats.ensure(32));
      HIPCHK(hipMemsetAsync(s->s1stats.p, 0, 32, st));
      xe.stats = s->s1stats.as<unsigned long long>();
    }
  }
  // S1 runs on the zero-padded view of the table when the index has one (dims below 256 other than 64 / 128)
  FpIndexDev D1 = D;
  const uint16_t* q_s1 = s->qpad.as<uint16_t>();
  if (D.cent_s1) {
    HIPCHK(s->qpad_s1.ensure((size_t)B * sh.Qp * D.dim_s1 * 2));
    D1.centroids = D.cent_s1;
    D1.dim = D.dim_s1;
    q_s1 = s->qpad_s1.as<uint16_t>();
  }
  fpk_pack_queries(s->qin.as<uint16_t>(), s->qpad.as<uint16_t>(), B, sh.Q, sh.Qp, D.dim, st, &zl, P.s1_mode ? s->wcol.as<float>() : nullptr,
                   s1x_w0 * dim_scale * ix->cent_norm_max, D.cent_s1 ? s->qpad_s1.as<uint16_t>() : nullptr, D.dim_s1);
  FpLazyS1 lzs{};
  if (P.lazy) {
    uint32_t* lzw = s->lz_state.as<uint32_t>();
    lzs = FpLazyS1{s->wcol.as<float>(), xe.kappa, 1.0f / (s1x_w0 * dim_scale), s->qpad.as<uint16_t>(), nullptr, s->lz_slackq.as<float>(),
                   s->lz_slackq.as<float>() + (size_t)B * sh.Qp, lzw, lzw + B, s->lz_gpid.as<int32_t>(), s->lz_gval.as<float>(), lz_gcap,
                   reinterpret_cast<int32_t*>(lzw + 2 * B), s1x_stats ? lzw + 2 * B + 4 : nullptr};
  }
  const FpLazyS1* lz = P.lazy ? &lzs : nullptr;
  STAGE_DONE(ST_CENTROID);
  HIPCHK(s->S.ensure((size_t)B * D.C * sh.Qp * 2));
  lzs.S = s->S.as<uint16_t>();
  // by-products of S1: 8-bit bins for S4's bound stage (decided for real once the candidate count is known) and
  // per-128-centroid column maxima for the threshold probe
  // FP_APPROX_IMPL: exact | q8 | l0 force a form of S4 (tests); otherwise chosen below from the table size and the candidate count
  static const int q8_env = [] {   // exact | q8 | l0 (code lines, sum of excesses) | l0h (hot codes, per-column maxima)
    const char* e = getenv("FP_APPROX_IMPL");
    return !e ? 0 : (e[0] == 'q' ? 1 : (e[0] == 'e' ? -1 : (e[0] == 'l' ? ((e[1] == '0' && e[2] == 'h') ? 3 : 2) : 0)));
  }();
  // (bound stages: level 0 for q_len <= 128, the 8-bit stage for q_len <= 64.)  Level 0 has two scans over the same excess table:
  // the sum of excesses along a document's packed code line (documents of <= 64 distinct codes on average: the index carries
  // the lines) and, for documents with MORE codes, the per-column maxima over the document's hot codes (k_l0h_scan; round 4 --
  // until then such indexes took the 8-bit stage, which gathers a row of bins for every code).  Either is dropped for the
  // scratch when it lets more than a quarter of the candidates through two batches running.
  const bool lines_ok = (q8_env == 0 || q8_env == 2) && fpk_l0_fits(D) && (q8_env == 2 || (D.U <= 64 * D.N && s->l0_poor < 2));
  // (the hot form pays from ~128 distinct codes per document: at 77 -- the Gaussian-mixture corpus of tools/bench_gmm.py -- floors low
  // enough for the top documents to rise above them leave a third of the codes hot, and the 8-bit stage is as fast)
  const bool hot_ok = !lines_ok && (q8_env == 0 || q8_env == 3) && fpk_l0h_fits(D) && (q8_env == 3 || (D.U > 128 * D.N && s->l0h_poor < 2));
  const bool l0_possible = lines_ok || hot_ok;
  P.l0_hot = hot_ok;
  const bool qp_ok = sh.Qp == 32 || sh.Qp == 64 || (sh.Qp == 128 && l0_possible);
  const bool want_s8 = !P.exact_all && qp_ok && q8_env >= 0 && (q8_env > 0 || D.C * 64 >= (2ll << 20));
  // Level 0 is decided HERE when the index suits it (the table fits, documents of <= 64 distinct codes on average; forced by
  // FP_APPROX_IMPL=l0): S1 then emits level 0's excess table from its epilogue -- floors from a pre-pass over a centroid
  // sample -- and the 8-bit table (C x Qp bytes per query, written and read back twice) is never materialised.  Whether S4
  // then prunes with it or scores everything exactly (few candidates) is still decided from the candidate count.
  P.l0_ready = want_s8 && l0_possible;
  const int nch128 = (int)((D.C + 127) / 128);
  HIPCHK(s->cmax128.ensure((size_t)B * sh.Qp * nch128 * 2));
  if (P.l0_ready) {
    const int64_t Cpad = (D.C + 15) & ~15ll;
    int64_t ns = 0, sstride = 1;
    fpk_l0_sample_plan(D, &ns, &sstride);
    HIPCHK(s->S8.ensure((size_t)B * ns * sh.Qp));          // bins of the sample only
    HIPCHK(s->Ssample.ensure((size_t)B * ns * sh.Qp * 2));
    HIPCHK(s->l0_floors.ensure((size_t)B * sh.Qp));
    HIPCHK(s->l0_F.ensure((size_t)B * 4));
    HIPCHK(s->l0_e8.ensure((size_t)B * Cpad));
    HIPCHK(s->l0_esc.ensure((size_t)B * 64 * 4));
    if (fpk_centroid_scores(D1, q_s1, s->Ssample.as<uint16_t>(), B, sh.Qp, s->S8.as<uint8_t>(), nullptr, st, ns, sstride))
      return fail(FP_EUNSUPPORTED, "dim");
    HIPCHK(s->l0_gfl.ensure((size_t)B * sh.Qp * 2));
    static const int s1_rd = fp_test_opt("s1_rd", 1) != 0 ? 1 : 0;   // the epilogue's one-fma form of the excess (0: the floor / clamp / subtract form, for A/B runs)
    fpk_l0_floors(s->S8.as<uint8_t>(), ns, sh, s->l0_floors.as<uint8_t>(), s->l0_F.as<uint32_t>(), s->l0_esc.as<uint32_t>(), s->l0_gfl.as<uint16_t>(), st,
                  P.l0_hot ? std::min(0.02f, std::max(0.001f, (float)((double)std::max<int64_t>(D.N, 1) / (1.2 * (double)std::max<int64_t>(D.U, 1))))) : 0.f,
                  s1_rd);
    if (Cpad > D.C) HIPCHK(hipMemsetAsync(s->l0_e8.p, 0, (size_t)B * Cpad, st));   // pad entries behind the table stay 0
    FpS1Excess ex{s->l0_floors.as<uint8_t>(), s->l0_gfl.as<uint16_t>(), s->l0_e8.as<uint8_t>(), s->l0_esc.as<uint32_t>(), Cpad, sh.Q, s1_rd};
    STAGE_DONE(ST_S1MAIN);
    if (fpk_centroid_scores(D1, q_s1, s->S.as<uint16_t>(), B, sh.Qp, nullptr, s->cmax128.as<uint16_t>(), st, 0, 1, &ex, &xe))
      return fail(FP_EUNSUPPORTED, "dim");
  } else {
    if (want_s8) HIPCHK(s->S8.ensure((size_t)B * D.C * sh.Qp));
    STAGE_DONE(ST_S1MAIN);
    if (fpk_centroid_scores(D1, q_s1, s->S.as<uint16_t>(), B, sh.Qp, want_s8 ? s->S8.as<uint8_t>() : nullptr,
                            s->cmax128.as<uint16_t>(), st, 0, 1, nullptr, &xe))
      return fail(FP_EUNSUPPORTED, "dim");
  }
  STAGE_DONE(ST_PROBE);
  // subset bitmaps
  P.W = ((D.N + 31) / 32 + 63) & ~63ll;
  if (P.W < 64) P.W = 64;
  P.Cw = (D.C + 31) / 32;
  if (has_subset) {
    const int Bl = P.sub_shared ? 1 : B;   // lists
    const int64_t ns = h_sub_off[Bl];
    int64_t max_len = 0;
    for (int i = 0; i < Bl; ++i) max_len = std::max(max_len, h_sub_off[i + 1] - h_sub_off[i]);
    HIPCHK(s->sub_ids.ensure((size_t)std::max<int64_t>(ns, 1) * 8));
    HIPCHK(s->sub_off.ensure((size_t)(B + 1) * 8));
    if (ns > 0) HIPCHK(hipMemcpyAsync(s->sub_ids.p, h_sub_ids, (size_t)ns * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(s->sub_off.p, h_sub_off, (size_t)(Bl + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(s->subbm.ensure((size_t)B * P.W * 4));
    HIPCHK(s->allow.ensure((size_t)B * P.Cw * 4));
    HIPCHK(hipMemsetAsync(s->subbm.p, 0, (size_t)Bl * P.W * 4, st));
    HIPCHK(hipMemsetAsync(s->allow.p, 0, (size_t)Bl * P.Cw * 4, st));
    fpk_subset_prepare(D, s->sub_ids.as<int64_t>(), s->sub_off.as<int64_t>(), Bl, s->subbm.as<uint32_t>(), P.W, s->allow.as<uint32_t>(),
                       P.Cw, s->invalid.as<int32_t>(), st, max_len, P.sub_shared ? B : 0);
  }
  // S2
  const int np = (int)std::max<int64_t>(sh.n_probe, 1);
  int NP = 1;
  while (NP < np) NP <<= 1;
  (void)NP;
  HIPCHK(s->cells.ensure((size_t)B * sh.Q * np * 4));
  HIPCHK(s->ucells.ensure((size_t)B * sh.Q * np * 4));
  HIPCHK(s->ncells.ensure((size_t)B * 4));
  if (sh.n_probe >= 1) {
    if (fpk_probe(D, s->S.as<uint16_t>(), sh, has_subset ? s->allow.as<uint32_t>() : nullptr, s->partial.as<unsigned long long>(),
                  P.nchunk, s->cells.as<int32_t>(), s->ucells.as<int32_t>(), s->ncells.as<int32_t>(), s->cmax128.as<uint16_t>(), st,
                  P.probe_prezeroed, !P.probe_no_fb, P.s1_mode != 0 ? &xe : nullptr, lz))
      return fail(FP_EUNSUPPORTED, "n_ivf_probe");
  } else {
    HIPCHK(hipMemsetAsync(s->ncells.p, 0, (size_t)B * 4, st));  // topk(0) -> no cells -> empty result
  }
  STAGE_DONE(ST_IVF);
  // S3
  HIPCHK(s->bitmap.ensure((size_t)B * P.W * 4));
  // k_ivf_mark writes every word of the bitmap (tiles are built in LDS): no memset needed
  fpk_ivf_mark(D, s->ucells.as<int32_t>(), s->ncells.as<int32_t>(), sh.Q * np, B, s->bitmap.as<uint32_t>(), P.W, st);
  P.nblk = (int)((P.W + fpk_cand_words_per_block() - 1) / fpk_cand_words_per_block());
  HIPCHK(s->blkcnt.ensure((size_t)B * P.nblk * 4));
  HIPCHK(s->ncand.ensure((size_t)B * 4));
  HIPCHK(s->cand_off.ensure((size_t)(B + 1) * 8));
  // The candidate total sizes the buffers and grids of S4 / S5.  The first batch of a shape waits for it (the one mid-pipeline
  // host sync, ~30 us of idle GPU); later batches of the same shape run on with the capacity learnt so far (125 % of the largest
  // total seen) as M -- every kernel below takes its ranges from cand_off, M only sizes things.  A batch that exceeds the
  // capacity is emptied on the device (k_cand_offsets) and run again by the caller once the true total is known.
  const bool spec_env = true;
  const int64_t key[4] = {B, sh.Q, sh.n_probe, has_subset ? 1 : 0};
  if (P.allow_spec && (s->spec_key[0] != key[0] || s->spec_key[1] != key[1] || s->spec_key[2] != key[2] || s->spec_key[3] != key[3])) {
    s->spec_cap = 0;
    for (int i = 0; i < 4; ++i) s->spec_key[i] = key[i];
  }
  P.spec = P.allow_spec && spec_env && s->spec_cap > 0;
  HIPCHK(s->spec_total.ensure(32));   // {candidate total, fp_shard_search's overflow mark (int32 at byte 8), the probe's overflow flag (int32 at byte 16)}
  int64_t* d_total = s->spec_total.as<int64_t>();
  fpk_cand_count(s->bitmap.as<uint32_t>(), has_subset ? s->subbm.as<uint32_t>() : nullptr, s->invalid.as<int32_t>(), B, P.W,
                 s->blkcnt.as<int32_t>(), P.nblk, s->ncand.as<int32_t>(), s->cand_off.as<int64_t>(), st, P.spec ? s->spec_cap : 0,
                 s->invalid.as<int32_t>(), d_total, tickets_on ? s->tickets.as<uint32_t>() : nullptr,
                 sh.n_probe >= 1 ? fpk_probe_flag(D, sh, P.nchunk, s->partial.as<unsigned long long>()) : nullptr);
  HIPCHK(s->h_small.ensure(4096));
  // total (8 bytes) and, at byte 16, the probe flag: one copy.  (Bytes 8..15 of the pinned block are rewritten by the survivor
  // total's copy further down when there is one.)
  if (sh.n_probe < 1) HIPCHK(hipMemsetAsync(reinterpret_cast<char*>(d_total) + 16, 0, 4, st));
  // (a speculative host-buffer search reads them after its final sync only: they travel in the result block's statistics instead,
  // one copy node and its ~10 us of dispatch less)
  P.total_folded = P.spec && s->fold_stats;
  if (!P.total_folded) HIPCHK(hipMemcpyAsync(s->h_small.p, d_total, 24, hipMemcpyDeviceToHost, st));
  STAGE_DONE(ST_COMPACT);
  int64_t M;
  if (P.spec) {
    M = s->spec_cap;
  } else {
    HIPCHK(hipStreamSynchronize(st));
    M = *reinterpret_cast<const int64_t*>(s->h_small.p);
  }
  P.M = M;
  HIPCHK(s->cand_pid.ensure((size_t)std::max<int64_t>(M, 1) * 4));
  HIPCHK(s->approx.ensure((size_t)std::max<int64_t>(M, 1) * 4));
  fpk_cand_compact(s->bitmap.as<uint32_t>(), has_subset ? s->subbm.as<uint32_t>() : nullptr, s->invalid.as<int32_t>(), B, P.W,
                   s->blkcnt.as<int32_t>(), P.nblk, s->cand_off.as<int64_t>(), s->cand_pid.as<int32_t>(), st);
  STAGE_DONE(ST_APREP);
  // S4.  Three forms, all giving the selection of "score every candidate exactly":
  //   level 0  (k_l0_scan): a scalar bound per centroid in LDS prunes ~99 % of the candidates without touching S -- when the
  //            byte table of C entries fits LDS and there is something to prune;
  //   8-bit bounds (k_approx_q8): when one query's slice of S overflows L2 (and level 0 does not apply);
  //   exact for all (k_approx).
  const int64_t* sel_off = s->cand_off.as<int64_t>();
  const int32_t* sel_src = s->cand_pid.as<int32_t>();
  // level 0 sums the excesses of a document's codes where the score takes their per-column maximum: with hundreds of distinct
  // codes per document (cfg4's 1024-token documents: ~300) the bound is too loose to prune (61 % of the candidates survived),
  // so it is prepared (P.l0_ready, before S1) only for documents of at most 64 distinct codes on average
  const bool l0_ok = P.l0_ready && M > 0;
  const bool l0_auto = l0_ok && M > 16 * (int64_t)B * sh.R;
  // 8-bit bounds: (a) one query's slice of S overflows L2, (b) there is something to prune and (c) one query alone can
  // fill the chip (~1280 resident workgroups x 128 lane pairs; the kernel spreads a candidate over 4 pairs when there are
  // fewer than 131072 candidates per query, e.g. cfg4's 1024-token documents or a small shard)
  const bool q8_auto = !P.l0_ready && D.C * 64 >= (2ll << 20) && M > 4 * (int64_t)B * sh.R && M / B >= 32768;
  P.approx_impl = 0;
  if (want_s8 && M > 0) {
    if (q8_env == 2 || q8_env == 3) P.approx_impl = l0_ok ? 2 : 0;
    else if (q8_env == 1) P.approx_impl = P.l0_ready ? 0 : 1;
    else P.approx_impl = l0_auto ? 2 : (q8_auto ? 1 : 0);
  }
  P.used_q8 = P.approx_impl != 0;
  if (P.approx_impl != 0) {
    const int64_t max_n = std::min<int64_t>(D.N, M);
    const int nblk2 = (int)((max_n + FP_SURV_CHUNK - 1) / FP_SURV_CHUNK);
    HIPCHK(s->cut.ensure((size_t)B * 4));
    HIPCHK(s->blkcnt2.ensure((size_t)B * nblk2 * 4));
    HIPCHK(s->nsurv.ensure((size_t)B * 4));
    HIPCHK(s->surv_off.ensure((size_t)(B + 1) * 8));
    HIPCHK(s->surv_pid.ensure((size_t)M * 4));
    if (P.approx_impl == 2) {
      const int64_t Cpad = (D.C + 15) & ~15ll;
      HIPCHK(s->l0_floors.ensure((size_t)B * sh.Qp));
      HIPCHK(s->l0_F.ensure((size_t)B * 4));
      HIPCHK(s->l0_e8.ensure((size_t)B * Cpad));
      HIPCHK(s->l0_esc.ensure((size_t)B * 64 * 4));
      HIPCHK(s->l0_ub.ensure(((size_t)M + 8 * (size_t)B + 16) * 2));   // rows start on 16-byte boundaries (l0_row)
      const int64_t ub_stride = ((int64_t)M + 8 * (int64_t)B + 16 + 7) & ~7ll;
      if (D.n_ranges > 1) HIPCHK(s->l0_ubp.ensure((size_t)ub_stride * D.n_ranges * 2 + 64));
      const int64_t pcap = fpk_l0_pilot_cap();
      HIPCHK(s->l0_hist.ensure(fpk_l0_hist_bytes(B)));
      HIPCHK(s->l0_npilot.ensure((size_t)B * 4));
      HIPCHK(s->l0_pilot_pid.ensure((size_t)B * pcap * 4));
      HIPCHK(s->l0_pilot_approx.ensure((size_t)B * pcap * 4));
      HIPCHK(s->l0_pilot_idx.ensure((size_t)B * pcap * 4));
      HIPCHK(s->l0_capprox.ensure((size_t)M * 4));
      HIPCHK(s->l0_thr.ensure((size_t)B * 4));
      HIPCHK(s->l0_nextra.ensure((size_t)B * 4));
      HIPCHK(s->l0_xpid.ensure((size_t)M * 4));
      HIPCHK(s->l0_xdst.ensure((size_t)M * 4));
      HIPCHK(s->l0_blkx.ensure((size_t)B * nblk2 * 4));
      FpL0Scratch w{s->l0_floors.as<uint8_t>(), s->l0_F.as<uint32_t>(), s->l0_e8.as<uint8_t>(), s->l0_esc.as<uint32_t>(),
                    s->l0_ub.as<uint16_t>(), s->l0_hist.as<uint32_t>(), s->cut.as<int32_t>(), s->blkcnt2.as<int32_t>(), s->l0_blkx.as<int32_t>(),
                    (int)((max_n + FP_L0_CHUNK - 1) / FP_L0_CHUNK),
                    s->l0_npilot.as<int32_t>(), s->l0_pilot_pid.as<int32_t>(), s->l0_pilot_approx.as<float>(),
                    s->l0_pilot_idx.as<int32_t>(), s->l0_capprox.as<float>(), s->l0_thr.as<int32_t>(), s->l0_nextra.as<int32_t>(),
                    s->l0_xpid.as<int32_t>(), s->l0_xdst.as<int32_t>(), tickets_on ? s->tickets.as<uint32_t>() + (B + 1) : nullptr,
                    D.n_ranges > 1 ? s->l0_ubp.as<uint16_t>() : nullptr, ub_stride};
      fpk_l0_prepare(D, nullptr, sh, w, st, P.l0hist_prezeroed);   // floors and table came with S1
      STAGE_DONE(ST_APPROX);
      // (grid: from the previous batch's total when M is the learnt capacity -- every extra workgroup copies the table into LDS;
      // the kernel strides if this batch has more)
      if (P.l0_hot) fpk_l0h_scan(D, s->S.as<uint16_t>(), sh, s->cand_off.as<int64_t>(), s->cand_pid.as<int32_t>(), P.spec && s->spec_last > 0 ? std::min(M, s->spec_last) : M, w, st);
      else
      fpk_l0_scan(D, sh, s->cand_off.as<int64_t>(), s->cand_pid.as<int32_t>(), P.spec && s->spec_last > 0 ? std::min(M, s->spec_last) : M, w, st);
      STAGE_DONE(ST_REFINE);
      fpk_l0_pilot(sh, s->cand_off.as<int64_t>(), s->cand_pid.as<int32_t>(), M, w, st);
      // exact scores of the pilot group (about FP_L0_PILOT x keep documents per query; ties of UB0 at the cut can add more)
      // (each score also lands at the document's candidate position: a pilot member that survives is not scored again)
      fpk_approx(D, s->S.as<uint16_t>(), sh, nullptr, w.pilot_pid, std::min<int64_t>(M, 6 * (int64_t)B * sh.R), w.pilot_approx, st, w.npilot, pcap,
                 w.cand_approx, w.pilot_idx, s->cand_off.as<int64_t>(), lz);
      fpk_l0_survivors(sh, s->cand_off.as<int64_t>(), s->cand_pid.as<int32_t>(), w, s->nsurv.as<int32_t>(), s->surv_off.as<int64_t>(),
                       s->surv_pid.as<int32_t>(), s->approx.as<float>(), st, lz);
      // survivors outside the pilot group (none when the threshold lies above the pilot cut, the usual case)
      fpk_approx(D, s->S.as<uint16_t>(), sh, s->surv_off.as<int64_t>(), w.xpid, std::min<int64_t>(M, 2 * (int64_t)B * sh.R), nullptr, st, w.nextra,
                 INT64_MAX, s->approx.as<float>(), w.xdst, s->surv_off.as<int64_t>(), lz);
    } else {
      const int nch8 = sh.Qp / 32;
      HIPCHK(s->kq.ensure((size_t)M * 4 * (nch8 > 1 ? 1 + nch8 : 1)));
      HIPCHK(s->q8hist.ensure((size_t)B * 8192 * nch8 * 4));
      STAGE_DONE(ST_APPROX);
      fpk_approx_q8_bounds(D, s->S8.as<uint8_t>(), sh, s->cand_off.as<int64_t>(), s->cand_pid.as<int32_t>(), M, s->kq.as<uint32_t>(), st);
      STAGE_DONE(ST_REFINE);
      fpk_approx_q8_cut(sh, s->cand_off.as<int64_t>(), s->cand_pid.as<int32_t>(), M, s->q8hist.as<uint32_t>(), s->kq.as<uint32_t>(),
                        s->cut.as<int32_t>(), s->blkcnt2.as<int32_t>(), nblk2, s->nsurv.as<int32_t>(), s->surv_off.as<int64_t>(),
                        s->surv_pid.as<int32_t>(), st, lz != nullptr);
    }
    sel_off = s->surv_off.as<int64_t>();
    sel_src = s->surv_pid.as<int32_t>();
    // survivor total for fp_last_search_counts: lands in pinned memory by the time the caller's final sync returns
    if (!s->fold_stats)
      HIPCHK(hipMemcpyAsync(static_cast<char*>(s->h_small.p) + 8, s->surv_off.as<int64_t>() + B, 8, hipMemcpyDeviceToHost, st));
    // grid sized for a few x R survivors per query (measured 1.7 x R at cfg2 after the 8-bit bounds); the kernel walks a
    // grid-stride loop if there are more
    if (P.approx_impl != 2)
      fpk_approx(D, s->S.as<uint16_t>(), sh, sel_off, sel_src, std::min<int64_t>(M, 8 * (int64_t)B * sh.R), s->approx.as<float>(), st, nullptr, 0,
                 nullptr, nullptr, nullptr, lz);
  } else {
    STAGE_DONE(ST_APPROX);
    fpk_approx(D, s->S.as<uint16_t>(), sh, sel_off, sel_src, M, s->approx.as<float>(), st, nullptr, 0, nullptr, nullptr, nullptr, lz);
    STAGE_DONE(ST_REFINE);
  }
  STAGE_DONE(ST_SELECT);
  // S5
  HIPCHK(s->hist.ensure(fpk_sel_hist_bytes(B)));
  HIPCHK(s->selstate.ensure((size_t)B * 8 * 4));
  HIPCHK(s->sel_pid.ensure((size_t)B * sh.R * 4));
  HIPCHK(s->sel_approx.ensure((size_t)B * sh.R * 4));
  HIPCHK(s->sel_cnt.ensure((size_t)B * 4));
  HIPCHK(s->tie_pid.ensure((size_t)B * sh.R * 4));
  HIPCHK(s->ms_pref.ensure((size_t)(B + 1) * 8));
  if (fpk_select(sh, sel_off, sel_src, s->approx.as<float>(), s->hist.as<uint32_t>(),
             s->selstate.as<uint32_t>(), s->sel_pid.as<int32_t>(), s->sel_approx.as<float>(), s->sel_cnt.as<int32_t>(),
             s->tie_pid.as<int32_t>(), st,
             /*short_lists: one workgroup per query; pays off when there are too few queries to fill the chip anyway (measured: B = 8
               46 vs 53 us, B = 64 90 vs 71 us)*/ !lz && B <= 16 && sh.R <= FP_MAX_SORT && (P.approx_impl != 0 || M <= 16384ll * B),
             P.selhist_prezeroed, s->ms_pref.as<int64_t>(), lz, &D,
             /*list length per query: the bound stages leave a few x R survivors, otherwise every candidate*/ sel_src != s->cand_pid.as<int32_t>() ? 8 * sh.R : (B > 0 ? M / B : 0)))
    return fail(FP_EINVAL, "internal: the lazy selection was asked for on a shape it does not serve");
  if (lz && !s->fold_stats)   // device-resident I/O: the overflow flag travels by itself (host-buffer calls: with the result block)
    HIPCHK(hipMemcpyAsync(static_cast<char*>(s->h_small.p) + 40, lz->flag, 4, hipMemcpyDeviceToHost, st));
  s->pref_ready = true;   // (whoever edits sel_cnt before S6 -- the sharded search's cut -- clears it)
  STAGE_DONE(ST_MAXSIM);
  return FP_OK;
}

// S6+S7 with the exact-order repair.  mode 1 (fp_search): MaxSim on MFMA, then the flagged columns of the documents that are
// near-tied in the final ranking are re-evaluated with the reference's ascending-k chain.  The sharded search splits this:
// mode 0 = MFMA pass only (scores + budgets stay in the scratch), mode 2 = repair every flagged document of the previous
// mode-0 pass (the final ranking is only known after the exchange).
static int run_maxsim(fp_index* ix, Scratch* s, const FpSearchShape& sh, int64_t R, int64_t top_k, int mode, bool stage_event = false,
                      bool pref_ready = false /*S5 of THIS batch left the count prefix in ms_pref and sel_cnt is untouched since*/) {
  const FpIndexDev& D = ix->d;
  hipStream_t st = s->st;
  const int B = sh.B;
  static const int repair_env = [] { const char* e = getenv("FP_MAXSIM_REPAIR"); return e ? atoi(e) : 1; }();   // 0: off, 2: every flagged document
  HIPCHK(s->exact.ensure((size_t)B * R * 4));
  HIPCHK(s->ms_pref.ensure((size_t)(B + 1) * 8));
  const bool repair = repair_env != 0 && fpk_maxsim_fast_shape(D.dim, D.nbits);
  s->ms_repairable = repair;
  FpMaxsimAux aux{nullptr, nullptr, nullptr, nullptr};
  if (repair) {
    HIPCHK(s->ms_cm16.ensure((size_t)B * R * sh.Qp * 2));
    HIPCHK(s->ms_unc.ensure((size_t)B * R * 4));
    HIPCHK(s->ms_uncm.ensure((size_t)B * R * 4));
    HIPCHK(s->ms_flags.ensure((size_t)B * R * (sh.Qp / 32) * 4));
    HIPCHK(s->ms_marks.ensure((size_t)B * R * 4));
    HIPCHK(s->ms_nmark.ensure((size_t)B * 4));
    aux = FpMaxsimAux{s->ms_cm16.as<uint16_t>(), s->ms_unc.as<float>(), s->ms_flags.as<uint32_t>(), s->ms_uncm.as<float>()};
  }
  if (mode != 2) {
    if (fpk_maxsim(D, s->qpad.as<uint16_t>(), sh, s->sel_pid.as<int32_t>(), s->sel_cnt.as<int32_t>(), R, s->exact.as<float>(),
                   s->ms_pref.as<int64_t>(), aux, st, pref_ready && s->pref_ready))
      return fail(FP_EUNSUPPORTED, "dim/nbits");
    s->pref_ready = false;
    LAUNCHCHK("MaxSim");
  }
  if (stage_event) STAGE_DONE(ST_REPAIR);   // "S6+S7 maxsim" is the MaxSim kernel alone (+ the 4 us prefix kernel in front of it)
  if (repair && mode != 0) {
    // the batch-wide work list needs a zeroed counter: the last word of the ticket block, cleared by the first kernel of a batch
    // whose front stages ran in this call (pref_ready says so)
    uint32_t* flat_n = nullptr;
    if (pref_ready && mode == 1 && s->tickets.p) {
      HIPCHK(s->ms_flat.ensure((size_t)B * R * 8));
      flat_n = s->tickets.as<uint32_t>() + (2 * B + 2);
    }
    const bool marked = mode == 1 && repair_env != 2 &&
                        fpk_final_mark(s->exact.as<float>(), aux.unc, aux.uncm, s->sel_cnt.as<int32_t>(), R, B, top_k, s->ms_marks.as<int32_t>(),
                                       s->ms_nmark.as<int32_t>(), st, flat_n, flat_n ? s->ms_flat.p : nullptr) == 0;
    s->ms_have_marks = false;
    s->ms_marked_now = marked;
    if (marked && !s->fold_stats && (size_t)B * 4 + 64 <= 4096) {   // marked documents per query -> pinned memory, summed for fp_last_search_counts after the final sync
      HIPCHK(s->h_small.ensure(4096));
      HIPCHK(hipMemcpyAsync(static_cast<char*>(s->h_small.p) + 64, s->ms_nmark.p, (size_t)B * 4, hipMemcpyDeviceToHost, st));
      s->ms_have_marks = true;
    }
    fpk_maxsim_repair(D, s->qpad.as<uint16_t>(), sh, s->sel_pid.as<int32_t>(), s->sel_cnt.as<int32_t>(), R,
                      marked ? s->ms_marks.as<int32_t>() : nullptr, marked ? s->ms_nmark.as<int32_t>() : nullptr, s->exact.as<float>(), aux, st,
                      marked ? flat_n : nullptr, marked && flat_n ? s->ms_flat.p : nullptr);
    LAUNCHCHK("MaxSim repair");
  }
  return FP_OK;
}

static FpSearchShape make_shape(int B, int Q, const fp_search_params* p) {
  FpSearchShape sh{};
  sh.B = B;
  sh.Q = Q;
  sh.Qp = fp_padded_qlen(Q);
  sh.n_probe = (int)p->n_ivf_probe;
  sh.n_full = p->n_full_scores;
  sh.R = std::max<int64_t>(p->n_full_scores / 4, 1);
  return sh;
}

// One sub-batch of fp_search / fp_search_device / fp_search_trace enqueued on its scratch's stream: query upload, S1 .. S8, the
// result block's download (host buffers: into s->h_out) or the copies into the caller's device buffers.  Nothing in here waits
// for the device when the batch runs on a learnt candidate capacity (P.spec is set then).
static int enqueue_batch(Pipe& P, const uint16_t* q_src, hipMemcpyKind q_kind, const fp_search_params* p, const OutLayout& ol,
                         const int64_t* sids, const int64_t* soff /*B+1 rebased*/, bool has_subset, bool trace, int64_t* dev_pids,
                         float* dev_scores, int32_t* dev_counts) {
  fp_index* ix = P.ix;
  Scratch* s = P.s;
  const FpIndexDev& D = ix->d;
  hipStream_t st = s->st;
  const int B = P.sh.B;
  const int64_t K = p->top_k;
  const bool dev_io = dev_pids != nullptr;
  const size_t qbytes = (size_t)B * P.sh.Q * D.dim * 2;
  STAGE_DONE(ST_UPLOAD);
  HIPCHK(s->qin.ensure(qbytes));
  HIPCHK(hipMemcpyAsync(s->qin.p, q_src, qbytes, q_kind, st));
  s->fold_stats = !dev_io;
  s->ms_marked_now = false;
  P.allow_spec = !trace && !has_subset;   // (subset searches: the candidate total follows the subset sizes, not the shape)
  if (int rc = run_front(P, sids, soff, has_subset)) return rc;
  // S6+S7
  const int64_t R = P.sh.R;
  if (int rc = run_maxsim(ix, s, P.sh, R, p->top_k, 1, true, true)) return rc;
  STAGE_DONE(ST_TOPK);
  // S8
  HIPCHK(s->out_all.ensure(ol.total));
  HIPCHK(s->h_out.ensure(ol.total));
  char* od = s->out_all.as<char>();
  if (const int trc = fpk_final_topk(s->exact.as<float>(), s->sel_pid.as<int32_t>(), nullptr, s->sel_cnt.as<int32_t>(), R, B, K, D.pid_offset,
                                     reinterpret_cast<int64_t*>(od), reinterpret_cast<float*>(od + ol.score_off), reinterpret_cast<int32_t*>(od + ol.cnt_off), st,
                                     (s->fold_stats && P.used_q8) ? s->surv_off.as<int64_t>() + B : nullptr,
                                     (s->fold_stats && s->ms_marked_now) ? s->ms_nmark.as<int32_t>() : nullptr,
                                     s->fold_stats ? reinterpret_cast<int64_t*>(od + ol.stat_off) : nullptr,
                                     P.lazy ? reinterpret_cast<const int32_t*>(s->lz_state.as<uint32_t>() + 2 * B) : nullptr,
                                     P.total_folded ? s->spec_total.as<int64_t>() : nullptr))
    return trc < 0 ? fail(FP_EUNSUPPORTED, "n_queries * max(n_full_scores / 4, 1) >= 2^31 in one sub-batch of the final ranking")
                   : fail(FP_EHIP, "final ranking failed (hip error " + std::to_string(trc) + ")");
  if (dev_io) {
    HIPCHK(hipMemcpyAsync(dev_pids, od, ol.nk * 8, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(dev_scores, od + ol.score_off, ol.nk * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(dev_counts, od + ol.cnt_off, ol.n * 4, hipMemcpyDeviceToDevice, st));
  } else {
    HIPCHK(hipMemcpyAsync(s->h_out.p, od, ol.total, hipMemcpyDeviceToHost, st));
  }
  STAGE_DONE(ST_N);   // (a capture records no stage events: its call reports zeros)
  return FP_OK;
}

// After the final sync of a sub-batch: the flags that void it, the learnt capacity, the graph bookkeeping, the call's counters
// and stage times.  POST_RETRY*: the batch must be run again (the state that makes the second attempt differ is already set).
enum { POST_DONE = 0, POST_RETRY = 1, POST_RETRY_EAGER = 2 };
struct BatchFlags { int64_t M_true; bool probe_ovf, lz_failed; };
// where the three words come from: the result block's statistics (speculative host-buffer batches and everything replayed) or the
// small copies of their own (first batches, device-resident I/O)
static BatchFlags batch_flags(const Scratch* s, const Pipe& P, const OutLayout& ol) {
  const int B = P.sh.B;
  const char* hs = static_cast<const char*>(s->h_small.p);
  const int64_t* stv = (s->fold_stats || P.total_folded) ? reinterpret_cast<const int64_t*>(static_cast<const char*>(s->h_out.p) + ol.stat_off) : nullptr;
  BatchFlags f;
  f.M_true = P.total_folded ? stv[2 + B] : *reinterpret_cast<const int64_t*>(hs);
  f.probe_ovf = P.total_folded ? stv[3 + B] != 0 : *reinterpret_cast<const int32_t*>(hs + 16) != 0;
  // S1's lazy form: the flag is 0 whenever the batch was not lazy (a replayed graph included)
  f.lz_failed = s->fold_stats ? stv[1 + B] != 0 : (P.lazy && *reinterpret_cast<const int32_t*>(hs + 40) != 0);
  return f;
}
static int post_batch(Scratch* s, const Pipe& P, const OutLayout& ol, bool replayed, bool capture, bool graph_ok, const int64_t (&gkey)[8],
                      bool first_sub_batch, bool stage_times = true) {
  const int B = P.sh.B;
  g_last_lazy = P.lazy ? 1 : (replayed ? -1 : 0);
  const BatchFlags f = batch_flags(s, P, ol);
  const int64_t M_true = f.M_true;
  if (P.probe_no_fb && f.probe_ovf) {
    // some column had more ties at its probe threshold than the candidate lists hold: the device probed nothing (k_probe_merge);
    // run the batch again with the register top-k fallback, and keep it enqueued for this scratch -- until 64 batches in a row
    // raise no flag (a tie-heavy batch is a property of its queries, not of the index)
    s->probe_fb = true;
    s->probe_fb_clean = 0;
    return POST_RETRY;
  }
  if (s->probe_fb && P.probe_fb_thr) {   // (the fallback is enqueued behind a threshold probe that still reports its overflows)
    if (f.probe_ovf) s->probe_fb_clean = 0;
    else if (++s->probe_fb_clean >= 64) { s->probe_fb = false; s->probe_fb_clean = 0; }
  }
  if (P.spec && M_true > P.M) {
    // more candidates than the capacity learnt from earlier batches: the device emptied the batch (k_cand_offsets); run it again,
    // this time waiting for the total
    s->spec_cap = 0;
    return POST_RETRY;
  }
  // S1's lazy form: a selection list overflowed (masses of near-tied approximate scores) -> the batch's results are void; run it
  // again with the eager S1.  Two such batches switch the lazy form off for the scratch -- not for good: after FP_LAZY_RETRY_AFTER
  // batches it is tried again (one overflowing batch then switches it off for the next stretch), and a run of clean lazy batches
  // forgets an old overflow.
  if (f.lz_failed) {
    s->lazy_fails++;
    s->lazy_clean = 0;
    s->lazy_off_batches = 0;
    return POST_RETRY_EAGER;
  }
  if (P.lazy || (replayed && P.want_lazy)) {
    if (s->lazy_fails == 1 && ++s->lazy_clean >= 64) s->lazy_fails = 0;
  } else if (s->lazy_fails >= 2 && ++s->lazy_off_batches >= 256) {
    s->lazy_fails = 1;
    s->lazy_off_batches = 0;
    s->lazy_clean = 0;
  }
  learn_capacity(s, M_true);
  if (graph_ok) {
    // remember what this call ran on: the next call may capture if it has the same shape, the capacity is still the one THIS
    // call sized every buffer for, and nothing was (re)allocated since
    if (P.spec && std::equal(gkey, gkey + 5, s->graph.last)) s->graph.warm++;
    else s->graph.warm = P.spec ? 1 : 0;
    std::copy(gkey, gkey + 8, s->graph.last);
    s->graph.last[5] = P.spec ? P.M : 0;
    s->graph.last[6] = (int64_t)s->alloc_gen;
  }
  if (first_sub_batch) g_last_counts[0] = g_last_counts[1] = g_last_counts[2] = g_last_counts[3] = 0;
  g_last_counts[4] = replayed ? -1 : (P.approx_impl == 2 && P.l0_hot ? 3 : P.approx_impl);
  g_last_counts[0] += M_true;
  g_last_counts[3] += 1;
  g_last_counts[5] = s->lazy_fails;
  if (s->fold_stats) {   // the statistics came down with the results
    const int64_t* stv = reinterpret_cast<const int64_t*>(static_cast<const char*>(s->h_out.p) + ol.stat_off);
    // Level 0 is worth its scan only while it prunes: a corpus whose documents' code sets overlap every query (survivors above a
    // quarter of the candidates in two batches running) is switched to the 8-bit bound stage for this scratch, for good (the
    // selection is the same in every form of S4, only the time differs)
    if (P.approx_impl == 2 && M_true > 0) {
      int& poor = P.l0_hot ? s->l0h_poor : s->l0_poor;
      if (stv[0] * 4 > M_true) poor = std::min(poor + 1, 4);
      else poor = 0;
    }
    g_last_counts[1] += P.used_q8 ? stv[0] : M_true;
    for (int i = 0; i < B; ++i) g_last_counts[2] += stv[1 + i];
  } else
    g_last_counts[1] += P.used_q8 ? *reinterpret_cast<const int64_t*>(static_cast<const char*>(s->h_small.p) + 8) : M_true;
  if (!s->fold_stats && s->ms_have_marks) {
    const int32_t* nm = reinterpret_cast<const int32_t*>(static_cast<const char*>(s->h_small.p) + 64);
    for (int i = 0; i < B; ++i) g_last_counts[2] += nm[i];
    s->ms_have_marks = false;
  }
  if (!stage_times) return POST_DONE;
  if (replayed || capture) {   // a graph launch records no stage events
    for (int i = 0; i < ST_N; ++i) g_last_ms[i] = 0.f;
    g_have_ms = true;
  } else if (first_sub_batch) {
    for (int i = 0; i < ST_N; ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, s->ev[i], s->ev[i + 1]) != hipSuccess) ms = 0.f;
      g_last_ms[i] = ms;
    }
    g_have_ms = true;
  } else {
    for (int i = 0; i < ST_N; ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, s->ev[i], s->ev[i + 1]) == hipSuccess) g_last_ms[i] += ms;
    }
  }
  return POST_DONE;
}

// dev_io: `queries` and the three outputs are DEVICE pointers on the index's GPU (fp_search_device); subsets stay host-side.
static int search_impl(fp_index* ix, const uint16_t* queries, int32_t nq, int32_t Q, const fp_search_params* p,
                       const int64_t* subset_ids, const int64_t* subset_off, int64_t* out_pids, float* out_scores,
                       int32_t* out_counts, TraceOut* tr, bool dev_io = false, int64_t shared_subset_n = -1 /*>= 0: subset_ids is ONE list of that many ids for every query (subset_off unused)*/) {
  const FpIndexDev& D = ix->d;
  HIPCHK(fp_set_device(ix->device));
  if (dev_io) {
    if (nq > 0) {
      hipStream_t us = util_stream(ix->device);
      HIPCHK(hipMemsetAsync(out_counts, 0, (size_t)nq * 4, us));
      HIPCHK(hipStreamSynchronize(us));
    }
  } else {
    for (int i = 0; i < nq; ++i) out_counts[i] = 0;
  }
  if (nq == 0 || p->top_k == 0) return FP_OK;
  const bool sub_shared = shared_subset_n >= 0;
  const bool has_subset = subset_off != nullptr || sub_shared;
  // reference: topk(k > C) errs inside search() -> per-query empty result (search.rs:268)
  if ((!has_subset && p->n_ivf_probe > D.C) || D.N == 0) {
    // (host buffers: the rows are written here too -- id -1, score 0 like every unused slot -- so that a caller need not pre-fill them)
    if (!dev_io) {
      std::fill(out_pids, out_pids + (size_t)nq * p->top_k, (int64_t)-1);
      std::fill(out_scores, out_scores + (size_t)nq * p->top_k, 0.f);
    }
    return FP_OK;
  }
  Scratch* s = acquire(ix);
  if (!s) return fail(FP_EHIP, "could not create a HIP stream");
  struct Rel {
    fp_index* ix; Scratch* s;
    ~Rel() {
      if (s->capturing) abandon_capture(s);   // an error return in the middle of a capture: close it, or the stream stays unusable
      release(ix, s);
    }
  } rel{ix, s};
  const bool graph_env = g_graph_replay.load(std::memory_order_relaxed) != 0;
  const int Qp = fp_padded_qlen(Q);
  // sub-batch so that the centroid-score table stays within a budget
  static const size_t budget_env = (size_t)fp_test_opt("s_budget_kb", 0) << 10;   // tests: force sub-batching
  const size_t perq = (size_t)D.C * Qp * 2;
  size_t free_b = 0, total_b = 0;
  if (!budget_env && (size_t)nq * perq > ((size_t)1 << 30))   // the driver call costs tens of microseconds: only when the table can exceed the floor of the budget
    (void)hipMemGetInfo(&free_b, &total_b);
  const size_t budget = budget_env ? budget_env : std::max<size_t>((size_t)1 << 30, std::min<size_t>((size_t)24 << 30, free_b / 3));
  int maxB = (int)std::max<size_t>(1, budget / perq);
  if (tr) maxB = 1;
  std::vector<int64_t> sub_off_local;
  bool eager_retry = false;   // the previous attempt at this sub-batch overflowed a list of the lazy S1: this one runs the eager form
  for (int b0 = 0; b0 < nq; b0 += maxB) {
    const int B = std::min(maxB, nq - b0);
    Pipe P{};
    P.ix = ix; P.s = s; P.sh = make_shape(B, Q, p);
    P.want_lazy = !tr && !has_subset && !s->probe_fb && !eager_retry && s->lazy_fails < 2;
    eager_retry = false;
    P.exact_all = tr != nullptr;   // the trace reports the approximate score of every candidate
    hipStream_t st = s->st;
    const int64_t K = p->top_k;
    OutLayout ol(B, K);
    // the state of this shape (learnt capacity, captured graph) becomes the live one
    if (!tr) {
      const int64_t skey[6] = {B, Q, p->n_ivf_probe, p->n_full_scores, p->top_k, has_subset ? 1 : 0};
      s->select_shape(skey);
    }
    // ---- FP_GRAPH: replay / capture (host-buffer calls of one sub-batch, no subset, no trace)
    const bool graph_ok = graph_env && !s1_stats_enabled() && !dev_io && !tr && !has_subset && nq <= maxB && s->graph.fails < 3 &&
                          P.sh.R <= FP_MAX_SORT;   // (beyond it the final ranking allocates and synchronises: not capturable)
    const size_t qbytes = (size_t)B * Q * D.dim * 2;
    P.probe_no_fb = !tr && !s->probe_fb;
    int64_t gkey[8] = {B, Q, p->n_ivf_probe, p->n_full_scores, p->top_k, s->spec_cap, (int64_t)s->alloc_gen, (s->probe_fb ? 2 : 1) + (P.want_lazy ? 4 : 0)};
    bool replayed = false;
    if (graph_ok && s->graph.valid && std::equal(gkey, gkey + 8, s->graph.key)) {
      memcpy(s->h_qin.p, queries, qbytes);
      if (hipGraphLaunch(s->graph.exec, st) == hipSuccess) {
        HIPCHK(hipStreamSynchronize(st));
        P.spec = true;
        P.M = s->spec_cap;
        P.used_q8 = s->graph.used_q8;
        P.total_folded = true;   // (a graph is only ever kept from a speculative host-buffer batch: its candidate total and probe flag are in the result block)
        // (whether a raised probe flag means "overflow, run again" is a property of the CAPTURED batch: a shape the threshold probe
        // does not serve -- n_ivf_probe > 32 -- raises the flag on purpose to route its register / select kernels.  Until round 6
        // the replay took the flag for an overflow: every replayed batch of such a shape was run twice and left the scratch on
        // the fallback + eager S1 for every later shape)
        P.probe_no_fb = s->graph.no_fb;
        P.probe_fb_thr = s->graph.fb_thr;
        s->fold_stats = true;
        s->ms_marked_now = s->graph.marked;
        replayed = true;
        g_graph_replays.fetch_add(1, std::memory_order_relaxed);
      } else {   // a runtime that will not launch the graph: plain path from here on
        (void)hipGetLastError();
        s->graph.valid = false;
        s->graph.fails = 1000;
      }
    }
    bool capture = false;
    if (graph_ok && !replayed) {
      // capture when the previous call was a speculative one with the same key (so that no buffer grows inside the capture)
      capture = s->spec_cap > 0 && s->graph.warm >= 1 && std::equal(gkey, gkey + 8, s->graph.last);
      if (capture) {
        HIPCHK(s->h_qin.ensure(qbytes));
        gkey[6] = (int64_t)s->alloc_gen;
        memcpy(s->h_qin.p, queries, qbytes);
        if (s->graph.exec) { (void)hipGraphExecDestroy(s->graph.exec); s->graph.exec = nullptr; }
        s->graph.valid = false;
        if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) == hipSuccess) {
          s->capturing = true;
          g_open_captures.fetch_add(1, std::memory_order_acq_rel);
        } else {   // no capture on this runtime / stream: plain path from here on
          (void)hipGetLastError();
          s->graph.fails = 1000;
          capture = false;
        }
      }
    }
    if (!replayed) {
    const int64_t* sids = nullptr;
    if (sub_shared) {
      sub_off_local.assign({0, shared_subset_n});
      sids = subset_ids;
      P.sub_shared = true;
    } else if (has_subset) {
      sub_off_local.resize((size_t)B + 1);
      for (int i = 0; i <= B; ++i) sub_off_local[i] = subset_off[b0 + i] - subset_off[b0];
      sids = subset_ids + subset_off[b0];
    }
    if (int rc = enqueue_batch(P, capture ? static_cast<const uint16_t*>(s->h_qin.p) : queries + (size_t)b0 * Q * D.dim,
                               dev_io ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, p, ol, sids, has_subset ? sub_off_local.data() : nullptr,
                               has_subset, tr != nullptr, dev_io ? out_pids + (size_t)b0 * K : nullptr, dev_io ? out_scores + (size_t)b0 * K : nullptr,
                               dev_io ? out_counts + b0 : nullptr)) {
      if (!capture) return rc;
      // an error INSIDE a capture need not be the batch's own: any thread of the process that makes a call the runtime refuses
      // while a capture is open (a synchronous copy on the legacy stream, say -- this library makes none, the application may)
      // invalidates every open capture.  Nothing of the batch has run: close the capture, stop capturing on this scratch and
      // run the batch on the plain path.
      abandon_capture(s);
      b0 -= maxB;
      continue;
    }
    if (dev_io) {
      HIPCHK(hipStreamSynchronize(st));   // results are complete in HBM when the call returns
    } else {
      if (capture) {
        // nothing has run yet: close the capture, keep the executable graph if the call was what a replay needs (speculative, no
        // buffer moved), and launch it -- or fall back to the plain path for this batch
        hipGraph_t g = nullptr;
        const hipError_t ce = hipStreamEndCapture(st, &g);
        s->capturing = false;
        g_open_captures.fetch_sub(1, std::memory_order_acq_rel);
        hipGraphExec_t ex = nullptr;
        const bool usable = ce == hipSuccess && g && P.spec && gkey[6] == (int64_t)s->alloc_gen &&
                            hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess && ex;
        if (g) (void)hipGraphDestroy(g);
        if (!usable) {
          (void)hipGetLastError();
          if (ex) (void)hipGraphExecDestroy(ex);
          s->graph.fails++;
          s->graph.warm = 0;
          if (ce != hipSuccess) fresh_stream(s);   // (invalidated from outside: see abandon_capture)
          b0 -= maxB;
          continue;
        }
        s->graph.exec = ex;
        s->graph.fails = 0;
        std::copy(gkey, gkey + 8, s->graph.key);
        s->graph.used_q8 = P.used_q8;
        s->graph.marked = s->ms_marked_now;
        s->graph.no_fb = P.probe_no_fb;
        s->graph.fb_thr = P.probe_fb_thr;
        s->graph.valid = true;
        if (hipGraphLaunch(ex, st) != hipSuccess) {   // nothing of this batch has run: take the plain path, for good
          (void)hipGetLastError();
          s->graph.valid = false;
          s->graph.fails = 1000;
          b0 -= maxB;
          continue;
        }
        g_graph_replays.fetch_add(1, std::memory_order_relaxed);   // (the capturing call is itself served by the graph's first launch)
      }
      HIPCHK(hipStreamSynchronize(st));
    }
    }   // !replayed
    if (!dev_io) ol.scatter(s->h_out.p, out_pids + (size_t)b0 * K, out_scores + (size_t)b0 * K, out_counts + b0);
    if (s1_stats_enabled() && !replayed && !capture && s->s1stats.p) {   // diagnostics: blocking copy after the call's sync
      uint64_t h4[4] = {0, 0, 0, 0};
      HIPCHK(copy_sync(h4, s->s1stats.p, 32, hipMemcpyDeviceToHost, s->st));
      if (P.lazy) {   // the lazy form's counters instead: {entries gathered by the selection, maybes recomputed, -, 3 = "lazy"}
        uint32_t l4[4] = {0, 0, 0, 0};
        HIPCHK(copy_sync(l4, s->lz_state.as<uint32_t>() + 2 * B + 4, 16, hipMemcpyDeviceToHost, s->st));
        h4[0] = l4[0]; h4[1] = l4[1]; h4[2] = l4[2]; h4[3] = l4[3];   // certain, maybes, most maybes of one query, (code, column) pairs re-evaluated
      }
      for (int i = 0; i < 4; ++i) g_last_s1[i] = (b0 == 0 ? 0 : g_last_s1[i]) + h4[i];
    }
    {
      const int act = post_batch(s, P, ol, replayed, capture, graph_ok, gkey, b0 == 0);
      if (act != POST_DONE) {   // a flag came down with the batch: its results are void, run it again
        eager_retry = act == POST_RETRY_EAGER;
        b0 -= maxB;
        continue;
      }
    }
    if (tr) {  // B == 1
      std::vector<int32_t> t32;
      int32_t nc = 0, ncand = 0, nr = 0;
      HIPCHK(copy_sync(&nc, s->ncells.p, 4, hipMemcpyDeviceToHost, s->st));
      HIPCHK(copy_sync(&ncand, s->ncand.p, 4, hipMemcpyDeviceToHost, s->st));
      HIPCHK(copy_sync(&nr, s->sel_cnt.p, 4, hipMemcpyDeviceToHost, s->st));
      if (tr->counts) { tr->counts[0] = nc; tr->counts[1] = ncand; tr->counts[2] = nr; }
      if (tr->S) {
        std::vector<uint16_t> hs((size_t)D.C * Qp);
        HIPCHK(copy_sync(hs.data(), s->S.p, hs.size() * 2, hipMemcpyDeviceToHost, s->st));
        for (int64_t c = 0; c < D.C; ++c)
          for (int q = 0; q < Q; ++q) tr->S[c * Q + q] = hs[(size_t)c * Qp + q];
      }
      if (tr->cells && nc > 0) {
        t32.resize((size_t)nc);
        HIPCHK(copy_sync(t32.data(), s->ucells.p, (size_t)nc * 4, hipMemcpyDeviceToHost, s->st));
        for (int i = 0; i < nc; ++i) tr->cells[i] = t32[i];
      }
      if (tr->cand && ncand > 0) {
        t32.resize((size_t)ncand);
        HIPCHK(copy_sync(t32.data(), s->cand_pid.p, (size_t)ncand * 4, hipMemcpyDeviceToHost, s->st));
        for (int i = 0; i < ncand; ++i) tr->cand[i] = (int64_t)t32[i] + D.pid_offset;
      }
      if (tr->approx && ncand > 0) HIPCHK(copy_sync(tr->approx, s->approx.p, (size_t)ncand * 4, hipMemcpyDeviceToHost, s->st));
      if (tr->rerank && nr > 0) {
        t32.resize((size_t)nr);
        HIPCHK(copy_sync(t32.data(), s->sel_pid.p, (size_t)nr * 4, hipMemcpyDeviceToHost, s->st));
        for (int i = 0; i < nr; ++i) tr->rerank[i] = (int64_t)t32[i] + D.pid_offset;
      }
      if (tr->exact && nr > 0) HIPCHK(copy_sync(tr->exact, s->exact.p, (size_t)nr * 4, hipMemcpyDeviceToHost, s->st));
    }
  }
  return FP_OK;
}

extern "C" int fp_search_shared_subset(const fp_index* index, const uint16_t* queries, int32_t n_queries, int32_t q_len, int32_t dim,
                                       const fp_search_params* params, const int64_t* subset_ids, int64_t n_subset, int64_t* out_pids,
                                       float* out_scores, int32_t* out_counts) {
  if (int rc = validate_search(index, n_queries, q_len, dim, params)) return rc;
  if (n_subset < 0 || (n_subset > 0 && !subset_ids)) return fail(FP_EINVAL, "subset");
  if (n_queries > 0 && (!queries || !out_counts || (params->top_k > 0 && (!out_pids || !out_scores))))
    return fail(FP_EINVAL, "null buffer");
  return search_impl(const_cast<fp_index*>(index), queries, n_queries, q_len, params, subset_ids, nullptr, out_pids, out_scores, out_counts, nullptr,
                     false, n_subset);
}

extern "C" int fp_search(const fp_index* index, const uint16_t* queries, int32_t n_queries, int32_t q_len, int32_t dim,
                         const fp_search_params* params, const int64_t* subset_ids, const int64_t* subset_offsets, int64_t* out_pids,
                         float* out_scores, int32_t* out_counts) {
  if (int rc = validate_search(index, n_queries, q_len, dim, params)) return rc;
  if (n_queries > 0 && (!queries || !out_counts || (params->top_k > 0 && (!out_pids || !out_scores))))
    return fail(FP_EINVAL, "null buffer");
  return search_impl(const_cast<fp_index*>(index), queries, n_queries, q_len, params, subset_ids, subset_offsets, out_pids, out_scores,
                     out_counts, nullptr);
}

static int64_t mem_info(int device_id, bool want_free) {
  size_t f = 0, t = 0;
  if (fp_set_device(device_id) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess) return -1;
  return (int64_t)(want_free ? f : t);
}
extern "C" int64_t fp_device_free_bytes(int device_id) { return mem_info(device_id, true); }
extern "C" int64_t fp_device_total_bytes(int device_id) { return mem_info(device_id, false); }

extern "C" int fp_dev_alloc(int device_id, size_t bytes, void** out) {
  if (!out) return fail(FP_EINVAL, "null argument");
  *out = nullptr;
  HIPCHK(fp_set_device(device_id));
  HIPCHK(hipMalloc(out, bytes ? bytes : 1));
  return FP_OK;
}
extern "C" int fp_dev_free(int device_id, void* p) {
  HIPCHK(fp_set_device(device_id));
  if (p) HIPCHK(hipFree(p));
  return FP_OK;
}
extern "C" int fp_dev_upload(int device_id, void* dst, const void* src, size_t bytes) {
  if (bytes && (!dst || !src)) return fail(FP_EINVAL, "null argument");
  HIPCHK(fp_set_device(device_id));
  HIPCHK(copy_sync(dst, src, bytes, hipMemcpyHostToDevice, util_stream(device_id)));
  return FP_OK;
}
extern "C" int fp_dev_download(int device_id, void* dst, const void* src, size_t bytes) {
  if (bytes && (!dst || !src)) return fail(FP_EINVAL, "null argument");
  HIPCHK(fp_set_device(device_id));
  HIPCHK(copy_sync(dst, src, bytes, hipMemcpyDeviceToHost, util_stream(device_id)));
  return FP_OK;
}

extern "C" int fp_search_device(const fp_index* index, const uint16_t* dev_queries, int32_t n_queries, int32_t q_len, int32_t dim,
                                const fp_search_params* params, int64_t* dev_out_pids, float* dev_out_scores, int32_t* dev_out_counts) {
  if (int rc = validate_search(index, n_queries, q_len, dim, params)) return rc;
  if (n_queries > 0 && (!dev_queries || !dev_out_counts || (params->top_k > 0 && (!dev_out_pids || !dev_out_scores))))
    return fail(FP_EINVAL, "null argument");
  return search_impl(const_cast<fp_index*>(index), dev_queries, n_queries, q_len, params, nullptr, nullptr, dev_out_pids, dev_out_scores,
                     dev_out_counts, nullptr, true);
}

extern "C" int fp_search_trace(const fp_index* index, const uint16_t* query, int32_t q_len, int32_t dim, const fp_search_params* params,
                               const int64_t* subset_ids, int64_t n_subset, int32_t has_subset, int64_t* out_pids, float* out_scores,
                               int32_t* out_count, uint16_t* S, int64_t* cells, int64_t* cand, float* approx, int64_t* rerank,
                               float* exact, int64_t* counts) {
  if (int rc = validate_search(index, 1, q_len, dim, params)) return rc;
  TraceOut tr{S, cells, cand, approx, rerank, exact, counts};
  if (counts) counts[0] = counts[1] = counts[2] = 0;
  int64_t off[2] = {0, n_subset};
  return search_impl(const_cast<fp_index*>(index), query, 1, q_len, params, subset_ids, has_subset ? off : nullptr, out_pids, out_scores,
                     out_count, &tr);
}

extern "C" int fp_last_search_counts(int64_t* out, int cap) {
  int n = std::min(cap, 6);
  for (int i = 0; i < n; ++i) out[i] = g_last_counts[i];
  return n;
}

extern "C" int fp_last_s1_counts(uint64_t* out, int cap) {
  const int n = std::min(cap, 5);
  for (int i = 0; i < n && i < 4; ++i) out[i] = g_last_s1[i];
  if (n > 4) out[4] = (uint64_t)(int64_t)g_last_lazy;
  return n;
}

extern "C" int fp_set_graph_replay(int enabled) { return g_graph_replay.exchange(enabled ? 1 : 0, std::memory_order_relaxed); }
extern "C" uint64_t fp_graph_replay_count(void) { return g_graph_replays.load(std::memory_order_relaxed); }

extern "C" int fp_last_search_timings(const char** names, float* ms, int cap) {
  if (!g_have_ms) return 0;
  int n = std::min(cap, (int)ST_N);
  for (int i = 0; i < n; ++i) {
    if (names) names[i] = kStageNames[i];
    if (ms) ms[i] = g_last_ms[i];
  }
  return n;
}

// ------------------------------------------------------------------------------------------
// diagnostic / test entry point: the MFMA pass of S6+S7 on given documents, without the exact-order repair
// ------------------------------------------------------------------------------------------
extern "C" int fp_maxsim_columns(const fp_index* cix, const uint16_t* query, int32_t Q, int32_t dim, const int64_t* pids, int64_t n,
                                 float* scores, uint16_t* col_max, float* unc, uint32_t* flags) {
  fp_index* ix = const_cast<fp_index*>(cix);
  if (!ix || !query || Q < 1 || (n > 0 && (!pids || !scores))) return fail(FP_EINVAL, "bad argument");
  if (dim != ix->d.dim) return fail(FP_EINVAL, "query dim does not match the index dim");
  if (n == 0) return FP_OK;
  if (n > (1 << 24)) return fail(FP_EINVAL, "too many documents");
  const FpIndexDev& D = ix->d;
  HIPCHK(fp_set_device(ix->device));
  std::vector<int32_t> loc((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    const int64_t d = pids[i] - D.pid_offset;
    if (d < 0 || d >= D.N) return fail(FP_EINVAL, "document id out of range");
    loc[(size_t)i] = (int32_t)d;
  }
  Scratch* s = acquire(ix);
  if (!s) return fail(FP_EHIP, "could not create a HIP stream");
  struct Rel { fp_index* ix; Scratch* s; ~Rel() { release(ix, s); } } rel{ix, s};
  hipStream_t st = s->st;
  fp_search_params prm{0, 4, 1, 1};
  FpSearchShape sh = make_shape(1, Q, &prm);
  const int nflag = sh.Qp / 32;
  const int32_t cnt = (int32_t)n;
  HIPCHK(s->qin.ensure((size_t)Q * dim * 2));
  HIPCHK(s->qpad.ensure((size_t)sh.Qp * dim * 2));
  HIPCHK(s->sel_pid.ensure((size_t)n * 4));
  HIPCHK(s->sel_cnt.ensure(4));
  HIPCHK(s->exact.ensure((size_t)n * 4));
  HIPCHK(s->ms_pref.ensure(16));
  HIPCHK(s->ms_cm16.ensure((size_t)n * sh.Qp * 2));
  HIPCHK(s->ms_unc.ensure((size_t)n * 4));
  HIPCHK(s->ms_uncm.ensure((size_t)n * 4));
  HIPCHK(s->ms_flags.ensure((size_t)n * nflag * 4));
  HIPCHK(hipMemcpyAsync(s->qin.p, query, (size_t)Q * dim * 2, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(s->sel_pid.p, loc.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(s->sel_cnt.p, &cnt, 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(s->ms_unc.p, 0, (size_t)n * 4, st));
  HIPCHK(hipMemsetAsync(s->ms_flags.p, 0, (size_t)n * nflag * 4, st));
  HIPCHK(hipMemsetAsync(s->ms_cm16.p, 0, (size_t)n * sh.Qp * 2, st));
  fpk_pack_queries(s->qin.as<uint16_t>(), s->qpad.as<uint16_t>(), 1, Q, sh.Qp, dim, st);
  const bool fast = fpk_maxsim_fast_shape(D.dim, D.nbits);
  FpMaxsimAux aux{fast ? s->ms_cm16.as<uint16_t>() : nullptr, s->ms_unc.as<float>(), fast ? s->ms_flags.as<uint32_t>() : nullptr, s->ms_uncm.as<float>()};
  if (fpk_maxsim(D, s->qpad.as<uint16_t>(), sh, s->sel_pid.as<int32_t>(), s->sel_cnt.as<int32_t>(), n, s->exact.as<float>(),
                 s->ms_pref.as<int64_t>(), aux, st))
    return fail(FP_EUNSUPPORTED, "dim/nbits");
  HIPCHK(hipMemcpyAsync(scores, s->exact.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  if (unc) HIPCHK(hipMemcpyAsync(unc, s->ms_unc.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  // the caller's `flags` is [n, ceil(q_len / 32)] (fastplaid.h); the device rows are Qp / 32 words wide, which is MORE for
  // 64 < q_len <= 96 (padded to 128 columns): repack like col_max below instead of copying Qp / 32 words per row
  std::vector<uint32_t> fl;
  if (flags) {
    fl.resize((size_t)n * nflag);
    HIPCHK(hipMemcpyAsync(fl.data(), s->ms_flags.p, fl.size() * 4, hipMemcpyDeviceToHost, st));
  }
  std::vector<uint16_t> cm;
  if (col_max) {
    cm.resize((size_t)n * sh.Qp);
    HIPCHK(hipMemcpyAsync(cm.data(), s->ms_cm16.p, cm.size() * 2, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  if (col_max)
    for (int64_t i = 0; i < n; ++i)
      for (int q = 0; q < Q; ++q) col_max[i * Q + q] = cm[(size_t)i * sh.Qp + q];
  if (flags) {
    const int nw = (Q + 31) / 32;
    for (int64_t i = 0; i < n; ++i)
      for (int w = 0; w < nw; ++w) flags[i * nw + w] = fl[(size_t)i * nflag + w];
  }
  return FP_OK;
}

// ------------------------------------------------------------------------------------------
// reconstruct_embeddings (embeddings.rs:12-69)
// ------------------------------------------------------------------------------------------
extern "C" int fp_reconstruct_embeddings(const fp_index* cix, const int64_t* doc_ids, int64_t n, float* out, int64_t cap_rows,
                                         int64_t* out_lengths) {
  fp_index* ix = const_cast<fp_index*>(cix);
  if (!ix || (n > 0 && (!doc_ids || !out_lengths))) return fail(FP_EINVAL, "null argument");
  HIPCHK(fp_set_device(ix->device));
  std::vector<int64_t> tok;
  for (int64_t i = 0; i < n; ++i) {
    int64_t d = doc_ids[i] - ix->d.pid_offset;
    if (d < 0 || d >= ix->d.N) return fail(FP_EINVAL, "document id out of range");
    const int64_t o = ix->h_doc_off[d], l = ix->h_doc_off[d + 1] - o;
    out_lengths[i] = l;
    const size_t base = tok.size();
    tok.resize(base + (size_t)l);
    if (ix->d.perm && l > 0) {  // stored row o+j holds original token perm[j]
      std::vector<uint16_t> pp((size_t)l);
      HIPCHK(copy_sync(pp.data(), ix->d.perm + o, (size_t)l * 2, hipMemcpyDeviceToHost, util_stream(ix->device)));
      for (int64_t j = 0; j < l; ++j) tok[base + pp[j]] = o + j;
    } else {
      for (int64_t j = 0; j < l; ++j) tok[base + j] = o + j;
    }
  }
  const int64_t rows = (int64_t)tok.size();
  if (rows > cap_rows) return fail(FP_EINVAL, "output capacity too small");
  if (rows == 0) return FP_OK;
  Scratch* s = acquire(ix);
  if (!s) return fail(FP_EHIP, "could not create a HIP stream");
  struct Rel { fp_index* ix; Scratch* s; ~Rel() { release(ix, s); } } rel{ix, s};
  HIPCHK(s->tok_idx.ensure((size_t)rows * 8));
  HIPCHK(s->recon.ensure((size_t)rows * ix->d.dim * 4));
  HIPCHK(hipMemcpyAsync(s->tok_idx.p, tok.data(), (size_t)rows * 8, hipMemcpyHostToDevice, s->st));
  fpk_reconstruct(ix->d, s->tok_idx.as<int64_t>(), rows, s->recon.as<float>(), s->st);
  HIPCHK(hipMemcpyAsync(out, s->recon.p, (size_t)rows * ix->d.dim * 4, hipMemcpyDeviceToHost, s->st));
  HIPCHK(hipStreamSynchronize(s->st));
  return FP_OK;
}

// ------------------------------------------------------------------------------------------
// token-score matrices of search hits (search.rs:294-363, :668-686)
// ------------------------------------------------------------------------------------------
extern "C" int fp_token_scores(const fp_index* cix, const uint16_t* queries, int32_t nq, int32_t Q, int32_t dim, const int64_t* pids,
                               const int32_t* counts, int64_t stride, int64_t* out_offsets, uint16_t* out, int64_t out_capacity) {
  fp_index* ix = const_cast<fp_index*>(cix);
  if (!ix || nq < 0 || (nq > 0 && (!queries || !pids || !counts)) || !out_offsets) return fail(FP_EINVAL, "null argument");
  if (dim != ix->d.dim) return fail(FP_EINVAL, "query dim does not match the index");
  if (Q < 1) return fail(FP_EINVAL, "need at least one query token");
  HIPCHK(fp_set_device(ix->device));
  std::vector<int32_t> hq, hp;
  int64_t off = 0, h = 0;
  out_offsets[0] = 0;
  for (int32_t b = 0; b < nq; ++b) {
    if (counts[b] < 0 || counts[b] > stride) return fail(FP_EINVAL, "bad hit count");
    for (int32_t i = 0; i < counts[b]; ++i) {
      const int64_t d = pids[(int64_t)b * stride + i] - ix->d.pid_offset;
      if (d < 0 || d >= ix->d.N) return fail(FP_EINVAL, "document id out of range");
      off += (int64_t)Q * (ix->h_doc_off[d + 1] - ix->h_doc_off[d]);
      out_offsets[++h] = off;
      hq.push_back(b);
      hp.push_back((int32_t)d);
    }
  }
  if (!out || h == 0 || off == 0) return FP_OK;   // sizing call, or nothing to compute
  if (off > out_capacity) return fail(FP_EINVAL, "output capacity too small");
  Scratch* s = acquire(ix);
  if (!s) return fail(FP_EHIP, "could not create a HIP stream");
  struct Rel { fp_index* ix; Scratch* s; ~Rel() { release(ix, s); } } rel{ix, s};
  const size_t qbytes = (size_t)nq * Q * dim * 2;
  HIPCHK(s->qin.ensure(qbytes));
  HIPCHK(s->tok_idx.ensure((size_t)h * 8 + (size_t)h * 8 + 64));   // hit_query | hit_pid (i32 each) + out offsets (i64)
  HIPCHK(s->tmpp.ensure((size_t)(h + 1) * 8));
  HIPCHK(s->recon.ensure((size_t)off * 2));
  int32_t* d_hq = s->tok_idx.as<int32_t>();
  int32_t* d_hp = d_hq + h;
  HIPCHK(hipMemcpyAsync(s->qin.p, queries, qbytes, hipMemcpyHostToDevice, s->st));
  HIPCHK(hipMemcpyAsync(d_hq, hq.data(), (size_t)h * 4, hipMemcpyHostToDevice, s->st));
  HIPCHK(hipMemcpyAsync(d_hp, hp.data(), (size_t)h * 4, hipMemcpyHostToDevice, s->st));
  HIPCHK(hipMemcpyAsync(s->tmpp.p, out_offsets, (size_t)(h + 1) * 8, hipMemcpyHostToDevice, s->st));
  if (fpk_token_scores(ix->d, s->qin.as<uint16_t>(), Q, d_hq, d_hp, h, s->tmpp.as<int64_t>(), s->recon.as<uint16_t>(), s->st))
    return fail(FP_EUNSUPPORTED, "q_len * dim too large for the token-score kernel");
  HIPCHK(hipMemcpyAsync(out, s->recon.p, (size_t)off * 2, hipMemcpyDeviceToHost, s->st));
  HIPCHK(hipStreamSynchronize(s->st));
  return FP_OK;
}

// ------------------------------------------------------------------------------------------
// index creation, device part (create.rs:148-184, :404-428)
// ------------------------------------------------------------------------------------------
extern "C" int fp_compress(int device_id, const uint16_t* centroids, int64_t C, int32_t dim, int32_t nbits, const uint16_t* cutoffs,
                           const uint16_t* emb, int64_t T, int64_t* out_codes, uint8_t* out_res) {
  if (!centroids || !cutoffs || C < 1 || T < 0 || (T > 0 && (!emb || !out_codes || !out_res))) return fail(FP_EINVAL, "bad argument");
  if (!(nbits == 1 || nbits == 2 || nbits == 4 || nbits == 8)) return fail(FP_EINVAL, "nbits must divide 8 (1, 2, 4 or 8)");
  if (dim < 1 || dim > 256 || (dim * nbits) % 8 != 0) return fail(FP_EUNSUPPORTED, "fp_compress: dim must be in [1, 256] with dim * nbits a multiple of 8");
  if (C >= 0x7FFFFFFFll) return fail(FP_EUNSUPPORTED, "too many centroids");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(FP_EHIP, "no HIP device available (this library has no CPU path)");
  if (device_id < 0 || device_id >= ndev) return fail(FP_EINVAL, "device index out of range");
  HIPCHK(fp_set_device(device_id));
  if (T == 0) return FP_OK;
  const int pr = dim * nbits / 8;
  const int64_t CHUNK = 1ll << 20;   // tokens per pass
  const int64_t cap = std::min<int64_t>(T, CHUNK);
  void *d_cent = nullptr, *d_cut = nullptr, *d_emb = nullptr, *d_c32 = nullptr, *d_c64 = nullptr, *d_out = nullptr, *d_work = nullptr;
  hipStream_t st = nullptr;
  int rc = FP_OK;
  float cmaxabs = 0.f;   // bounds the MFMA summation error in the assignment (fp16 -> float by hand: no device needed)
  for (int64_t i = 0; i < C * dim; ++i) {
    const uint16_t hbits = centroids[i] & 0x7FFF;
    const int ex = hbits >> 10, ma = hbits & 0x3FF;
    const float v = ex == 0 ? std::ldexp((float)ma, -24) : (ex == 31 ? INFINITY : std::ldexp((float)(ma | 0x400), ex - 25));
    if (v > cmaxabs) cmaxabs = v;
  }
  auto cleanup = [&]() {
    for (void* p : {d_cent, d_cut, d_emb, d_c32, d_c64, d_out, d_work})
      if (p) (void)hipFree(p);
    if (st) (void)hipStreamDestroy(st);
  };
#define CCHK(x)                                                                                   \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess) {                                                                       \
      cleanup();                                                                                  \
      return fail(FP_EHIP, std::string("HIP error in fp_compress: ") + hipGetErrorString(e_));    \
    }                                                                                             \
  } while (0)
  CCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CCHK(hipMalloc(&d_cent, (size_t)C * dim * 2));
  CCHK(hipMalloc(&d_cut, 1024));   // up to 255 cutoffs (nbits = 8)
  CCHK(hipMalloc(&d_emb, (size_t)cap * dim * 2));
  CCHK(hipMalloc(&d_c32, (size_t)cap * 4));
  CCHK(hipMalloc(&d_c64, (size_t)cap * 8));
  CCHK(hipMalloc(&d_out, (size_t)cap * pr));
  CCHK(hipMalloc(&d_work, fpk_compress_work_bytes(cap)));
  CCHK(hipMemcpyAsync(d_cent, centroids, (size_t)C * dim * 2, hipMemcpyHostToDevice, st));
  CCHK(hipMemcpyAsync(d_cut, cutoffs, (size_t)((1 << nbits) - 1) * 2, hipMemcpyHostToDevice, st));
  for (int64_t t0 = 0; t0 < T && rc == FP_OK; t0 += CHUNK) {
    const int64_t n = std::min<int64_t>(CHUNK, T - t0);
    CCHK(hipMemcpyAsync(d_emb, emb + t0 * dim, (size_t)n * dim * 2, hipMemcpyHostToDevice, st));
    if (fpk_compress(static_cast<const uint16_t*>(d_emb), n, static_cast<const uint16_t*>(d_cent), C, dim, nbits,
                     static_cast<const uint16_t*>(d_cut), cmaxabs, static_cast<int32_t*>(d_c32), static_cast<int64_t*>(d_c64),
                     static_cast<uint8_t*>(d_out), d_work, st)) {
      cleanup();
      return fail(FP_EUNSUPPORTED, "dim");
    }
    CCHK(hipMemcpyAsync(out_codes + t0, d_c64, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    CCHK(hipMemcpyAsync(out_res + t0 * pr, d_out, (size_t)n * pr, hipMemcpyDeviceToHost, st));
    CCHK(hipStreamSynchronize(st));
  }
#undef CCHK
  cleanup();
  return rc;
}

extern "C" int fp_assign_l2(int device_id, const uint16_t* centroids, const float* half_sqnorm, int64_t C, int32_t dim, const uint16_t* emb,
                            int64_t T, int64_t* out_labels) {
  if (!centroids || !half_sqnorm || C < 1 || T < 0 || (T > 0 && (!emb || !out_labels))) return fail(FP_EINVAL, "bad argument");
  if (dim < 1 || dim > 256) return fail(FP_EUNSUPPORTED, "fp_assign_l2: dim must be in [1, 256]");
  if (C >= 0x7FFFFFFFll) return fail(FP_EUNSUPPORTED, "too many centroids");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(FP_EHIP, "no HIP device available (this library has no CPU path)");
  if (device_id < 0 || device_id >= ndev) return fail(FP_EINVAL, "device index out of range");
  HIPCHK(fp_set_device(device_id));
  if (T == 0) return FP_OK;
  const int64_t CHUNK = 1ll << 20;
  const int64_t cap = std::min<int64_t>(T, CHUNK);
  void *d_cent = nullptr, *d_hn = nullptr, *d_emb = nullptr, *d_c32 = nullptr, *d_c64 = nullptr;
  hipStream_t st = nullptr;
  auto cleanup = [&]() {
    for (void* p : {d_cent, d_hn, d_emb, d_c32, d_c64})
      if (p) (void)hipFree(p);
    if (st) (void)hipStreamDestroy(st);
  };
#define ACHK(x)                                                                                   \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess) {                                                                       \
      cleanup();                                                                                  \
      return fail(FP_EHIP, std::string("HIP error in fp_assign_l2: ") + hipGetErrorString(e_));   \
    }                                                                                             \
  } while (0)
  ACHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  ACHK(hipMalloc(&d_cent, (size_t)C * dim * 2));
  ACHK(hipMalloc(&d_hn, (size_t)C * 4));
  ACHK(hipMalloc(&d_emb, (size_t)cap * dim * 2));
  ACHK(hipMalloc(&d_c32, (size_t)cap * 4));
  ACHK(hipMalloc(&d_c64, (size_t)cap * 8));
  ACHK(hipMemcpyAsync(d_cent, centroids, (size_t)C * dim * 2, hipMemcpyHostToDevice, st));
  ACHK(hipMemcpyAsync(d_hn, half_sqnorm, (size_t)C * 4, hipMemcpyHostToDevice, st));
  for (int64_t t0 = 0; t0 < T; t0 += CHUNK) {
    const int64_t n = std::min<int64_t>(CHUNK, T - t0);
    ACHK(hipMemcpyAsync(d_emb, emb + t0 * dim, (size_t)n * dim * 2, hipMemcpyHostToDevice, st));
    if (fpk_assign_l2(static_cast<const uint16_t*>(d_emb), n, static_cast<const uint16_t*>(d_cent), static_cast<const float*>(d_hn), C, dim,
                      static_cast<int32_t*>(d_c32), static_cast<int64_t*>(d_c64), st)) {
      cleanup();
      return fail(FP_EUNSUPPORTED, "dim");
    }
    ACHK(hipMemcpyAsync(out_labels + t0, d_c64, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    ACHK(hipStreamSynchronize(st));
  }
#undef ACHK
  cleanup();
  return FP_OK;
}

// ------------------------------------------------------------------------------------------
// document-sharded search (see fastplaid.h)
// ------------------------------------------------------------------------------------------
struct fp_shard_ctx {
  fp_index* ix;
  Scratch* s;
  Pipe P;
  fp_search_params params;
  int B, Q;
  bool empty_all;
};

extern "C" int fp_shard_begin(const fp_index* cix, const uint16_t* queries, int32_t nq, int32_t Q, int32_t dim,
                              const fp_search_params* p, fp_shard_ctx** out) {
  if (!out) return fail(FP_EINVAL, "null argument");
  *out = nullptr;
  if (int rc = validate_search(cix, nq, Q, dim, p)) return rc;
  fp_index* ix = const_cast<fp_index*>(cix);
  if (nq < 1) return fail(FP_EINVAL, "need at least one query");
  const int64_t R = std::max<int64_t>(p->n_full_scores / 4, 1);
  HIPCHK(fp_set_device(ix->device));
  Scratch* s = acquire(ix);
  if (!s) return fail(FP_EHIP, "could not create a HIP stream");
  fp_shard_ctx* c = new fp_shard_ctx();
  c->ix = ix; c->s = s; c->params = *p; c->B = nq; c->Q = Q;
  c->P = Pipe{};
  c->P.ix = ix; c->P.s = s; c->P.sh = make_shape(nq, Q, p);
  c->empty_all = (p->n_ivf_probe > ix->d.C) || p->top_k == 0 || ix->d.N == 0;
  (void)R;
  hipError_t e = s->qin.ensure((size_t)nq * Q * ix->d.dim * 2);
  if (e == hipSuccess) e = hipMemcpyAsync(s->qin.p, queries, (size_t)nq * Q * ix->d.dim * 2, hipMemcpyHostToDevice, s->st);
  if (e != hipSuccess) {
    release(ix, s);
    delete c;
    return fail(FP_EHIP, hipGetErrorString(e));
  }
  *out = c;
  return FP_OK;
}

extern "C" int64_t fp_shard_R(const fp_shard_ctx* c) { return c ? c->P.sh.R : 0; }

extern "C" void fp_shard_end(fp_shard_ctx* c) {
  if (!c) return;
  (void)hipStreamSynchronize(c->s->st);
  release(c->ix, c->s);
  delete c;
}

extern "C" int fp_shard_stage1(fp_shard_ctx* c, void* dev_rec1) {
  if (!c || !dev_rec1) return fail(FP_EINVAL, "null argument");
  HIPCHK(fp_set_device(c->ix->device));
  Scratch* s = c->s;
  const int64_t R = c->P.sh.R;
  const int B = c->B;
  if (c->empty_all) {
    HIPCHK(s->sel_cnt.ensure((size_t)B * 4));
    HIPCHK(s->sel_pid.ensure((size_t)B * R * 4));
    HIPCHK(s->sel_approx.ensure((size_t)B * R * 4));
    HIPCHK(hipMemsetAsync(s->sel_cnt.p, 0, (size_t)B * 4, s->st));
  } else {
    s->fold_stats = false;
    if (int rc = run_front(c->P, nullptr, nullptr, false)) return rc;
  }
  fpk_shard_pack1(s->sel_approx.as<float>(), s->sel_pid.as<int32_t>(), s->sel_cnt.as<int32_t>(), B, R, c->ix->d.pid_offset, dev_rec1, s->st);
  LAUNCHCHK("fp_shard_stage1");
  HIPCHK(hipStreamSynchronize(s->st));
  return FP_OK;
}

extern "C" int fp_shard_stage2(fp_shard_ctx* c, const void* dev_all_rec1, int32_t G, void* dev_rec2) {
  if (!c || !dev_all_rec1 || !dev_rec2 || G < 1) return fail(FP_EINVAL, "bad argument");
  HIPCHK(fp_set_device(c->ix->device));
  Scratch* s = c->s;
  const FpIndexDev& D = c->ix->d;
  const int64_t R = c->P.sh.R;
  const int B = c->B;
  if (fpk_shard_global_cut(dev_all_rec1, G, B, R, D.pid_offset, D.pid_offset + D.N, s->sel_pid.as<int32_t>(), s->sel_cnt.as<int32_t>(), s->st))
    return fail(FP_EUNSUPPORTED, "n_ranks * max(n_full_scores/4, 1) is too large for the LDS cut of the sharded search (limit 16384 entries)");
  HIPCHK(s->exact.ensure((size_t)B * R * 4));
  if (!c->empty_all) {
    // MFMA scores + uncertainties (what the unsharded search ranks by); the near-tied documents are repaired in stage 3, once
    // the union of all ranks' survivors is known
    if (int rc = run_maxsim(c->ix, s, c->P.sh, R, c->params.top_k, /*mode*/ 0)) return rc;
    fpk_shard_pack2(s->exact.as<float>(), s->ms_repairable ? s->ms_unc.as<float>() : nullptr, s->ms_repairable ? s->ms_uncm.as<float>() : nullptr,
                    s->sel_pid.as<int32_t>(), s->sel_cnt.as<int32_t>(), B, R,
                    D.pid_offset, dev_rec2, s->st);
  } else {
    fpk_shard_pack2(s->exact.as<float>(), nullptr, nullptr, s->sel_pid.as<int32_t>(), s->sel_cnt.as<int32_t>(), B, R, D.pid_offset, dev_rec2, s->st);
  }
  LAUNCHCHK("fp_shard_stage2");
  HIPCHK(hipStreamSynchronize(s->st));
  return FP_OK;
}

// whether the third exchange of the sharded search happens, as a pure function of the environment and R (every rank must agree
// without looking at what its own stages did): 0 no repair at all, 1 near-tied documents, 2 every flagged document (also when R
// is beyond the LDS of the marking kernel)
static int shard_marks_mode(int64_t R) {
  static const int repair_env = [] { const char* e = getenv("FP_MAXSIM_REPAIR"); return e ? atoi(e) : 1; }();
  if (repair_env == 0) return 0;
  int np2 = 2;
  while (np2 < R) np2 <<= 1;
  return (repair_env != 2 && np2 <= 8192) ? 1 : 2;
}
// union + near-tie marking (identical on every rank) + repair of the marked documents this rank holds -> dev_x [B][R] f32
static int shard_mark_and_repair(fp_index* ix, Scratch* s, const FpSearchShape& sh, const void* all_rec2, int G, int rank, bool empty_local,
                                 int64_t K, float* x, hipStream_t st) {
  const int B = sh.B;
  const int64_t R = sh.R;
  s->sh_marks_mode = 0;   // (set again below; a failure before that point must not leave the previous call's value behind)
  HIPCHK(s->tmpp.ensure((size_t)B * R * 8));
  HIPCHK(s->tmpf.ensure((size_t)4 * B * R * 4));
  HIPCHK(s->u_cnt.ensure((size_t)B * 4));
  HIPCHK(s->ms_marks.ensure((size_t)B * R * 4));
  HIPCHK(s->ms_nmark.ensure((size_t)B * 4));
  HIPCHK(s->sh_lmarks.ensure((size_t)B * R * 4));
  HIPCHK(s->sh_lnmark.ensure((size_t)B * 4));
  float* u_score = s->tmpf.as<float>();
  int32_t* u_src = reinterpret_cast<int32_t*>(u_score + (size_t)B * R);
  float* u_unc = u_score + (size_t)2 * B * R;
  float* u_uncm = u_score + (size_t)3 * B * R;
  if (fpk_shard_union(all_rec2, G, B, R, s->tmpp.as<int64_t>(), u_score, u_src, u_unc, u_uncm, s->u_cnt.as<int32_t>(), st))
    return fail(FP_EUNSUPPORTED, "n_ranks * max(n_full_scores/4, 1) is too large for the LDS merge of the sharded search (limit 16384 entries)");
  static const int repair_env = [] { const char* e = getenv("FP_MAXSIM_REPAIR"); return e ? atoi(e) : 1; }();
  s->sh_marks_mode = 0;   // 0 nothing to exchange, 1 near-tied documents (marks), 2 every flagged document
  if (repair_env != 0) {
    s->sh_marks_mode = (repair_env != 2 && fpk_final_mark(u_score, u_unc, u_uncm, s->u_cnt.as<int32_t>(), R, B, K, s->ms_marks.as<int32_t>(),
                                                          s->ms_nmark.as<int32_t>(), st) == 0) ? 1 : 2;
    const int32_t* marks = s->sh_marks_mode == 1 ? s->ms_marks.as<int32_t>() : nullptr;
    fpk_shard_local_marks(marks, s->ms_nmark.as<int32_t>(), u_unc, s->u_cnt.as<int32_t>(), u_src, B, R, rank, s->sh_lmarks.as<int32_t>(),
                          s->sh_lnmark.as<int32_t>(), st);
    if (!empty_local && s->ms_repairable) {
      FpMaxsimAux aux{s->ms_cm16.as<uint16_t>(), s->ms_unc.as<float>(), s->ms_flags.as<uint32_t>(), s->ms_uncm.as<float>()};
      fpk_maxsim_repair(ix->d, s->qpad.as<uint16_t>(), sh, s->sel_pid.as<int32_t>(), s->sel_cnt.as<int32_t>(), R, s->sh_lmarks.as<int32_t>(),
                        s->sh_lnmark.as<int32_t>(), s->exact.as<float>(), aux, st);
    }
    fpk_shard_pack3(marks, s->ms_nmark.as<int32_t>(), u_unc, s->u_cnt.as<int32_t>(), u_src, B, R, rank, s->exact.as<float>(), x, st);
  }
  return FP_OK;
}

static int shard_apply_and_rank(Scratch* s, const FpSearchShape& sh, const float* xall, int64_t K, int64_t* out_pids, float* out_scores,
                                int32_t* out_counts, hipStream_t st, int64_t xstride = 0) {
  const int B = sh.B;
  const int64_t R = sh.R;
  float* u_score = s->tmpf.as<float>();
  int32_t* u_src = reinterpret_cast<int32_t*>(u_score + (size_t)B * R);
  float* u_unc = u_score + (size_t)2 * B * R;
  if (s->sh_marks_mode != 0)
    fpk_shard_apply3(s->sh_marks_mode == 1 ? s->ms_marks.as<int32_t>() : nullptr, s->ms_nmark.as<int32_t>(), u_unc, s->u_cnt.as<int32_t>(), u_src, B,
                     R, xall, u_score, st, xstride);
  OutLayout ol(B, K);
  HIPCHK(s->out_all.ensure(ol.total));
  HIPCHK(s->h_out.ensure(ol.total));
  char* od = s->out_all.as<char>();
  if (const int trc = fpk_final_topk(u_score, nullptr, s->tmpp.as<int64_t>(), s->u_cnt.as<int32_t>(), R, B, K, 0, reinterpret_cast<int64_t*>(od),
                                     reinterpret_cast<float*>(od + ol.score_off), reinterpret_cast<int32_t*>(od + ol.cnt_off), st))
    return trc < 0 ? fail(FP_EUNSUPPORTED, "n_queries * n_ranks * max(n_full_scores / 4, 1) >= 2^31 in the final ranking of the union")
                   : fail(FP_EHIP, "final ranking of the union failed (hip error " + std::to_string(trc) + ")");
  LAUNCHCHK("sharded merge");
  HIPCHK(hipMemcpyAsync(s->h_out.p, od, ol.total, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  ol.scatter(s->h_out.p, out_pids, out_scores, out_counts);
  return FP_OK;
}

extern "C" int fp_shard_stage3(fp_shard_ctx* c, const void* dev_all_rec2, int32_t G, int32_t rank, void* dev_x) {
  if (!c || !dev_all_rec2 || !dev_x || G < 1 || rank < 0 || rank >= G) return fail(FP_EINVAL, "bad argument");
  HIPCHK(fp_set_device(c->ix->device));
  Scratch* s = c->s;
  if (c->params.top_k == 0) return FP_OK;
  if (int rc = shard_mark_and_repair(c->ix, s, c->P.sh, dev_all_rec2, G, rank, c->empty_all, c->params.top_k, static_cast<float*>(dev_x), s->st)) return rc;
  LAUNCHCHK("fp_shard_stage3");
  HIPCHK(hipStreamSynchronize(s->st));
  return FP_OK;
}

extern "C" int fp_shard_stage4(fp_shard_ctx* c, const void* dev_all_x, int32_t G, int64_t* out_pids, float* out_scores, int32_t* out_counts) {
  if (!c || !dev_all_x || G < 1 || !out_counts) return fail(FP_EINVAL, "bad argument");
  HIPCHK(fp_set_device(c->ix->device));
  for (int i = 0; i < c->B; ++i) out_counts[i] = 0;
  if (c->params.top_k == 0) return FP_OK;
  return shard_apply_and_rank(c->s, c->P.sh, static_cast<const float*>(dev_all_x), c->params.top_k, out_pids, out_scores, out_counts, c->s->st);
}

// ------------------------------------------------------------------------------------------
// document-sharded search with the collectives issued by the library itself (RCCL over xGMI): the three all-gathers go on the
// search stream right behind the kernels that fill their send buffers -- no host synchronisation between the stages, no
// framework in the data path.  librccl is bound at run time (dlopen), so single-GPU users never need it.
// ------------------------------------------------------------------------------------------
#include <dlfcn.h>
struct FpNcclUid { char internal[128]; };   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
struct RcclApi {
  void* h = nullptr;
  int (*GetUniqueId)(FpNcclUid*) = nullptr;
  int (*CommInitRank)(void**, int, FpNcclUid, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommAbort)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string why;
};
static RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* env = getenv("FP_RCCL_LIB");
    const char* names[] = {env ? env : "librccl.so.1", "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    // a process that already carries an RCCL (torch bundles one) must keep using THAT copy: look for a loaded one first
    for (const char* n : names)
      if (!api.h) api.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)
      if (!api.h) api.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!api.h) { api.why = std::string("librccl not found (") + (dlerror() ? dlerror() : "dlopen failed") + "); set FP_RCCL_LIB"; return; }
    api.GetUniqueId = reinterpret_cast<int (*)(FpNcclUid*)>(dlsym(api.h, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<int (*)(void**, int, FpNcclUid, int)>(dlsym(api.h, "ncclCommInitRank"));
    api.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(api.h, "ncclAllGather"));
    api.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(api.h, "ncclCommDestroy"));
    api.CommAbort = reinterpret_cast<int (*)(void*)>(dlsym(api.h, "ncclCommAbort"));
    api.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(api.h, "ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy) { api.why = "librccl lacks the expected symbols"; api.h = nullptr; }
  });
  return api.h ? &api : nullptr;
}
static int rccl_fail(const char* what, int rc) {
  RcclApi* a = rccl_api();
  return fail(FP_EHIP, std::string("RCCL: ") + what + " failed: " + ((a && a->GetErrorString) ? a->GetErrorString(rc) : "error ") + " (" + std::to_string(rc) + ")");
}

struct fp_comm {
  void* comm = nullptr;
  int device = 0, n_ranks = 1, rank = 0;
};

extern "C" int fp_comm_unique_id(void* out_id_128) {
  if (!out_id_128) return fail(FP_EINVAL, "null argument");
  RcclApi* a = rccl_api();
  if (!a) return fail(FP_EUNSUPPORTED, "RCCL unavailable: librccl could not be loaded (set FP_RCCL_LIB)");
  FpNcclUid id;
  if (int rc = a->GetUniqueId(&id)) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(out_id_128, &id, sizeof(id));
  return FP_OK;
}

extern "C" int fp_comm_create(int device_id, int n_ranks, int rank, const void* unique_id_128, fp_comm** out) {
  if (!out || !unique_id_128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(FP_EINVAL, "bad argument");
  *out = nullptr;
  RcclApi* a = rccl_api();
  if (!a) return fail(FP_EUNSUPPORTED, "RCCL unavailable: librccl could not be loaded (set FP_RCCL_LIB)");
  HIPCHK(fp_set_device(device_id));
  FpNcclUid id;
  memcpy(&id, unique_id_128, sizeof(id));
  fp_comm* c = new fp_comm();
  c->device = device_id; c->n_ranks = n_ranks; c->rank = rank;
  if (int rc = a->CommInitRank(&c->comm, n_ranks, id, rank)) { delete c; return rccl_fail("ncclCommInitRank", rc); }
  *out = c;
  return FP_OK;
}

extern "C" void fp_comm_destroy(fp_comm* c) {
  if (!c) return;
  RcclApi* a = rccl_api();
  if (a && c->comm) { (void)fp_set_device(c->device); (void)a->CommDestroy(c->comm); }
  delete c;
}
extern "C" int fp_comm_n_ranks(const fp_comm* c) { return c ? c->n_ranks : 0; }
extern "C" int fp_comm_rank(const fp_comm* c) { return c ? c->rank : -1; }

// One batch (<= the S budget) of the document-sharded search.  COLLECTIVE SAFETY: between the first and the last all-gather
// nothing returns.  Every buffer the protocol itself needs is allocated before the first collective; a stage that fails locally
// (a kernel launch, an allocation inside the front half, ...) only records its error, empties this rank's contribution, and
// raises bit 1 of the status word that travels in record 0 of the rank's block of the next exchange (bit 0: the learnt candidate
// capacity overflowed).  Every rank reads the OR of all status words after its final sync: any failure -> every rank returns an
// error for the batch; an overflow -> every rank runs the batch again, waiting for its candidate total this time.
enum { SH_OVERFLOW = 1, SH_FAILED = 2 };
static int shard_search_batch(fp_index* ix, Scratch* s, fp_comm* comm, RcclApi* api, const uint16_t* queries, int B, int32_t Q,
                              const fp_search_params* p, int64_t* out_pids, float* out_scores, int32_t* out_counts) {
  const FpIndexDev& D = ix->d;
  const int G = comm->n_ranks;
  hipStream_t st = s->st;
  const int64_t K = p->top_k;
  for (int attempt = 0; attempt < 2; ++attempt) {
    Pipe P{};
    P.ix = ix; P.s = s; P.sh = make_shape(B, Q, p);
    P.allow_spec = attempt == 0;
    const int64_t R = P.sh.R;
    const int64_t xstride = (int64_t)B * R + 16;   // floats per rank in the third exchange: [B][R] scores + a status tail
    // every rank must issue every collective whatever its shard holds (an empty shard contributes only padding records)
    const bool empty_local = (p->n_ivf_probe > D.C) || D.N == 0;
    // ---- everything the protocol needs, before the first collective (a failure here leaves no peer inside a collective of
    // THIS call; the communicator is aborted so that peers already waiting for this rank get an error instead of a hang) ----
    int status = 0;
    std::string first_err;
    // a failure that makes THIS rank leave the protocol (a collective that could not be enqueued, a device error behind the last
    // exchange): the communicator is aborted, so that the peers -- inside this batch's collectives or about to enter the next
    // sub-batch's -- get an error from RCCL instead of waiting for a rank that has returned
    auto leave = [&](int rc) {
      if (api->CommAbort && comm->comm) { (void)api->CommAbort(comm->comm); comm->comm = nullptr; }
      return rc;
    };
    auto note = [&](int rc) {   // a local failure: remember the first message, keep going with an empty contribution
      if (rc != FP_OK && !(status & SH_FAILED)) { status |= SH_FAILED; first_err = g_err; }
    };
    {
      hipError_t e = hipSuccess;
      auto need = [&](hipError_t r) { if (e == hipSuccess) e = r; };
      need(s->qin.ensure((size_t)B * Q * D.dim * 2));
      need(s->sh_rec.ensure((size_t)B * R * 24));
      need(s->sh_all.ensure((size_t)G * B * R * 24));
      need(s->sh_x.ensure((size_t)xstride * 4));
      need(s->sh_xall.ensure((size_t)G * xstride * 4));
      need(s->sel_cnt.ensure((size_t)B * 4));
      need(s->sel_pid.ensure((size_t)B * R * 4));
      need(s->sel_approx.ensure((size_t)B * R * 4));
      need(s->exact.ensure((size_t)B * R * 4));
      need(s->h_small.ensure(4096));
      need(s->spec_total.ensure(32));
      need(s->tmpp.ensure((size_t)B * R * 8));
      need(s->tmpf.ensure((size_t)4 * B * R * 4));
      need(s->u_cnt.ensure((size_t)B * 4));
      need(s->ms_marks.ensure((size_t)B * R * 4));
      need(s->ms_nmark.ensure((size_t)B * 4));
      need(s->sh_lmarks.ensure((size_t)B * R * 4));
      need(s->sh_lnmark.ensure((size_t)B * 4));
      OutLayout ol(B, K);
      need(s->out_all.ensure(ol.total));
      need(s->h_out.ensure(ol.total));
      if (e == hipSuccess) e = hipMemcpyAsync(s->qin.p, queries, (size_t)B * Q * D.dim * 2, hipMemcpyHostToDevice, st);
      if (e != hipSuccess) {
        if (api->CommAbort && comm->comm) { (void)api->CommAbort(comm->comm); comm->comm = nullptr; }
        return fail(FP_EHIP, std::string("HIP error: ") + hipGetErrorString(e) + " (sharded search buffers; the communicator was aborted)");
      }
    }
    int32_t* d_flags = s->spec_total.as<int32_t>() + 2;   // (behind the 8 bytes of the candidate total)
    (void)hipMemsetAsync(d_flags, 0, 4, st);
    static const int fail_at = (int)fp_test_opt("shard_fail_at", 0);   // testing: a local failure in stage n
    // ---- front half (S1..S5 on the shard) ----
    bool have_front = false;
    if (fail_at == 1) note(fail(FP_EHIP, "injected failure in the front half (FP_SHARD_FAIL_AT=1)"));
    else if (!empty_local) {
      s->fold_stats = false;
      const int rc = run_front(P, nullptr, nullptr, false);
      note(rc);
      have_front = rc == FP_OK;
    }
    if (!have_front) (void)hipMemsetAsync(s->sel_cnt.p, 0, (size_t)B * 4, st);   // nothing to offer: padding records only
    // exchange 1: local top-R by approximate score
    fpk_shard_pack1(s->sel_approx.as<float>(), s->sel_pid.as<int32_t>(), s->sel_cnt.as<int32_t>(), B, R, D.pid_offset, s->sh_rec.p, st,
                    (have_front && P.spec) ? s->spec_total.as<int64_t>() : nullptr, (have_front && P.spec) ? P.M : 0, status);
    if (int rc = api->AllGather(s->sh_rec.p, s->sh_all.p, (size_t)B * R * 16, /*ncclInt8*/ 0, comm->comm, st)) return leave(rccl_fail("ncclAllGather", rc));
    fpk_shard_status(s->sh_all.p, G, (int64_t)B * R * 16, 12, d_flags, st);
    // global cut on the union (sort in LDS, or select + ordered compaction when n_ranks * R is beyond it)
    if (fpk_shard_global_cut(s->sh_all.p, G, B, R, D.pid_offset, D.pid_offset + D.N, s->sel_pid.as<int32_t>(), s->sel_cnt.as<int32_t>(), st)) {
      note(fail(FP_EUNSUPPORTED, "n_ranks * max(n_full_scores/4, 1) does not fit a 32-bit union index"));
      (void)hipMemsetAsync(s->sel_cnt.p, 0, (size_t)B * 4, st);
    }
    // exchange 2: MFMA scores + uncertainties of the local survivors
    bool have_scores = false;
    if (fail_at == 2) note(fail(FP_EHIP, "injected failure before the MaxSim pass (FP_SHARD_FAIL_AT=2)"));
    if (have_front && !(status & SH_FAILED)) {
      const int rc = run_maxsim(ix, s, P.sh, R, K, 0);
      note(rc);
      have_scores = rc == FP_OK;
    }
    if (!have_scores) (void)hipMemsetAsync(s->sel_cnt.p, 0, (size_t)B * 4, st);
    fpk_shard_pack2(s->exact.as<float>(), (have_scores && s->ms_repairable) ? s->ms_unc.as<float>() : nullptr,
                    (have_scores && s->ms_repairable) ? s->ms_uncm.as<float>() : nullptr, s->sel_pid.as<int32_t>(), s->sel_cnt.as<int32_t>(), B, R,
                    D.pid_offset, s->sh_rec.p, st, status);
    if (int rc = api->AllGather(s->sh_rec.p, s->sh_all.p, (size_t)B * R * 24, 0, comm->comm, st)) return leave(rccl_fail("ncclAllGather", rc));
    fpk_shard_status(s->sh_all.p, G, (int64_t)B * R * 24, 20, d_flags, st);
    // exchange 3: the union and its near-tie marking are identical on every rank; each rank repairs the marked documents it holds
    // and ships their scores ([B][R] floats by union position)
    (void)hipMemsetAsync(s->sh_x.p, 0, (size_t)xstride * 4, st);
    if (fail_at == 3) note(fail(FP_EHIP, "injected failure before the marking (FP_SHARD_FAIL_AT=3)"));
    note(shard_mark_and_repair(ix, s, P.sh, s->sh_all.p, G, comm->rank, !have_scores, K, s->sh_x.as<float>(), st));
    if (shard_marks_mode(R) != 0) {   // (the same on every rank whatever happened locally: a function of the environment and R)
      (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(s->sh_x.as<int32_t>() + (int64_t)B * R), status, 1, st);
      if (int rc = api->AllGather(s->sh_x.p, s->sh_xall.p, (size_t)xstride * 4, 0, comm->comm, st)) return leave(rccl_fail("ncclAllGather", rc));
      fpk_shard_status(s->sh_xall.p, G, xstride * 4, (int64_t)B * R * 4, d_flags, st);
    }
    // ---- past the last collective: the merge, the download, and the verdict of all ranks ----
    (void)hipMemcpyAsync(static_cast<char*>(s->h_small.p) + 40, d_flags, 4, hipMemcpyDeviceToHost, st);
    const int mrc = (status & SH_FAILED) ? FP_OK : shard_apply_and_rank(s, P.sh, s->sh_xall.as<float>(), K, out_pids, out_scores, out_counts, st, xstride);
    if (hipError_t se = hipStreamSynchronize(st); se != hipSuccess)
      return leave(fail(FP_EHIP, std::string("HIP error: ") + hipGetErrorString(se) + " (sharded search; the communicator was aborted)"));
    const bool late_err = [&] {   // an error that surfaced behind the last exchange: the peers have not heard of it
      if (hipError_t le = hipGetLastError(); le != hipSuccess) { (void)fail(FP_EHIP, std::string("HIP error: ") + hipGetErrorString(le) + " (sharded search)"); return true; }
      return false;
    }();
    const int flags = *reinterpret_cast<const int32_t*>(static_cast<const char*>(s->h_small.p) + 40);
    if (status & SH_FAILED) return fail(FP_EHIP, "fp_shard_search failed on this rank (every rank reports the batch as failed): " + first_err);
    if (flags & SH_FAILED) {
      for (int i = 0; i < B; ++i) out_counts[i] = 0;
      return fail(FP_EHIP, "fp_shard_search: another rank failed during this batch (its error is reported there); no rank returns results");
    }
    // every rank saw the same status words: an overflow anywhere -> EVERY rank goes round again, whatever its own merge did with
    // the discarded attempt (a rank that returned here instead would leave the others alone in attempt 1's collectives)
    if (attempt == 0 && (flags & SH_OVERFLOW)) {
      s->spec_cap = 0;   // (the rank that overflowed; harmless on the others: the second attempt waits everywhere)
      continue;
    }
    // a failure of this rank's own merge / ranking behind the last exchange is not in any status word: the other ranks return
    // results and move on to the next sub-batch's collectives, so this rank aborts the communicator on its way out
    if (mrc) return leave(mrc);
    if (late_err) return leave(FP_EHIP);
    if (have_front) learn_capacity(s, *reinterpret_cast<const int64_t*>(s->h_small.p));
    return FP_OK;
  }
  return fail(FP_EHIP, "fp_shard_search: the candidate capacity overflowed twice");
}

extern "C" int fp_shard_search(const fp_index* cix, fp_comm* comm, const uint16_t* queries, int32_t nq, int32_t Q, int32_t dim,
                               const fp_search_params* p, int64_t* out_pids, float* out_scores, int32_t* out_counts) {
  if (!comm || !comm->comm) return fail(FP_EINVAL, "null (or aborted) communicator");
  if (int rc = validate_search(cix, nq, Q, dim, p)) return rc;
  if (nq < 1 || !queries || !out_counts || (p->top_k > 0 && (!out_pids || !out_scores))) return fail(FP_EINVAL, "bad argument");
  RcclApi* api = rccl_api();
  if (!api) return fail(FP_EUNSUPPORTED, "RCCL unavailable");
  fp_index* ix = const_cast<fp_index*>(cix);
  if (ix->device != comm->device) return fail(FP_EINVAL, "index and communicator live on different devices");
  HIPCHK(fp_set_device(ix->device));
  for (int i = 0; i < nq; ++i) out_counts[i] = 0;
  if (p->top_k == 0) return FP_OK;
  Scratch* s = acquire(ix);
  if (!s) return fail(FP_EHIP, "could not create a HIP stream");
  struct Rel { fp_index* ix; Scratch* s; ~Rel() { (void)hipStreamSynchronize(s->st); release(ix, s); } } rel{ix, s};
  // Sub-batches against a FIXED budget of the centroid-score table (every rank must split the batch identically, so the budget
  // may not depend on a rank's free memory as fp_search's does): 16 GiB, FP_SHARD_S_BUDGET_KB for tests
  static const int64_t budget = [] { const double kb = fp_test_opt("shard_s_budget_kb", 0); return kb > 0 ? (int64_t)kb * 1024 : (16ll << 30); }();
  const int Qp = fp_padded_qlen(Q);
  const int64_t per_query = (int64_t)ix->d.C * Qp * 2;
  const int maxB = (int)std::max<int64_t>(1, std::min<int64_t>(nq, budget / std::max<int64_t>(per_query, 1)));
  const int64_t K = p->top_k;
  for (int b0 = 0; b0 < nq; b0 += maxB) {
    const int B = std::min(maxB, nq - b0);
    if (int rc = shard_search_batch(ix, s, comm, api, queries + (size_t)b0 * Q * dim, B, Q, p, out_pids ? out_pids + (size_t)b0 * K : nullptr,
                                    out_scores ? out_scores + (size_t)b0 * K : nullptr, out_counts + b0))
      return rc;
  }
  return FP_OK;
}


// ------------------------------------------------------------------------------------------
// exhaustive arithmetic self-test (see fp_kernels.hip k_selftest_arith)
// ------------------------------------------------------------------------------------------
extern "C" int fp_selftest_arith(int device_id, uint64_t* out_mismatches /*[16]*/) {
  if (!out_mismatches) return fail(FP_EINVAL, "null argument");
  HIPCHK(fp_set_device(device_id));
  unsigned long long* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, 128));
  hipStream_t us = util_stream(device_id);
  HIPCHK(hipMemsetAsync(d, 0, 128, us));
  fpk_selftest_arith(d, us);
  hipError_t e = copy_sync(out_mismatches, d, 128, hipMemcpyDeviceToHost, us);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(FP_EHIP, hipGetErrorString(e));
  return FP_OK;
}

// ------------------------------------------------------------------------------------------
// export the index arrays back to the host in the reference's construct_index layout
// ------------------------------------------------------------------------------------------
extern "C" int64_t fp_index_ivf_total(const fp_index* ix) {
  if (!ix || !ix->has_ivf) return 0;
  (void)fp_set_device(ix->device);
  int64_t tot = 0;
  if (copy_sync(&tot, ix->d.ivf_off + ix->d.P, 8, hipMemcpyDeviceToHost, util_stream(ix->device)) != hipSuccess) return fail(FP_EHIP, "memcpy");
  return tot;
}

extern "C" int fp_index_export(const fp_index* ix, int64_t* doc_codes, uint8_t* doc_residuals, int64_t* doc_lengths, int64_t* ivf,
                               int32_t* ivf_lengths) {
  if (!ix) return fail(FP_EINVAL, "null argument");
  HIPCHK(fp_set_device(ix->device));
  const FpIndexDev& D = ix->d;
  if (doc_lengths)
    for (int64_t i = 0; i < D.N; ++i) doc_lengths[i] = ix->h_doc_off[i + 1] - ix->h_doc_off[i];
  const int64_t chunk = 64ll << 20;
  std::vector<int32_t> tmp;
  std::vector<uint16_t> hperm;
  if (D.perm && D.T > 0 && (doc_codes || doc_residuals)) {
    hperm.resize((size_t)D.T);
    HIPCHK(copy_sync(hperm.data(), D.perm, (size_t)D.T * 2, hipMemcpyDeviceToHost, util_stream(ix->device)));
  }
  // stored token i of document d is original token perm[i]: write it back to its original slot
  auto orig_row = [&](int64_t doc, int64_t i) { return hperm.empty() ? i : ix->h_doc_off[doc] + hperm[i]; };
  if (doc_codes) {
    tmp.resize((size_t)std::max<int64_t>(D.T, 1));
    if (D.T > 0) HIPCHK(copy_sync(tmp.data(), D.codes, (size_t)D.T * 4, hipMemcpyDeviceToHost, util_stream(ix->device)));
    for (int64_t d0 = 0; d0 < D.N; ++d0)
      for (int64_t i = ix->h_doc_off[d0]; i < ix->h_doc_off[d0 + 1]; ++i) doc_codes[orig_row(d0, i)] = tmp[i];
  }
  if (doc_residuals && D.T > 0) {
    if (hperm.empty() && !D.resid_native) {
      HIPCHK(copy_sync(doc_residuals, D.residuals, (size_t)D.T * D.pr, hipMemcpyDeviceToHost, util_stream(ix->device)));
    } else {
      std::vector<uint8_t> rt((size_t)std::min<int64_t>(chunk, D.T) * D.pr);
      int64_t d0 = 0;
      for (int64_t s0 = 0; s0 < D.T; s0 += chunk) {
        const int64_t m = std::min(chunk, D.T - s0);
        HIPCHK(copy_sync(rt.data(), D.residuals + s0 * D.pr, (size_t)m * D.pr, hipMemcpyDeviceToHost, util_stream(ix->device)));
        for (int64_t i = s0; i < s0 + m; ++i) {
          while (ix->h_doc_off[d0 + 1] <= i) ++d0;
          row_to_reference_order(ix, rt.data() + (i - s0) * D.pr, doc_residuals + orig_row(d0, i) * D.pr);
        }
      }
    }
  }
  if (ix->has_ivf && (ivf || ivf_lengths)) {
    std::vector<int64_t> off((size_t)D.P + 1);
    HIPCHK(copy_sync(off.data(), D.ivf_off, off.size() * 8, hipMemcpyDeviceToHost, util_stream(ix->device)));
    if (ivf_lengths)
      for (int64_t i = 0; i < D.P; ++i) ivf_lengths[i] = (int32_t)(off[i + 1] - off[i]);
    if (ivf) {
      const int64_t tot = off[D.P];
      tmp.resize((size_t)std::min<int64_t>(chunk, std::max<int64_t>(tot, 1)));
      for (int64_t s = 0; s < tot; s += chunk) {
        const int64_t m = std::min(chunk, tot - s);
        HIPCHK(copy_sync(tmp.data(), D.ivf_pids + s, (size_t)m * 4, hipMemcpyDeviceToHost, util_stream(ix->device)));
        for (int64_t i = 0; i < m; ++i) ivf[s + i] = tmp[i];
      }
    }
  }
  return FP_OK;
}
