/*
 * fastplaid.h -- C ABI of the MI355X-native PLAID search engine (libfastplaid_hip.so).
 *
 * This is the drop-in boundary for the fast-plaid *search* hot path.  Every entry point
 * names the reference interface it replaces (paths relative to the lightonai/fast-plaid
 * tree, v1.4.6).  Plain pointers and sizes only: no torch / tch / PyO3 types.  A Rust
 * `extern "C"` block, a ctypes stub or a cgo import can bind this header unchanged
 * (INTEGRATION.md shows the Rust and ctypes bindings).
 *
 * Conventions
 *   - All functions return 0 on success, a negative FP_E* code on failure; the message
 *     is available from fp_last_error() (thread-local), mirroring anyhow_to_pyerr
 *     (rust/utils/errors.rs:5-7).
 *   - "host" pointers are ordinary CPU memory owned by the caller; the library copies
 *     what it needs.  "dev" pointers are HIP device memory on the index's device.
 *   - fp16 values travel as uint16_t bit patterns (IEEE binary16).
 *   - fp_search* are re-entrant on a shared fp_index (rust/search/load.rs:58-59
 *     `unsafe impl Send/Sync for LoadedIndex`): the index is immutable after creation,
 *     per-call scratch comes from an internal pool.
 */
#ifndef FASTPLAID_H
#define FASTPLAID_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FP_OK 0
#define FP_EINVAL (-1)        /* bad argument (shape, dtype contract, device string)   */
#define FP_ECOMPRESS_ONLY (-2) /* index has no IVF: search.rs:227-232                   */
#define FP_EHIP (-3)          /* HIP runtime failure                                   */
#define FP_EUNSUPPORTED (-4)  /* dim / nbits / size outside what the kernels are built for */
#define FP_ENOMEM (-5)

typedef struct fp_index fp_index;       /* replaces PyLoadedIndex / LoadedIndex, load.rs:46-69 */
typedef struct fp_shard_ctx fp_shard_ctx; /* one in-flight sharded search (multi-GPU path)   */

/* Argument set of `construct_index` (rust/search/load.rs:124-138), as host arrays.
 * Trailing padding rows on doc_codes/doc_residuals (python load.py:298-320) are ignored:
 * only the first sum(doc_lengths) rows are read.
 * Validation (fp_index_create returns FP_EINVAL): every doc_codes entry must be a centroid id in [0, n_centroids), every ivf
 * entry a document id in [0, n_docs) -- the reference fails on such arrays inside index_select at search time; here they would be
 * out-of-bounds device reads.  IVF lists may come in any order and repeat ids (the reference sorts and de-duplicates the gathered
 * ids per query, search.rs:538-541): lists that are not strictly ascending are sorted and de-duplicated once, at creation. */
typedef struct fp_index_desc {
  int32_t nbits;                  /* 1, 2, 4 or 8 (8 % nbits == 0)                        */
  int32_t dim;                    /* embedding dimension: any multiple of 8 up to 1024    */
  int64_t n_centroids;            /* rows of `centroids`                                  */
  const uint16_t* centroids;      /* [n_centroids, dim] f16                               */
  const uint16_t* avg_residual;   /* [dim] f16 -- accepted, unused by search (may be NULL) */
  const uint16_t* bucket_cutoffs; /* [2^nbits-1] f16 -- accepted, unused (may be NULL)    */
  const uint16_t* bucket_weights; /* [2^nbits] f16                                        */
  const int64_t* ivf;             /* concatenated IVF lists (doc ids); NULL = compress_only */
  const int32_t* ivf_lengths;     /* [n_ivf_lists]; NULL = compress_only                  */
  int64_t n_ivf_lists;            /* may exceed n_centroids (surplus lists are empty)     */
  const int64_t* doc_codes;       /* [>= n_tokens] centroid id per token                  */
  const uint8_t* doc_residuals;   /* [>= n_tokens, dim*nbits/8] packed residuals          */
  const int64_t* doc_lengths;     /* [n_docs] tokens per document                         */
  int64_t n_docs;
  int64_t pid_offset;             /* added to every returned doc id (0 for a whole index; */
                                  /* first global id of this shard for a document shard)  */
} fp_index_desc;

/* SearchParameters pyclass (rust/search/search.rs:171-200). batch_size is accepted for
 * API compatibility; it only chunks work in the reference and has no effect here. */
typedef struct fp_search_params {
  int64_t batch_size;
  int64_t n_full_scores;
  int64_t top_k;
  int64_t n_ivf_probe;
} fp_search_params;

/* ---- library ------------------------------------------------------------------------ */
const char* fp_last_error(void);        /* errors.rs:5-7 */
const char* fp_version(void);
int fp_device_count(void);
/* free / total HBM of a device in bytes (hipMemGetInfo); negative on error */
int64_t fp_device_free_bytes(int device_id);
int64_t fp_device_total_bytes(int device_id);

/* ---- index lifetime ----------------------------------------------------------------- */
/* construct_index (load.rs:124-186): uploads and re-lays-out the arrays on device
 * `device_id`.  The caller keeps ownership of the host buffers. */
int fp_index_create(const fp_index_desc* desc, int device_id, fp_index** out);
/* Drop of PyLoadedIndex (load.rs:61-69) / FastPlaid.close() (fast_plaid.py:387-396). */
void fp_index_destroy(fp_index* index);
/* Introspection (sizes the caller needs to allocate outputs). */
int64_t fp_index_num_docs(const fp_index* index);
int64_t fp_index_num_tokens(const fp_index* index);
int64_t fp_index_num_centroids(const fp_index* index);
int32_t fp_index_dim(const fp_index* index);
int32_t fp_index_nbits(const fp_index* index);
int64_t fp_index_device_bytes(const fp_index* index);
/* Entries of the per-document unique-code lists (the approximate stage walks these instead of the per-token codes). */
int64_t fp_index_num_unique_codes(const fp_index* index);
/* 128-byte lines of the packed per-document unique codes that S4's level-0 stage streams (0 when the index has none) */
int64_t fp_index_num_code_lines(const fp_index* index);
/* tokens whose normalisation has no one-multiply reciprocal (the MaxSim kernel takes the compensated path for their steps) */
int64_t fp_index_num_hard_tokens(const fp_index* index);
/* 1 when the fence-free "last workgroup finishes the job" launches passed their self-test on the index's device at creation (they
 * then replace the count -> scan -> offsets launch chains), 0 when the plain chains are used -- results are the same. */
int32_t fp_index_tickets_ok(const fp_index* index);

/* ---- search: `pysearch` (rust/lib.rs:195-223) -> search_many (search.rs:219-288) ----- */
/* queries: host [n_queries, q_len, dim] f16.  subset_offsets NULL = no subset; otherwise
 * [n_queries+1] offsets into subset_ids (per-query doc-id lists, lib.rs:202).
 * Outputs (host, caller-allocated): out_pids/out_scores [n_queries, top_k] row-major,
 * out_counts [n_queries] = results per query = min(top_k, max(n_full/4,1), candidates)
 * (search.rs:614, :666); rows are sorted by descending score; the slots [count, top_k) of
 * a row hold id -1 and score 0.  A query whose per-query search fails in the reference
 * (search.rs:268 `.unwrap_or_default()`) gets count 0 and a row of -1 / 0 like any other unused slot -- also when the whole
 * call returns before anything is enqueued (an empty index, n_ivf_probe > n_centroids without a subset): every row the caller
 * passed is written by every successful call, so the buffers need no pre-fill.
 *
 * What is exact, and on what evidence.  The returned ids are the reference's, in the reference's order (documents whose
 * reference scores are EXACTLY equal may come in another order: the reference's own tie order is implementation-defined; ours
 * is id ascending).  Probed cells, candidate sets and the rerank set are the reference's; fp_search_trace additionally
 * reproduces the centroid scores and every candidate's approximate score bit for bit (fp_search itself stores upper
 * candidates of the centroid scores and settles exactly the entries its decisions depend on: see fp_last_s1_counts).  The
 * returned MaxSim SCORES are within 1e-3 of the reference's: the MFMA pass can round one query column of a document one fp16
 * step away (<= 4.9e-4); only the documents whose rank could depend on that -- the near-tied ones -- are re-evaluated in the
 * reference's summation order (their scores are then the reference's bit for bit), the others keep the MFMA score.
 * FP_MAXSIM_REPAIR=2 in the environment re-evaluates every flagged document: every returned score is then the reference's.
 *
 * This rests on two CERTIFICATION WINDOWS that are empirical margins, not worst-case bounds.  An fp32 MFMA result is taken as
 * "the reference's ascending fp32 chain rounds to the same fp16 value" when it lies further from every fp16 rounding boundary
 * than w0 |q| c_max + kappa |x| (centroid scores: w0 = 2^-21.5, kappa = 2^-20) resp. 2^-19 |q_col| (MaxSim column maxima); the
 * worst-case reordering error of a D-term fp32 sum, D 2^-24 sum |a_k b_k|, is ~2^-17 at D = 128 and would flag far more.
 * Measured: 0 unflagged differences in 12.9 G centroid scores and 12.8 M column maxima, the first unflagged difference appears
 * at a QUARTER of the shipped centroid window (profiles/r04_s1_window_sweep.txt, r04_cert_stats.jsonl).  So "identical ids"
 * holds with overwhelming probability on data like the tested corpora, not by construction; an adversarially built input is
 * not covered.  FP_TEST=s1_w0_log2=..,s1_kappa_log2=.. widen the centroid window (cost: more entries re-evaluated).
 */
int fp_search(const fp_index* index, const uint16_t* queries, int32_t n_queries, int32_t q_len, int32_t dim,
              const fp_search_params* params, const int64_t* subset_ids, const int64_t* subset_offsets,
              int64_t* out_pids, float* out_scores, int32_t* out_counts);

/* fp_search with ONE subset for every query: `subset: list[int]` of FastPlaid.search (fast_plaid.py:784-793), which the
 * reference replicates per query before the native call (lib.rs:202 takes Vec<Vec<i64>>).  Passing the list once saves the
 * n_queries - 1 copies, uploads and bitmap builds; results are those of fp_search with the list repeated for every query. */
int fp_search_shared_subset(const fp_index* index, const uint16_t* queries, int32_t n_queries, int32_t q_len, int32_t dim,
                            const fp_search_params* params, const int64_t* subset_ids, int64_t n_subset,
                            int64_t* out_pids, float* out_scores, int32_t* out_counts);

/* Plain device-memory helpers for callers without a HIP binding of their own (bench.py, tests): allocate / free / copy on
 * the given GPU through the same HIP runtime this library uses. */
int fp_dev_alloc(int device_id, size_t bytes, void** out_dev_ptr);
int fp_dev_free(int device_id, void* dev_ptr);
int fp_dev_upload(int device_id, void* dev_dst, const void* host_src, size_t bytes);
int fp_dev_download(int device_id, void* host_dst, const void* dev_src, size_t bytes);

/* fp_search with the queries already in HBM and the results left in HBM: every pointer except `params` is a DEVICE
 * pointer on the index's GPU (no subsets).  Results are complete when the call returns.  bench.py times fp_search (the host
 * boundary) for `value` and this entry point for `value_device_io`; the difference is the PCIe copy of queries and results. */
int fp_search_device(const fp_index* index, const uint16_t* dev_queries, int32_t n_queries, int32_t q_len, int32_t dim,
                     const fp_search_params* params, int64_t* dev_out_pids, float* dev_out_scores, int32_t* dev_out_counts);

/* Same search with every stage output copied back (parity tests / profiling only).
 * One query.  Any output pointer may be NULL.  Capacities: S [n_centroids*q_len] f16 as
 * [c][q]; cells [q_len*n_ivf_probe]; cand/approx [n_docs]; rerank/exact [max(n_full/4,1)].
 * counts[3] = {n_cells, n_cand, n_rerank}.  rerank is in ascending doc-id order. */
int fp_search_trace(const fp_index* index, const uint16_t* query, int32_t q_len, int32_t dim,
                    const fp_search_params* params, const int64_t* subset_ids, int64_t n_subset, int32_t has_subset,
                    int64_t* out_pids, float* out_scores, int32_t* out_count, uint16_t* S, int64_t* cells,
                    int64_t* cand, float* approx, int64_t* rerank, float* exact, int64_t* counts);

/* Per-stage device time of the most recent fp_search on this thread, in milliseconds
 * (HIP events on the search stream).  names[i] are static strings.  Returns the number
 * of stages written (<= cap). */
int fp_last_search_timings(const char** names, float* ms, int cap);

/* fp_search replays one captured HIP graph per batch from a shape's third batch on (query upload from a pinned staging buffer,
 * every launch and fill, result download: one hipGraphLaunch instead of ~55 enqueues).  A graph launch records no per-stage
 * events, so fp_last_search_timings reports zeros for such a call; a profiler or benchmark that wants stage times switches the
 * replay off for the calls it measures.  enabled: 0 / 1; returns the previous setting.  Process-wide; default 1 (FP_GRAPH=0 in
 * the environment starts with 0).
 * Threads: the capture runs on the call's own non-blocking stream in relaxed mode; every entry point of this library puts its
 * thread into the relaxed capture-interaction mode and none uses the legacy stream (one exception: a 64-byte hipMemset that ends an index construction when no capture of the library is open), so concurrent calls into the library do not
 * disturb a capture.  A synchronous legacy-stream call (hipMemcpy, hipMemset) made by ANOTHER thread of the application while a
 * capture is open (~0.1 ms, once per shape) is refused by the runtime and invalidates the capture: fp_search then runs that batch
 * on the plain path and stops capturing on that scratch -- its result is unaffected.  Applications with such threads may prefer
 * to switch the replay off ("Graph capture and the application's other threads"). */
int fp_set_graph_replay(int enabled);
/* fp_search calls served by a graph replay since the process started.  The learnt candidate capacity and the captured graph are
 * kept per shape {n_queries, q_len, n_ivf_probe, n_full_scores, top_k} for the eight most recently used shapes of a scratch, so a
 * caller that alternates between a few shapes keeps replaying each of them. */
uint64_t fp_graph_replay_count(void);

/* Work counters of the most recent fp_search on this thread: out[0] = candidate documents
 * summed over queries (S3 output), out[1] = candidates that reached the exact approximate-score
 * kernel (== out[0] unless a bound stage pruned), out[2] = documents whose near-tied final score was re-evaluated in the
 * reference's summation order (exact-order repair), out[3] = sub-batches, out[4] = the form of S4 the last sub-batch ran
 * (0 every candidate scored exactly, 1 8-bit bound stage, 2 level 0, 3 level 0 over the hot codes; -1 when the call was a graph
 * replay), out[5] = lazy-S1 list overflows the scratch that served the call remembers (>= 2: S1's lazy form is off for it, to
 * be tried again after 256 batches).  Returns entries written (cap >= 6 for all). */
int fp_last_search_counts(int64_t* out, int cap);

/* Centroid scores (S1, search.rs:491).  Every fp32 MFMA result x is certified against the fp16 rounding boundaries with the
 * window u = w0 |q| c_max + kappa |x| (see fp_search).  FP_S1_EXACT in the environment picks what happens to the entries inside
 * the window: 3 (default) = LAZY form wherever the threshold probe and the general selection serve the shape (no subset,
 * n_ivf_probe <= 32, q_len <= 128, n_full_scores / 4 <= 8192, fewer than 2^24 centroids): S holds h(x + u) for EVERY entry -- the
 * reference's value or slightly above it, with a lower bound computable from the stored value alone -- and the consumers whose
 * decisions depend on exact values (the probe's ranking, the rerank selection) re-evaluate with the reference's ascending chain
 * exactly the entries that can change them; probed cells, candidates, the selected set and the results are the reference's.
 * Elsewhere, and always with 1 = EAGER form (fp_search_trace, subsets, the sharded search's staged protocol): the flagged
 * entries are re-evaluated inside the kernel and S itself equals the reference's bit for bit.  2 = every entry re-evaluated
 * (tests), 0 = no certification (S differs from the reference's by one fp16 ulp in ~0.05 % of its entries).
 * With FP_S1_STATS set (diagnostics: it also switches the graph replay off) the counters of the most recent fp_search on this
 * thread are kept.  Eager batch: out[0] = flagged entries, out[1] = re-evaluated entries whose value is not the upper candidate
 * that was staged, out[2] = entries that overflowed a wave's list (slow path), out[3] = FP_S1_EXACT=2 only: entries the window
 * did NOT flag whose chain value differs from the MFMA's -- must be 0.  Lazy batch: out[0] = documents the selection took as
 * certain, out[1] = "maybes" whose approximate score was recomputed exactly, out[2] = the most maybes of one query, out[3] =
 * (code, column) pairs re-evaluated.  out[4] (kept without FP_S1_STATS too) = 1 the last batch ran the lazy form, 0 the eager
 * one, -1 it was a replayed graph.  Returns entries written (cap >= 5 for all). */
int fp_last_s1_counts(uint64_t* out, int cap);

/* Diagnostic / test entry point: the MFMA pass of the exact stage (S6+S7) on the given documents for ONE query, before the
 * exact-order repair.  scores [n]; col_max [n, q_len] f16 bits = per query column the maximum over the document's tokens as the
 * kernel rounds it; unc [n] = uncertainty budget (sum of the fp16 ulps of the flagged columns); flags [n, ceil(q_len/32)] = bit
 * per flagged column (a column is flagged when its fp32 maximum lies so close to an fp16 rounding boundary that the
 * reference's summation order could round it the other way).  Shapes without an MFMA kernel: col_max / flags come back zero
 * (that path is exact: nothing is ever flagged).  Any output but scores may be NULL. */
int fp_maxsim_columns(const fp_index* index, const uint16_t* query_f16, int32_t q_len, int32_t dim, const int64_t* pids, int64_t n,
                      float* scores, uint16_t* col_max, float* unc, uint32_t* flags);

/* ---- `pysearch_with_token_scores` (rust/lib.rs:243-290 -> search_many_with_token_scores,
 *      rust/search/search.rs:294-363; matrix extraction :668-686) ------------------------------
 * Token-level similarity matrices of search hits: for hit h = (query b, rank i) the matrix
 * [q_len, doc_len(pid)] of fp16 values h(sum_fp32 e_hat[t] . q[j]) in the document's ORIGINAL
 * token order, row-major, at out + out_offsets[h] (hits ordered query-major, then rank).
 * Call fp_search first and pass its pids / counts.  With out == NULL only out_offsets
 * [n_hits + 1] (element offsets) is filled, so the caller can size `out`; out_capacity is in
 * elements.  Ids outside the index give FP_EINVAL. */
int fp_token_scores(const fp_index* index, const uint16_t* queries_f16, int32_t n_queries, int32_t q_len, int32_t dim,
                    const int64_t* pids /*[n_queries, stride]*/, const int32_t* counts /*[n_queries]*/, int64_t stride,
                    int64_t* out_offsets /*[sum(counts) + 1]*/, uint16_t* out /*f16, nullable*/, int64_t out_capacity);

/* ---- index creation, device part -----------------------------------
 * rust/index/create.rs:148-170 compress_into_codes, :404-428 process_batch, :176-184 packbits:
 *   codes[t]  = argmax_c h(sum_fp32 emb[t] . cent[c])          (first maximal index, ascending-k fp32 sum)
 *   r[t][d]   = h(emb[t][d] - cent[codes[t]][d])
 *   bucket    = #{ i : cutoffs[i] < r }                          (torch.bucketize, right=False)
 *   bytes     = each bucket as nbits bits LSB first, the bit stream packed 8 per byte big-endian.
 * Stand-alone (no index handle): host buffers in and out, processed in device-sized chunks.
 * k-means, codec training (quantiles) and file writing stay on the host (fast-plaid_amd/create.py). */
int fp_compress(int device_id, const uint16_t* centroids_f16 /*[n_centroids, dim]*/, int64_t n_centroids, int32_t dim, int32_t nbits,
                const uint16_t* bucket_cutoffs_f16 /*[2^nbits - 1]*/, const uint16_t* embeddings_f16 /*[n_tokens, dim]*/,
                int64_t n_tokens, int64_t* out_codes /*[n_tokens]*/, uint8_t* out_residuals /*[n_tokens, dim*nbits/8]*/);

/* K-means assignment step for the Python k-means driver (python/fast_plaid/search/fast_plaid.py:71-185 calls the
 * external fastkmeans package; its Lloyd iteration assigns every point to the centroid of least squared L2 distance):
 * out_labels[t] = argmax_c ( emb[t] . cent[c] - half_sqnorm[c] ), fp32, ties -> lowest index. */
int fp_assign_l2(int device_id, const uint16_t* centroids_f16 /*[n_centroids, dim]*/, const float* half_sqnorm /*[n_centroids]*/,
                 int64_t n_centroids, int32_t dim, const uint16_t* embeddings_f16 /*[n_tokens, dim]*/, int64_t n_tokens,
                 int64_t* out_labels /*[n_tokens]*/);

/* ---- `reconstruct_embeddings` (rust/utils/embeddings.rs:12-69) ----------------------- */
/* Decompresses whole documents to fp32 rows. out: host [sum(len(doc)) , dim] f32 in the
 * order of doc_ids (ids may repeat); out_lengths [n]. `out_capacity_rows` guards the buffer: when it is too small the call fails
 * with FP_EINVAL AFTER filling out_lengths for every requested document, so sum(out_lengths) is the capacity to come back with. */
int fp_reconstruct_embeddings(const fp_index* index, const int64_t* doc_ids, int64_t n, float* out,
                              int64_t out_capacity_rows, int64_t* out_lengths);

/* ---- document-sharded search (one process per GPU; collectives stay with the caller) -
 * New relative to the reference, which only runs full replicas per device
 * (fast_plaid.py:893-928).  Every rank holds the full centroid table and a disjoint
 * document shard (fp_index_desc.pid_offset = first global id).  The four stages are
 * separated exactly where data must cross ranks; between them the caller all-gathers ONE
 * fixed-size device buffer of records per rank (RCCL: torch.distributed "nccl" on ROCm, or
 * ncclAllGather on bytes):
 *
 *   stage1: S1-S4 on the local shard + local top-R candidates by approximate score
 *           -> dev rec1 [B,R]  (fp_shard_rec1; padding: pid -1, approx -inf)
 *   (all-gather to [G,B,R])
 *   stage2: global top-R cut (reproduces search.rs:605-619 on the union), exact MaxSim of
 *           the survivors that live on this rank
 *           -> dev rec2 [B,R]  (fp_shard_rec2: the MFMA score, its uncertainty budget `unc` and the part of it
 *              by which the reference's score may be lower, `unc_down`)
 *   (all-gather to [G,B,R])
 *   stage3: union in ascending id order, the unsharded search's near-tie marking on it
 *           (identical on every rank); this rank repairs the marked documents it holds
 *           -> dev x [B,R] f32 (by union position; only this rank's marked entries are meaningful)
 *   (all-gather to [G,B,R])
 *   stage4: marked documents take the repaired score of their rank, global sort + top_k
 *           -> host outputs as fp_search.
 * R = max(n_full_scores/4, 1).  The result is identical to fp_search on the whole corpus,
 * bit for bit, for any G. */
typedef struct fp_shard_rec1 { int64_t pid; float approx; int32_t pad; } fp_shard_rec1;                        /* 16 bytes */
typedef struct fp_shard_rec2 { int64_t pid; float score; float unc_down; float unc; int32_t pad; } fp_shard_rec2; /* 24 bytes */
int fp_shard_begin(const fp_index* index, const uint16_t* queries, int32_t n_queries, int32_t q_len, int32_t dim,
                   const fp_search_params* params, fp_shard_ctx** out);
int64_t fp_shard_R(const fp_shard_ctx* ctx);
int fp_shard_stage1(fp_shard_ctx* ctx, void* dev_rec1 /*[B,R] fp_shard_rec1*/);
int fp_shard_stage2(fp_shard_ctx* ctx, const void* dev_all_rec1 /*[G,B,R]*/, int32_t n_ranks, void* dev_rec2 /*[B,R] fp_shard_rec2*/);
int fp_shard_stage3(fp_shard_ctx* ctx, const void* dev_all_rec2 /*[G,B,R]*/, int32_t n_ranks, int32_t rank, void* dev_x /*[B,R] f32*/);
int fp_shard_stage4(fp_shard_ctx* ctx, const void* dev_all_x /*[G,B,R] f32*/, int32_t n_ranks, int64_t* out_pids, float* out_scores,
                    int32_t* out_counts);
void fp_shard_end(fp_shard_ctx* ctx);

/* The same sharded search with the collectives issued by the library (RCCL over xGMI, bound with dlopen at first use): the
 * all-gathers are enqueued on the search stream behind the kernels that fill their send buffers -- no host synchronisation
 * between the stages, no framework in the data path (three all-gathers: 16 + 24 + 4 bytes per rerank slot).  One communicator per process / GPU:
 *   rank 0: fp_comm_unique_id(id) -> the caller ships the 128 bytes to every rank (any out-of-band channel)
 *   all   : fp_comm_create(device, n_ranks, rank, id, &comm)      (collective: ncclCommInitRank)
 *   all   : fp_shard_search(shard_index, comm, ...)                (collective; identical results on every rank, == fp_search
 *                                                                   on the whole corpus)
 * Every rank passes the same queries / parameters.  Not re-entrant on one communicator.
 * From a shape's second batch on, a rank sizes its candidate buffers from earlier batches instead of waiting for the current
 * total (as fp_search does).  A rank whose batch outgrows that capacity marks record 0 of its first-exchange block
 * (fp_shard_rec1.pad = 1); every rank sees the mark and the whole call is run once more inside the library -- callers of
 * fp_shard_search notice nothing, callers of the staged entry points (which always wait) always see pad = 0. */
typedef struct fp_comm fp_comm;
int fp_comm_unique_id(void* out_id_128_bytes);
int fp_comm_create(int device_id, int n_ranks, int rank, const void* unique_id_128_bytes, fp_comm** out);
void fp_comm_destroy(fp_comm* comm);
int fp_comm_n_ranks(const fp_comm* comm);
int fp_comm_rank(const fp_comm* comm);
int fp_shard_search(const fp_index* index, fp_comm* comm, const uint16_t* queries, int32_t n_queries, int32_t q_len, int32_t dim,
                    const fp_search_params* params, int64_t* out_pids, float* out_scores, int32_t* out_counts);

/* ---- synthetic corpora generated in HBM (benchmark + full-size property tests) ------- */
/* Builds an index whose codes / residual bytes / lengths / IVF are generated on the
 * device from the counter hash documented in fast-plaid_amd/synth.py (bit-identical to
 * that numpy twin).  Centroids and bucket weights come from the host.  Documents
 * [doc_begin, doc_end) of the virtual corpus are materialised (a shard); pid_offset is
 * doc_begin. */
typedef struct fp_synth_desc {
  int32_t nbits, dim;
  int64_t n_centroids;          /* power of two */
  const uint16_t* centroids;    /* host [n_centroids, dim] f16 */
  const uint16_t* bucket_weights; /* host [2^nbits] f16 */
  int64_t n_docs_total;         /* size of the virtual corpus */
  int64_t doc_begin, doc_end;   /* shard of it held by this index */
  int32_t doc_len;              /* tokens per doc (max when variable_len) */
  int32_t variable_len;         /* lengths ~ U[doc_len/4, doc_len] */
  uint64_t seed;
} fp_synth_desc;
int fp_index_create_synthetic(const fp_synth_desc* desc, int device_id, fp_index** out);
/* Copies a document's compressed arrays back to the host (tests): codes [len] i64,
 * residuals [len, dim*nbits/8] u8. Returns the length, or a negative error. */
int64_t fp_index_read_doc(const fp_index* index, int64_t local_doc, int64_t* codes, uint8_t* residuals,
                          int64_t capacity_tokens);
/* Copies the IVF list of one cell back (tests). Returns its length. */
int64_t fp_index_read_ivf(const fp_index* index, int64_t cell, int64_t* pids, int64_t capacity);

/* Copies the index arrays back to the host in the construct_index layout (any pointer may
 * be NULL): doc_codes [n_tokens] i64, doc_residuals [n_tokens, dim*nbits/8], doc_lengths
 * [n_docs], ivf [fp_index_ivf_total] i64 (local doc ids), ivf_lengths [n_centroids] i32.
 * Lets a device-generated or device-built index be handed to the reference
 * (search/index_io.py writes the reference's directory format) or to the CPU oracle. */
int64_t fp_index_ivf_total(const fp_index* index);
int fp_index_export(const fp_index* index, int64_t* doc_codes, uint8_t* doc_residuals, int64_t* doc_lengths, int64_t* ivf,
                    int32_t* ivf_lengths);

/* ---- self-test --------------------------------------------------------------------------- */
/* Exhaustive (all 2^32 fp16 pairs) device check that the two arithmetic shortcuts of the
 * MaxSim kernel equal the reference formulation "fp32 op + one rounding to fp16":
 * out[0] = mismatches of the compensated reciprocal product h(fma(e, r_hi, e*r_lo)) vs h(e / n)
 *          over the reachable domain (n >= 0 or NaN, |e| <= n(1+2^-9)),
 * out[1] = mismatches of the packed fp16 add vs h(fp32 add) over all pairs: both must be 0.
 * Informational: out[2] = plain single product h(e * fl32(1/n)) on the reachable domain,
 * out[3] = compensated product over ALL pairs, out[4] = sample count, out[5..15] = samples
 * packed as e | n<<16 | fast<<32 | reference<<48.  out_mismatches has room for 16 values. */
int fp_selftest_arith(int device_id, uint64_t* out_mismatches);

#ifdef __cplusplus
}
#endif
#endif /* FASTPLAID_H */
