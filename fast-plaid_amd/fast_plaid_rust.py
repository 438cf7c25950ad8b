"""Mirror of the reference's native module ``fast_plaid.fast_plaid_rust`` (rust/lib.rs:366-383)
for the search path, on top of the C ABI in include/fastplaid.h.

Same names, argument meaning and error behaviour as the PyO3 functions:
  SearchParameters, QueryResult          rust/search/search.rs:114-200
  construct_index(...) -> PyLoadedIndex  rust/search/load.rs:124-186
  pysearch(...) -> list[QueryResult]     rust/lib.rs:195-223
  reconstruct_embeddings(...)            rust/utils/embeddings.rs:12-69
  initialize_torch(...)                  rust/lib.rs:100-104 (no libtorch here: a no-op)
Index maintenance entry points (create / update / delete, lib.rs:132-165, :302-364) are outside
the search hot path; create / update / delete delegate to create.py / maintain.py (their native part).

Tensors may be torch CPU tensors or numpy arrays; nothing here needs torch.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N


def _np(x, dtype):
    if x is None:
        return None
    if hasattr(x, "detach") and hasattr(x, "cpu"):  # torch tensor
        x = x.detach().cpu()
        if str(x.dtype) == "torch.bfloat16":
            x = x.float()
        x = x.numpy()
    return np.ascontiguousarray(np.asarray(x), dtype=dtype)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _device_id(device: str) -> int:
    """rust/search/load.rs:16-37 get_device ("cuda" is how ROCm devices are named in torch)."""
    d = str(device).lower()
    if d in ("cuda", "hip"):
        return 0
    if d.startswith("cuda:") or d.startswith("hip:"):
        parts = d.split(":")
        if len(parts) == 2:
            try:
                return int(parts[1])
            except ValueError:
                raise ValueError(f"Invalid CUDA device index: '{parts[1]}'")
        raise ValueError("Invalid CUDA device format. Expected 'cuda:N'.")
    if d == "cpu":
        raise ValueError("This build is MI355X-only: device 'cpu' is not available (no CPU fallback).")
    raise ValueError(f"Unsupported device string: '{device}'")


class SearchParameters:
    """rust/search/search.rs:171-200."""

    def __init__(self, batch_size: int, n_full_scores: int, top_k: int, n_ivf_probe: int):
        self.batch_size = int(batch_size)
        self.n_full_scores = int(n_full_scores)
        self.top_k = int(top_k)
        self.n_ivf_probe = int(n_ivf_probe)

    def _c(self) -> N.FpSearchParams:
        return N.FpSearchParams(self.batch_size, self.n_full_scores, self.top_k, self.n_ivf_probe)


class QueryResult:
    """rust/search/search.rs:114-126."""

    __slots__ = ("query_id", "passage_ids", "scores")

    def __init__(self, query_id, passage_ids, scores):
        self.query_id = query_id
        self.passage_ids = passage_ids
        self.scores = scores


class PyLoadedIndex:
    """Opaque handle (rust/search/load.rs:61-69); frees device memory when dropped."""

    def __init__(self, handle, device_id: int, keepalive=()):
        self._h = handle
        self.device_id = device_id
        self._keep = keepalive

    @property
    def n_docs(self):
        return int(N.lib().fp_index_num_docs(self._h))

    @property
    def n_tokens(self):
        return int(N.lib().fp_index_num_tokens(self._h))

    @property
    def n_centroids(self):
        return int(N.lib().fp_index_num_centroids(self._h))

    @property
    def dim(self):
        return int(N.lib().fp_index_dim(self._h))

    @property
    def nbits(self):
        return int(N.lib().fp_index_nbits(self._h))

    @property
    def device_bytes(self):
        return int(N.lib().fp_index_device_bytes(self._h))

    @property
    def n_unique_codes(self):
        return int(N.lib().fp_index_num_unique_codes(self._h))

    @property
    def n_code_lines(self):
        """128-byte lines of packed unique codes that S4's level-0 stage streams (all centroid ranges)."""
        return int(N.lib().fp_index_num_code_lines(self._h))

    @property
    def n_hard_tokens(self):
        """tokens whose normalisation has no one-multiply reciprocal (k_token_rinv): their steps take the compensated path"""
        return int(N.lib().fp_index_num_hard_tokens(self._h))

    @property
    def tickets_ok(self):
        """the fence-free "last workgroup finishes" launches passed their self-test on this index's device (else: plain launch chains)"""
        return bool(N.lib().fp_index_tickets_ok(self._h))

    def close(self):
        if getattr(self, "_h", None):
            N.lib().fp_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def initialize_torch(torch_path: str | None = None) -> None:  # lib.rs:100-104
    return None


def construct_index(nbits, centroids, avg_residual, bucket_cutoffs, bucket_weights, ivf, ivf_lengths, doc_codes,
                    doc_residuals, doc_lengths, device: str, low_memory: bool = False, pid_offset: int = 0) -> PyLoadedIndex:
    """rust/search/load.rs:124-186.  `low_memory` (keep document tensors on the host) has no
    MI355X counterpart -- 288 GB of HBM holds the whole index -- and is accepted and ignored."""
    del low_memory
    dev = _device_id(device)
    cent = _np(centroids, np.float16)
    if cent.ndim != 2:
        raise ValueError("centroids must be [n_centroids, dim]")
    bw = _np(bucket_weights, np.float16)
    if bw.shape[0] != (1 << int(nbits)):
        raise ValueError("bucket_weights must have 2^nbits entries")
    ivf_a = _np(ivf, np.int64)
    ivfl = _np(ivf_lengths, np.int32)
    codes = _np(doc_codes, np.int64)
    res = _np(doc_residuals, np.uint8)
    lens = _np(doc_lengths, np.int64)
    dim = int(cent.shape[1])
    T = int(lens.sum()) if lens.size else 0
    if codes.shape[0] < T or res.shape[0] < T:
        raise ValueError("doc_codes / doc_residuals hold fewer rows than sum(doc_lengths)")
    if res.ndim != 2 or res.shape[1] != dim * int(nbits) // 8:
        raise ValueError("doc_residuals must be [n_tokens, dim*nbits/8]")
    if ivf_a is not None and ivfl is not None and int(ivfl.sum()) > ivf_a.shape[0]:
        raise ValueError("ivf shorter than sum(ivf_lengths)")
    if T and (codes[:T].min() < 0 or codes[:T].max() >= cent.shape[0]):
        raise ValueError("doc_codes out of range of the centroid table")
    if ivf_a is not None and ivf_a.size and (ivf_a.min() < 0 or ivf_a.max() >= lens.shape[0]):
        raise ValueError("ivf holds document ids out of range")
    d = N.FpIndexDesc(
        int(nbits), dim, int(cent.shape[0]), _ptr(cent), None, None, _ptr(bw), _ptr(ivf_a), _ptr(ivfl),
        0 if ivfl is None else int(ivfl.shape[0]), _ptr(codes), _ptr(res), _ptr(lens), int(lens.shape[0]), int(pid_offset))
    h = C.c_void_p()
    N.check(N.lib().fp_index_create(C.byref(d), dev, C.byref(h)))
    return PyLoadedIndex(h, dev)


def construct_synthetic_index(spec, device: str, doc_begin: int = 0, doc_end: int | None = None,
                              centroids=None, bucket_weights=None) -> PyLoadedIndex:
    """Device-generated corpus (include/fastplaid.h fp_index_create_synthetic); `spec` is a
    synth.SynthSpec.  Not part of the reference API: benchmark / property-test plumbing."""
    from . import synth
    dev = _device_id(device)
    cent = _np(synth.centroids(spec) if centroids is None else centroids, np.float16)
    bw = _np(synth.bucket_weights(spec) if bucket_weights is None else bucket_weights, np.float16)
    end = spec.n_docs if doc_end is None else int(doc_end)
    d = N.FpSynthDesc(spec.nbits, spec.dim, spec.n_centroids, _ptr(cent), _ptr(bw), spec.n_docs, int(doc_begin), end,
                      spec.doc_len, 1 if spec.variable_len else 0, spec.seed)
    h = C.c_void_p()
    N.check(N.lib().fp_index_create_synthetic(C.byref(d), dev, C.byref(h)))
    return PyLoadedIndex(h, dev)


def _flatten_subset(subset, n_queries):
    if subset is None:
        return None, None
    if len(subset) != n_queries:
        raise ValueError("Subset length must match number of queries.")
    off = np.zeros(n_queries + 1, dtype=np.int64)
    for i, s in enumerate(subset):
        off[i + 1] = off[i] + len(s)
    ids = np.zeros(max(int(off[-1]), 1), dtype=np.int64)
    for i, s in enumerate(subset):
        if len(s):
            ids[off[i]: off[i + 1]] = np.asarray(s, dtype=np.int64)
    return ids, off


def search_arrays(index: PyLoadedIndex, queries_f16: np.ndarray, params: SearchParameters, subset=None):
    """Array-level pysearch: returns (pids [B,top_k] i64, scores [B,top_k] f32, counts [B] i32)."""
    q = _np(queries_f16, np.float16)
    if q.ndim != 3:
        raise ValueError(f"Expected a 3D tensor for queries, but got shape {list(q.shape)}")
    B, Q, D = q.shape
    k = max(params.top_k, 0)
    # (every successful fp_search writes every row it was passed -- unused slots -1 / 0 -- so the result buffers are not
    # pre-filled: np.full over 512 KB was ~1 % of a cfg2 call)
    pids = np.empty((B, max(k, 1)), dtype=np.int64)
    scores = np.empty((B, max(k, 1)), dtype=np.float32)
    counts = np.zeros(max(B, 1), dtype=np.int32)
    if k == 0 or B == 0:
        pids.fill(-1)
        scores.fill(0)
    p = params._c()
    if subset is not None and len(subset) == B and B > 1 and all(s is subset[0] for s in subset):
        # one list object repeated for every query (what FastPlaid.search makes of `subset: list[int]`): converted and passed once
        ids = np.ascontiguousarray(np.asarray(subset[0], dtype=np.int64)) if len(subset[0]) else np.zeros(1, np.int64)
        N.check(N.lib().fp_search_shared_subset(index._h, _ptr(q), B, Q, D, C.byref(p), _ptr(ids), len(subset[0]), _ptr(pids), _ptr(scores),
                                                _ptr(counts)))
        return pids[:, :k] if k else pids[:, :0], scores[:, :k] if k else scores[:, :0], counts[:B]
    ids, off = _flatten_subset(subset, B)
    N.check(N.lib().fp_search(index._h, _ptr(q), B, Q, D, C.byref(p), _ptr(ids), _ptr(off), _ptr(pids), _ptr(scores), _ptr(counts)))
    return pids[:, :k] if k else pids[:, :0], scores[:, :k] if k else scores[:, :0], counts[:B]


class DeviceBuffer:
    """a block of HBM on the index's GPU (fp_dev_alloc); freed when dropped."""

    def __init__(self, device_id: int, nbytes: int):
        self.device_id, self.nbytes = int(device_id), int(nbytes)
        p = C.c_void_p()
        N.check(N.lib().fp_dev_alloc(self.device_id, self.nbytes, C.byref(p)))
        self.ptr = p

    def upload(self, arr: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(arr)
        if a.nbytes > self.nbytes:
            raise ValueError("device buffer too small")
        N.check(N.lib().fp_dev_upload(self.device_id, self.ptr, _ptr(a), a.nbytes))
        return self

    def download(self, dtype, shape) -> np.ndarray:
        out = np.zeros(shape, dtype)
        if out.nbytes > self.nbytes:
            raise ValueError("device buffer too small")
        N.check(N.lib().fp_dev_download(self.device_id, _ptr(out), self.ptr, out.nbytes))
        return out

    def close(self):
        if getattr(self, "ptr", None) is not None and self.ptr:
            N.lib().fp_dev_free(self.device_id, self.ptr)
            self.ptr = None

    __del__ = close


def search_device(index: PyLoadedIndex, dev_queries: DeviceBuffer, n_queries: int, q_len: int, params: SearchParameters,
                  dev_pids: DeviceBuffer, dev_scores: DeviceBuffer, dev_counts: DeviceBuffer) -> None:
    """fp_search_device: queries already in HBM ([n_queries, q_len, dim] f16), results left in HBM
    ([n_queries, top_k] i64 / f32, [n_queries] i32)."""
    N.check(N.lib().fp_search_device(index._h, dev_queries.ptr, n_queries, q_len, index.dim, C.byref(params._c()), dev_pids.ptr,
                                     dev_scores.ptr, dev_counts.ptr))


def pysearch(index: PyLoadedIndex, device: str, queries_embeddings, search_parameters: SearchParameters,
             show_progress: bool = False, subset=None) -> list[QueryResult]:
    """rust/lib.rs:195-223.  `device` must name the device the index lives on."""
    del show_progress
    if _device_id(device) != index.device_id:
        raise ValueError(f"index lives on device {index.device_id}, search requested on '{device}'")
    pids, scores, counts = search_arrays(index, queries_embeddings, search_parameters, subset)
    return [QueryResult(b, pids[b, : counts[b]].tolist(), scores[b, : counts[b]].tolist()) for b in range(pids.shape[0])]


def search_trace(index: PyLoadedIndex, query_f16, params: SearchParameters, subset=None) -> dict:
    """One query with every stage output (tests / profiling); see fp_search_trace."""
    q = _np(query_f16, np.float16)
    Q, D = q.shape
    Cn, Nd = index.n_centroids, index.n_docs
    R = max(params.n_full_scores // 4, 1)
    k = max(params.top_k, 1)
    S = np.zeros((Cn, Q), np.float16)
    cells = np.zeros(max(Q * max(params.n_ivf_probe, 1), 1), np.int64)
    cand = np.zeros(max(Nd, 1), np.int64)
    approx = np.zeros(max(Nd, 1), np.float32)
    rer = np.zeros(R, np.int64)
    exact = np.zeros(R, np.float32)
    counts = np.zeros(3, np.int64)
    pids = np.zeros(k, np.int64)
    scores = np.zeros(k, np.float32)
    cnt = np.zeros(1, np.int32)
    sub = None if subset is None else np.ascontiguousarray(np.asarray(subset, dtype=np.int64).reshape(-1))
    if sub is not None and sub.size == 0:
        sub = np.zeros(1, np.int64)[:0]
    p = params._c()
    N.check(N.lib().fp_search_trace(
        index._h, _ptr(q), Q, D, C.byref(p), _ptr(sub) if sub is not None and sub.size else None,
        0 if sub is None else int(sub.shape[0]), 0 if sub is None else 1, _ptr(pids), _ptr(scores), _ptr(cnt), _ptr(S),
        _ptr(cells), _ptr(cand), _ptr(approx), _ptr(rer), _ptr(exact), _ptr(counts)))
    nc, ncand, nr = (int(x) for x in counts)
    n = int(cnt[0])
    return dict(S=S, cells=cells[:nc].copy(), cand=cand[:ncand].copy(), approx=approx[:ncand].copy(),
                rerank=rer[:nr].copy(), exact=exact[:nr].copy(), pids=pids[:n].copy(), scores=scores[:n].copy())


def last_search_timings() -> dict:
    names = (C.c_char_p * 16)()
    ms = (C.c_float * 16)()
    n = N.lib().fp_last_search_timings(C.cast(names, C.c_void_p), C.cast(ms, C.c_void_p), 16)
    return {names[i].decode(): float(ms[i]) for i in range(n)}


def set_graph_replay(enabled: bool) -> bool:
    """fp_set_graph_replay: HIP-graph replay of fp_search on / off (process-wide); returns the previous setting."""
    return bool(N.lib().fp_set_graph_replay(1 if enabled else 0))


def graph_replay_count() -> int:
    """fp_graph_replay_count: fp_search calls served by one hipGraphLaunch since the process started."""
    return int(N.lib().fp_graph_replay_count())


def maxsim_columns(index: PyLoadedIndex, query_f16, pids) -> dict:
    """fp_maxsim_columns: the exact stage's MFMA pass on `pids` for one query [q_len, dim], before the exact-order repair."""
    q = np.ascontiguousarray(query_f16, dtype=np.float16)
    p = np.ascontiguousarray(pids, dtype=np.int64)
    n, Q = int(p.shape[0]), int(q.shape[0])
    scores = np.zeros(n, np.float32)
    cm = np.zeros((n, Q), np.uint16)
    unc = np.zeros(n, np.float32)
    flags = np.zeros((n, (Q + 31) // 32), np.uint32)
    N.check(N.lib().fp_maxsim_columns(index._h, _ptr(q), Q, int(q.shape[1]), _ptr(p), n, _ptr(scores), _ptr(cm), _ptr(unc), _ptr(flags)))
    return dict(scores=scores, col_max=cm.view(np.float16), unc=unc, flags=flags)


def last_search_counts() -> dict:
    out = (C.c_int64 * 6)()
    N.lib().fp_last_search_counts(C.cast(out, C.c_void_p), 6)
    return dict(candidates=int(out[0]), approx_exact=int(out[1]), repaired=int(out[2]), sub_batches=int(out[3]),
                s4_form={0: "exact", 1: "q8", 2: "l0", 3: "l0h", -1: "replayed graph"}.get(int(out[4]), "?"),
                lazy_overflows=int(out[5]))


def last_s1_counts() -> dict:
    """fp_last_s1_counts (FP_S1_STATS in the environment): the certification counters of S1 -- or, when the last batch ran S1's
    lazy form (lazy == 1), flagged = entries the selection gathered, changed = maybes it recomputed from scratch.  lazy is
    reported without FP_S1_STATS too (-1: the batch was a replayed graph)."""
    out = (C.c_uint64 * 5)()
    N.lib().fp_last_s1_counts(C.cast(out, C.c_void_p), 5)
    lazy = int(out[4])
    return dict(flagged=int(out[0]), changed=int(out[1]), slow_path=int(out[2]), unflagged_differences=int(out[3]),
                lazy=lazy - (1 << 64) if lazy >= (1 << 63) else lazy)


def export_index_arrays(index: PyLoadedIndex, centroids=None, bucket_weights=None) -> dict:
    """construct_index argument set copied back from the device (fp_index_export)."""
    T, Nd, Cn = index.n_tokens, index.n_docs, index.n_centroids
    pr = index.dim * index.nbits // 8
    codes = np.zeros(max(T, 1), np.int64)[:T]
    res = np.zeros((max(T, 1), pr), np.uint8)[:T]
    lens = np.zeros(max(Nd, 1), np.int64)[:Nd]
    tot = int(N.lib().fp_index_ivf_total(index._h))
    ivf = np.zeros(max(tot, 1), np.int64)[:tot]
    ivfl = np.zeros(Cn, np.int32)
    N.check(N.lib().fp_index_export(index._h, _ptr(codes), _ptr(res), _ptr(lens), _ptr(ivf), _ptr(ivfl)))
    return dict(nbits=index.nbits, centroids=centroids, bucket_weights=bucket_weights, ivf=ivf, ivf_lengths=ivfl,
                doc_codes=codes, doc_residuals=res, doc_lengths=lens)


def reconstruct_embeddings(index: PyLoadedIndex, subset, device: str = "cuda"):
    """rust/utils/embeddings.rs:12-69 -> list of [doc_len, dim] float32 arrays."""
    del device
    ids = np.ascontiguousarray(np.asarray(subset, dtype=np.int64))
    n = int(ids.shape[0])
    lens = np.zeros(max(n, 1), np.int64)
    cap = index.n_tokens if n else 0
    # upper bound on rows: sum of requested doc lengths is unknown before the call -> two-step
    out = np.zeros((max(min(cap, 1 << 22), 1), index.dim), np.float32)
    rc = N.lib().fp_reconstruct_embeddings(index._h, _ptr(ids), n, _ptr(out), out.shape[0], _ptr(lens))
    if rc != 0 and "capacity" in N.last_error():
        # (the call has filled in every requested document's length before it looked at the capacity: ids may repeat, so the rows
        # needed can exceed the index's token count -- round 6's fuzz asked for six documents of a two-document index)
        out = np.zeros((max(int(lens[:n].sum()), 1), index.dim), np.float32)
        rc = N.lib().fp_reconstruct_embeddings(index._h, _ptr(ids), n, _ptr(out), out.shape[0], _ptr(lens))
    N.check(rc)
    res, o = [], 0
    for i in range(n):
        res.append(out[o: o + int(lens[i])].copy())
        o += int(lens[i])
    return res


def read_doc(index: PyLoadedIndex, local_doc: int, capacity: int = 4096):
    codes = np.zeros(capacity, np.int64)
    res = np.zeros((capacity, index.dim * index.nbits // 8), np.uint8)
    n = N.lib().fp_index_read_doc(index._h, int(local_doc), _ptr(codes), _ptr(res), capacity)
    if n < 0:
        raise ValueError(N.last_error())
    return codes[:n].copy(), res[:n].copy()


def read_ivf(index: PyLoadedIndex, cell: int, capacity: int = 1 << 22):
    pids = np.zeros(capacity, np.int64)
    n = N.lib().fp_index_read_ivf(index._h, int(cell), _ptr(pids), capacity)
    if n < 0:
        raise ValueError(N.last_error())
    return pids[:n].copy()


def create(index: str, torch_path: str, device: str, embedding_dim: int, nbits: int, embeddings, centroids, batch_size: int = 25_000,
           seed: int | None = None, compress_only: bool = False) -> None:
    """rust/lib.rs:132-165 (torch_path / batch_size accepted and ignored)."""
    del torch_path, batch_size
    from . import create as _create
    if embeddings and int(_np(embeddings[0], np.float16).shape[1]) != int(embedding_dim):
        raise ValueError("embedding_dim does not match the embeddings")
    _create.create_index(index, embeddings, centroids, nbits=nbits, device=device, seed=seed, compress_only=compress_only)


def update(index_path: str, index, torch_path: str, device: str, embeddings, batch_size: int = 25_000,
           update_threshold_centroids: bool | None = None) -> None:
    """rust/lib.rs:292-320 (the loaded index argument only carries the codec in the reference; it is read from the directory here)."""
    del index, torch_path, batch_size
    from . import maintain
    maintain.update_index(index_path, embeddings, device, update_threshold=bool(update_threshold_centroids))


def delete(index: str, torch_path: str, device: str, subset) -> None:
    """rust/lib.rs:322-340."""
    del torch_path, device
    from . import maintain
    maintain.delete_from_index(index, subset)


class QueryResultWithTokenScores:
    """rust/search/search.rs:128-169: a QueryResult plus, per hit, the [query_tokens, doc_tokens] fp16
    similarity matrix."""

    __slots__ = ("query_id", "passage_ids", "scores", "token_scores")

    def __init__(self, query_id, passage_ids, scores, token_scores):
        self.query_id = query_id
        self.passage_ids = passage_ids
        self.scores = scores
        self.token_scores = token_scores


def token_score_matrices(index: PyLoadedIndex, queries_embeddings, pids: np.ndarray, counts: np.ndarray) -> list[list[np.ndarray]]:
    """fp_token_scores: for every hit (query b, rank i < counts[b]) the [Q, doc_len] fp16 matrix."""
    q = _np(queries_embeddings, np.float16)
    if q.ndim != 3:
        raise ValueError(f"Expected a 3D tensor for queries, but got shape {tuple(q.shape)}")
    B, Q, D = q.shape
    pids = np.ascontiguousarray(pids, np.int64)
    counts = np.ascontiguousarray(counts, np.int32)
    stride = pids.shape[1] if pids.ndim == 2 else 0
    n_hits = int(counts.sum())
    offs = np.zeros(n_hits + 1, np.int64)
    N.check(N.lib().fp_token_scores(index._h, _ptr(q), B, Q, D, _ptr(pids), _ptr(counts), stride, _ptr(offs), None, 0))
    out = np.zeros(max(int(offs[-1]), 1), np.float16)
    if n_hits and offs[-1] > 0:
        N.check(N.lib().fp_token_scores(index._h, _ptr(q), B, Q, D, _ptr(pids), _ptr(counts), stride, _ptr(offs), _ptr(out), int(offs[-1])))
    res, h = [], 0
    for b in range(B):
        row = []
        for _ in range(int(counts[b])):
            n = int(offs[h + 1] - offs[h])
            row.append(out[offs[h]: offs[h + 1]].reshape(Q, n // Q).copy())
            h += 1
        res.append(row)
    return res


def pysearch_with_token_scores(index: PyLoadedIndex, device: str, queries_embeddings, search_parameters: SearchParameters,
                               show_progress: bool = False, subset=None) -> list[QueryResultWithTokenScores]:
    """rust/lib.rs:243-290 -> search_many_with_token_scores (search.rs:294-363)."""
    del show_progress
    if _device_id(device) != index.device_id:
        raise ValueError(f"index lives on device {index.device_id}, search requested on '{device}'")
    pids, scores, counts = search_arrays(index, queries_embeddings, search_parameters, subset)
    mats = token_score_matrices(index, queries_embeddings, pids, counts)
    return [QueryResultWithTokenScores(b, pids[b, : counts[b]].tolist(), scores[b, : counts[b]].tolist(), mats[b])
            for b in range(pids.shape[0])]
