"""ORACLE wrapper (test infrastructure; see oracle/plaid_oracle.c for scope and pinning).

ctypes binding of the plain-C restatement of the reference search path.  numpy only.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libplaid_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "plaid_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libplaid_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
        L.pl_index_create.restype = vp
        L.pl_index_create.argtypes = [i32, i64, i32, vp, vp, vp, vp, i64, vp, vp, vp, i64]
        L.pl_index_destroy.argtypes = [vp]
        L.pl_decompress.argtypes = [vp, vp, vp, i64, vp]
        L.pl_search.restype = i32
        L.pl_search.argtypes = [vp, vp, i32, i32, i64, i64, i64, vp, vp, vp, vp, vp, i32]
        L.pl_search_trace.restype = i64
        L.pl_search_trace.argtypes = [vp, vp, i32, i64, i64, i64, vp, i64, i32] + [vp] * 9
        L.pl_exact_scores.argtypes = [vp, vp, i32, vp, i64, vp]
        L.pl_token_scores.argtypes = [vp, vp, i32, i64, vp]
        L.pl_column_maxima.argtypes = [vp, vp, i32, vp, i64, vp]
        L.pl_num_procs.restype = i32
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleIndex:
    """Mirror of the reference's construct_index argument set (rust/search/load.rs:124-138)."""

    def __init__(self, nbits, centroids, bucket_weights, ivf, ivf_lengths, doc_codes, doc_residuals,
                 doc_lengths, avg_residual=None, bucket_cutoffs=None):
        self.nbits = int(nbits)
        self.centroids = np.ascontiguousarray(centroids, dtype=np.float16)
        self.bucket_weights = np.ascontiguousarray(bucket_weights, dtype=np.float16)
        self.ivf = None if ivf is None else np.ascontiguousarray(ivf, dtype=np.int64)
        self.ivf_lengths = None if ivf_lengths is None else np.ascontiguousarray(ivf_lengths, dtype=np.int32)
        self.doc_codes = np.ascontiguousarray(doc_codes, dtype=np.int64)
        self.doc_residuals = np.ascontiguousarray(doc_residuals, dtype=np.uint8)
        self.doc_lengths = np.ascontiguousarray(doc_lengths, dtype=np.int64)
        self.dim = int(self.centroids.shape[1])
        self.n_docs = int(self.doc_lengths.shape[0])
        P = 0 if self.ivf_lengths is None else int(self.ivf_lengths.shape[0])
        self._h = lib().pl_index_create(
            self.nbits, self.centroids.shape[0], self.dim, _p(self.centroids), _p(self.bucket_weights),
            _p(self.ivf), _p(self.ivf_lengths), P, _p(self.doc_codes), _p(self.doc_residuals),
            _p(self.doc_lengths), self.n_docs)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().pl_index_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ #
    def search(self, queries, top_k=10, n_full_scores=4096, n_ivf_probe=8, subset=None, nthreads=1):
        """queries [B,Q,D] f16 -> list of (pids int64[], scores f32[]) per query."""
        q = np.ascontiguousarray(queries, dtype=np.float16)
        if q.ndim != 3:
            raise ValueError(f"Expected a 3D tensor for queries, but got shape {list(q.shape)}")
        B, Q, _ = q.shape
        pids = np.zeros((B, max(top_k, 1)), dtype=np.int64)
        scores = np.zeros((B, max(top_k, 1)), dtype=np.float32)
        counts = np.zeros(B, dtype=np.int32)
        sub_ids = sub_off = None
        if subset is not None:
            sub_off = np.zeros(B + 1, dtype=np.int64)
            for i, s in enumerate(subset):
                sub_off[i + 1] = sub_off[i] + len(s)
            sub_ids = np.ascontiguousarray(np.concatenate([np.asarray(s, dtype=np.int64) for s in subset])
                                           if sub_off[-1] else np.zeros(1, np.int64))
        rc = lib().pl_search(self._h, _p(q), B, Q, n_ivf_probe, n_full_scores, top_k, _p(sub_ids), _p(sub_off),
                             _p(pids), _p(scores), _p(counts), nthreads)
        if rc == -2:
            raise ValueError("This index was built with compress_only=True and does not support search. "
                             "Rebuild with compress_only=False to enable search.")
        return [(pids[b, : counts[b]].copy(), scores[b, : counts[b]].copy()) for b in range(B)]

    def search_trace(self, query, top_k=10, n_full_scores=4096, n_ivf_probe=8, subset=None):
        """One query [Q,D] f16 with every stage output (dict)."""
        q = np.ascontiguousarray(query, dtype=np.float16)
        Q = q.shape[0]
        Cn, N = self.centroids.shape[0], self.n_docs
        S = np.zeros((Cn, Q), np.float16)
        cells = np.zeros(max(Q * max(n_ivf_probe, 1), 1), np.int64)
        cand = np.zeros(max(N, 1), np.int64)
        approx = np.zeros(max(N, 1), np.float32)
        rer = np.zeros(max(N, 1), np.int64)
        exact = np.zeros(max(N, 1), np.float32)
        counts = np.zeros(3, np.int64)
        pids = np.zeros(max(top_k, 1), np.int64)
        scores = np.zeros(max(top_k, 1), np.float32)
        sub = None if subset is None else np.ascontiguousarray(subset, dtype=np.int64)
        r = lib().pl_search_trace(self._h, _p(q), Q, n_ivf_probe, n_full_scores, top_k, _p(sub),
                                  0 if sub is None else sub.shape[0], 0 if sub is None else 1, _p(pids), _p(scores),
                                  _p(S), _p(cells), _p(cand), _p(approx), _p(rer), _p(exact), _p(counts))
        r = max(int(r), 0)
        nc, ncand, nr = (int(x) for x in counts)
        return dict(S=S, cells=cells[:nc].copy(), cand=cand[:ncand].copy(), approx=approx[:ncand].copy(),
                    rerank=rer[:nr].copy(), exact=exact[:nr].copy(), pids=pids[:r].copy(), scores=scores[:r].copy())

    def decompress(self, codes, residuals):
        codes = np.ascontiguousarray(codes, np.int64)
        res = np.ascontiguousarray(residuals, np.uint8)
        out = np.zeros((codes.shape[0], self.dim), np.float16)
        lib().pl_decompress(self._h, _p(codes), _p(res), codes.shape[0], _p(out))
        return out

    def exact_scores(self, query, pids):
        q = np.ascontiguousarray(query, np.float16)
        p = np.ascontiguousarray(pids, np.int64)
        out = np.zeros(p.shape[0], np.float32)
        lib().pl_exact_scores(self._h, _p(q), q.shape[0], _p(p), p.shape[0], _p(out))
        return out


def _column_maxima(self, query, pids):
    """[n, Q] fp16: per document the column maxima of its token-score matrix (what exact_scores sums)."""
    q = np.ascontiguousarray(query, np.float16)
    p = np.ascontiguousarray(pids, np.int64)
    out = np.zeros((p.shape[0], q.shape[0]), np.float16)
    lib().pl_column_maxima(self._h, _p(q), q.shape[0], _p(p), p.shape[0], _p(out))
    return out


OracleIndex.column_maxima = _column_maxima


def _token_scores(self, query, pid):
    """[Q, len(doc)] fp16 similarity matrix (search.rs:651-653, :668-686)."""
    q = np.ascontiguousarray(query, np.float16)
    ln = int(self.doc_lengths[int(pid)])
    out = np.zeros((q.shape[0], ln), np.float16)
    lib().pl_token_scores(self._h, _p(q), q.shape[0], int(pid), _p(out))
    return out


OracleIndex.token_scores = _token_scores


def num_procs() -> int:
    return int(lib().pl_num_procs())
