"""Mirror of ``fast_plaid.search.FastPlaid`` (python/fast_plaid/search/fast_plaid.py:325-1186)
for the search path on MI355X.

Same constructor and ``search`` signature/semantics as the reference:
  FastPlaid(index, device=None, low_memory=True)                       :328-385
  search(queries_embeddings, top_k=10, batch_size=2000, n_full_scores=4096, n_ivf_probe=8,
         show_progress=True, subset=None, n_processes=None)            :931-983
    -> list[list[tuple[int, float]]]
Queries may be a [B,Q,D] tensor/array or a list of [Q_i,D] (zero-padded to the longest,
:772-780); `subset` a list[int] (broadcast) or list[list[int]] (:784-793).  Several devices =
full replica per device + query split in threads (:893-928), exactly the reference's
multi-GPU behaviour; the document-sharded mode lives in fast_plaid_amd.sharded.
``create``/``update``/``delete`` (index maintenance) are not part of the search hot path.
"""
from __future__ import annotations

import math
import os
from concurrent.futures import ThreadPoolExecutor
from typing import Any

import numpy as np

from .. import _native as N
from .. import fast_plaid_rust as native
from .index_io import load_index_arrays


def _to_np16(x) -> np.ndarray:
    if hasattr(x, "detach"):
        x = x.detach().cpu()
        if str(x.dtype) != "torch.float16":
            x = x.to(dtype=__import__("torch").float16)  # fast_plaid.py:241
        return np.ascontiguousarray(x.numpy())
    return np.ascontiguousarray(np.asarray(x), dtype=np.float16)


class FastPlaid:
    def __init__(self, index: str, device: str | list[str] | None = None, low_memory: bool = True, **kwargs: Any) -> None:
        del kwargs
        if device is not None and isinstance(device, str):
            self.devices = [device]
        elif isinstance(device, list):
            self.devices = device
        else:
            n = N.lib().fp_device_count()
            if n < 1:
                raise RuntimeError("no MI355X device visible (this build has no CPU path)")
            self.devices = [f"cuda:{i}" for i in range(n)]
        self.devices = ["cuda:0" if d == "cuda" else d for d in self.devices]  # :358-359
        self.devices = list(dict.fromkeys(self.devices))  # :362
        self.index = index
        self.low_memory = low_memory
        self.indices: dict[str, Any] = {}
        self._last_known_mtime = 0.0
        if index is not None and os.path.isdir(index):
            self._check_and_reload_index()

    # ---- construction --------------------------------------------------------------------
    @classmethod
    def from_arrays(cls, arrays: dict, device: str | list[str] = "cuda:0") -> "FastPlaid":
        """Build directly from the construct_index argument set (no directory)."""
        self = cls(index=None, device=device)  # type: ignore[arg-type]
        self._load_arrays(arrays)
        return self

    def _load_arrays(self, arrays: dict) -> None:
        def one(dev):
            return dev, native.construct_index(
                nbits=arrays["nbits"], centroids=arrays["centroids"], avg_residual=arrays.get("avg_residual"),
                bucket_cutoffs=arrays.get("bucket_cutoffs"), bucket_weights=arrays["bucket_weights"],
                ivf=arrays.get("ivf"), ivf_lengths=arrays.get("ivf_lengths"), doc_codes=arrays["doc_codes"],
                doc_residuals=arrays["doc_residuals"], doc_lengths=arrays["doc_lengths"], device=dev,
                low_memory=self.low_memory)
        if len(self.devices) == 1:
            d, idx = one(self.devices[0])
            self.indices = {d: idx}
        else:  # load.py:419-424
            with ThreadPoolExecutor(max_workers=len(self.devices)) as ex:
                self.indices = dict(ex.map(one, self.devices))

    def _check_and_reload_index(self) -> bool:
        """mtime-based reload (fast_plaid.py:433-479), without the cross-process file lock."""
        meta = os.path.join(self.index, "metadata.json")
        if not os.path.exists(meta):
            self.indices = {d: None for d in self.devices}
            return True
        m = os.stat(meta).st_mtime
        if m <= self._last_known_mtime and any(v is not None for v in self.indices.values()):
            return True
        arrays = load_index_arrays(self.index)
        self._load_arrays(arrays)
        self._last_known_mtime = m
        return True

    def close(self) -> None:  # :387-396
        for v in self.indices.values():
            if v is not None:
                v.close()
        self.indices.clear()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- search ----------------------------------------------------------------------------
    def _prepare_search(self, queries_embeddings, subset):
        """fast_plaid.py:743-795."""
        if self.index is not None and os.path.isdir(self.index):
            self._check_and_reload_index()
            if not os.path.exists(os.path.join(self.index, "metadata.json")):
                raise FileNotFoundError(f"Index metadata not found in '{self.index}'. Please create the index before searching.")
        for d in self.devices:
            if self.indices.get(d) is None:
                raise RuntimeError(f"Index could not be loaded on device '{d}'.")
        if isinstance(queries_embeddings, (list, tuple)):  # :772-780 pad_sequence(padding_value=0.0)
            qs = [_to_np16(q[0] if getattr(q, "ndim", 2) == 3 else q) for q in queries_embeddings]
            L = max(q.shape[0] for q in qs)
            out = np.zeros((len(qs), L, qs[0].shape[1]), np.float16)
            for i, q in enumerate(qs):
                out[i, : q.shape[0]] = q
            q3 = out
        else:
            q3 = _to_np16(queries_embeddings)
        nq = q3.shape[0]
        if subset is not None:  # :784-793
            if isinstance(subset, int):
                subset = [subset] * nq
            if isinstance(subset, list) and len(subset) == 0:
                subset = None
            if isinstance(subset, list) and isinstance(subset[0], (int, np.integer)):
                subset = [subset] * nq
            if subset is not None and len(subset) != nq:
                raise ValueError("Subset length must match number of queries.")
        return q3, subset

    def search(self, queries_embeddings, top_k: int = 10, batch_size: int = 2000, n_full_scores: int = 4096,
               n_ivf_probe: int = 8, show_progress: bool = True, subset=None, n_processes: int | None = None):
        del n_processes  # CPU-only joblib knob in the reference (:841-878)
        q3, subset = self._prepare_search(queries_embeddings, subset)
        params = native.SearchParameters(batch_size=batch_size, n_full_scores=n_full_scores, top_k=top_k, n_ivf_probe=n_ivf_probe)

        def on_device(dev, q, sub):  # search_on_device :188-253
            res = native.pysearch(self.indices[dev], dev, q, params, show_progress, sub)
            return [[(pid, sc) for sc, pid in zip(r.scores, r.passage_ids)] for r in res]

        nq = q3.shape[0]
        if len(self.devices) == 1 or nq == 0:
            return on_device(self.devices[0], q3, subset)
        chunk = math.ceil(nq / len(self.devices))  # :893-928
        jobs = []
        with ThreadPoolExecutor(max_workers=len(self.devices)) as ex:
            for i, dev in enumerate(self.devices):
                s, e = i * chunk, min(nq, (i + 1) * chunk)
                if s >= e:
                    break
                jobs.append(ex.submit(on_device, dev, q3[s:e], None if subset is None else subset[s:e]))
        out = []
        for j in jobs:
            out.extend(j.result())
        return out

    def search_token_scores(self, queries_embeddings, top_k: int = 10, batch_size: int = 2000, n_full_scores: int = 4096,
                            n_ivf_probe: int = 8, show_progress: bool = True, subset=None, n_processes: int | None = None):
        """fast_plaid.py:986-1047: like search(), each hit is (doc_id, score, [query_tokens, doc_tokens] fp16 matrix)."""
        del n_processes
        q3, subset = self._prepare_search(queries_embeddings, subset)
        params = native.SearchParameters(batch_size=batch_size, n_full_scores=n_full_scores, top_k=top_k, n_ivf_probe=n_ivf_probe)
        out = []
        nq = q3.shape[0]
        chunk = max(1, math.ceil(nq / len(self.devices))) if nq else 1
        for i, dev in enumerate(self.devices):  # search_on_device_with_token_scores :256-322
            s, e = i * chunk, min(nq, (i + 1) * chunk)
            if s >= e:
                break
            res = native.pysearch_with_token_scores(self.indices[dev], dev, q3[s:e], params, show_progress,
                                                    None if subset is None else subset[s:e])
            out.extend([[(pid, sc, m) for sc, pid, m in zip(r.scores, r.passage_ids, r.token_scores)] for r in res])
        return out

    def get_embeddings(self, subset: list[int]):
        """fast_plaid.py:1160-1186 -> list of [doc_len, dim] float32 arrays."""
        return native.reconstruct_embeddings(self.indices[self.devices[0]], subset, self.devices[0])

    def create(self, documents_embeddings, kmeans_niters: int = 4, max_points_per_centroid: int = 256, nbits: int = 4,
               n_samples_kmeans: int | None = None, seed: int = 42, use_triton_kmeans: bool | None = None, metadata=None,
               compress_only: bool = False, centroids=None):
        """fast_plaid.py:398-560: k-means centroids (unless given), then the native part of the reference's create --
        codec training, compression, IVF, directory (rust/index/create.rs)."""
        del use_triton_kmeans
        if metadata is not None:
            raise NotImplementedError("metadata filtering (fast_plaid.filtering) is outside the MI355X search hot path")
        if centroids is None:   # fast_plaid.py:71-185 compute_kmeans (functional equivalent, see kmeans.py)
            from .. import kmeans as _kmeans
            first = documents_embeddings[0]
            centroids = _kmeans.compute_kmeans(documents_embeddings, int(first.shape[1]), self.devices[0], kmeans_niters,
                                               max_points_per_centroid, seed, n_samples_kmeans)
        if self.index is None:
            raise ValueError("FastPlaid.create needs an index directory")
        from .. import create as _create
        _create.create_index(self.index, documents_embeddings, centroids, nbits=nbits, device=self.devices[0], seed=seed,
                             compress_only=compress_only)
        self._last_known_mtime = 0.0
        self._check_and_reload_index()
        return self

    def update(self, documents_embeddings, metadata=None, batch_size: int = 25_000, update_threshold_centroids: bool = False, **kwargs):
        """fast_plaid.py update -> rust/index/update.rs: append documents with the index's existing codec.  (The reference's
        Python layer additionally buffers small updates and may re-cluster, which needs the external usearch package;
        here every call appends directly.)"""
        del batch_size, kwargs
        if metadata is not None:
            raise NotImplementedError("metadata filtering (fast_plaid.filtering) is outside the MI355X search hot path")
        if self.index is None or not os.path.exists(os.path.join(self.index, "metadata.json")):
            raise FileNotFoundError("no index to update: create it first")
        from .. import maintain
        maintain.update_index(self.index, documents_embeddings, self.devices[0], update_threshold=update_threshold_centroids)
        self._last_known_mtime = 0.0
        self._check_and_reload_index()
        return self

    def delete(self, subset: list[int], **kwargs):
        """fast_plaid.py:1049-1100 -> rust/index/delete.rs: documents are addressed by position; survivors are renumbered."""
        del kwargs
        if self.index is None or not os.path.exists(os.path.join(self.index, "metadata.json")):
            raise FileNotFoundError("no index to delete from")
        from .. import maintain
        maintain.delete_from_index(self.index, subset)
        self._last_known_mtime = 0.0
        self._check_and_reload_index()
        return self
