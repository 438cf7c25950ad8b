"""K-means centroids for index creation: the protocol of python/fast_plaid/search/fast_plaid.py:71-185
(compute_kmeans: sample 1 + 16*sqrt(120*N) documents, K = 2^floor(log2(16*sqrt(estimated tokens))), Lloyd
iterations, L2-normalise, fp16) with the assignment step on the device (fp_assign_l2).

The reference delegates the iterations to the external `fastkmeans` package (or its own torch fallback,
kmeans.py:61-223) with torch's RNG; this is a functional equivalent with numpy's generator -- same protocol and
arithmetic shape, not the same random stream, so centroids are not comparable value for value.
"""
from __future__ import annotations

import math

import numpy as np

from . import _native as N
from .fast_plaid_rust import _device_id, _np, _ptr


def assign_l2(centroids_f16: np.ndarray, data_f16: np.ndarray, device: str = "cuda:0") -> np.ndarray:
    """label[t] = argmin_c ||x_t - c||^2 = argmax_c (x_t . c - ||c||^2 / 2), fp32, ties -> lowest index."""
    cent = _np(centroids_f16, np.float16)
    data = _np(data_f16, np.float16)
    hn = (0.5 * (cent.astype(np.float32) ** 2).sum(axis=1, dtype=np.float32)).astype(np.float32)
    labels = np.zeros(data.shape[0], np.int64)
    N.check(N.lib().fp_assign_l2(_device_id(device), _ptr(cent), _ptr(hn), cent.shape[0], cent.shape[1], _ptr(data), data.shape[0], _ptr(labels)))
    return labels


def lloyd(data_f16: np.ndarray, k: int, niter: int, rng: np.random.Generator, device: str = "cuda:0",
          max_points_per_centroid: int | None = 256) -> np.ndarray:
    """kmeans.py:61-223: subsample to k*max_points_per_centroid, random-point init, `niter` Lloyd iterations, empty
    clusters re-seeded from random points.  Returns float32 [k, dim]."""
    data = _np(data_f16, np.float16)
    n, dim = data.shape
    if max_points_per_centroid is not None and n > k * max_points_per_centroid:
        data = data[rng.permutation(n)[: k * max_points_per_centroid]]
        n = data.shape[0]
    if n < k:
        raise ValueError(f"Number of training points ({n}) is less than k ({k}).")
    cent = data[rng.permutation(n)[:k]].astype(np.float32)
    for _ in range(max(niter, 0)):
        labels = assign_l2(cent.astype(np.float16), data, device)
        order = np.argsort(labels, kind="stable")
        sl = labels[order]
        starts = np.flatnonzero(np.r_[True, sl[1:] != sl[:-1]])
        sums = np.add.reduceat(data[order].astype(np.float32), starts, axis=0)
        counts = np.diff(np.r_[starts, n]).astype(np.float32)
        new = np.zeros_like(cent)
        ids = sl[starts]
        new[ids] = sums / counts[:, None]
        empty = np.setdiff1d(np.arange(k), ids, assume_unique=True)
        if empty.size:
            new[empty] = data[rng.integers(0, n, empty.size)].astype(np.float32)
        cent = new
    return cent


def compute_kmeans(documents_embeddings, dim: int, device: str = "cuda:0", kmeans_niters: int = 4, max_points_per_centroid: int = 256,
                   seed: int = 42, n_samples_kmeans: int | None = None, num_partitions: int | None = None) -> np.ndarray:
    """fast_plaid.py:71-185 -> fp16 [K, dim] unit-norm centroids."""
    docs = [_np(d, np.float16) for d in documents_embeddings]
    n_docs = len(docs)
    if n_docs == 0:
        raise ValueError("no documents")
    rng = np.random.default_rng(seed)
    if n_samples_kmeans is None:
        n_samples_kmeans = min(1 + int(16 * math.sqrt(120 * n_docs)), n_docs)
    n_samples_kmeans = min(n_docs, n_samples_kmeans)
    picked = rng.permutation(n_docs)[:n_samples_kmeans]
    samples = np.concatenate([docs[i] for i in picked]).astype(np.float16)
    if samples.shape[1] != dim:
        raise ValueError("embedding dim mismatch")
    total = samples.shape[0]
    if num_partitions is None:
        est = total / n_samples_kmeans * n_docs
        num_partitions = int(2 ** math.floor(math.log2(16 * math.sqrt(est))))
    k = min(num_partitions, total)
    cent = lloyd(samples, k, kmeans_niters, rng, device, max_points_per_centroid)
    nrm = np.linalg.norm(cent, axis=1, keepdims=True)
    return (cent / np.maximum(nrm, 1e-12)).astype(np.float16)
