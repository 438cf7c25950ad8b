"""On-disk index directory reader/writer, format of the reference (rust/index/create.rs:380-397,
:476-491, :548-582; python/fast_plaid/search/load.py:220-322):

  metadata.json {num_chunks, nbits, num_partitions, num_embeddings, avg_doclen, num_documents, compress_only}
  centroids.npy (f16)  bucket_weights.npy / bucket_cutoffs.npy / avg_residual.npy (f32 or f16)
  {i}.codes.npy (i64)  {i}.residuals.npy (u8 [T_i, dim*nbits/8])  doclens.{i}.json
  ivf.npy (i64)  ivf_lengths.npy (i32)          (absent for compress_only indexes)
  cluster_threshold.npy (f32 scalar, create.rs:331-339)  plan.json {nbits, num_chunks} (create.rs:296-299)

The reference merges the chunk files into one padded mmap before construct_index
(load.py:35-217); here the chunks are simply concatenated -- the device index owns its copy.
"""
from __future__ import annotations

import json
import os

import numpy as np


def load_index_arrays(index_path: str) -> dict | None:
    meta_path = os.path.join(index_path, "metadata.json")
    if not os.path.exists(meta_path):
        return None
    with open(meta_path) as f:
        meta = json.load(f)
    n_chunks = int(meta["num_chunks"])

    def small(name, dtype):
        p = os.path.join(index_path, name)
        if not os.path.exists(p):
            raise FileNotFoundError(f"Missing index file: {p}")
        return np.ascontiguousarray(np.load(p).astype(dtype))

    data = dict(
        nbits=int(meta["nbits"]),
        centroids=small("centroids.npy", np.float16),
        avg_residual=small("avg_residual.npy", np.float16),
        bucket_cutoffs=small("bucket_cutoffs.npy", np.float16),
        bucket_weights=small("bucket_weights.npy", np.float16),
    )
    ivf_p, ivfl_p = os.path.join(index_path, "ivf.npy"), os.path.join(index_path, "ivf_lengths.npy")
    if os.path.exists(ivf_p) and os.path.exists(ivfl_p):
        data["ivf"] = small("ivf.npy", np.int64)
        data["ivf_lengths"] = small("ivf_lengths.npy", np.int32)
    else:
        data["ivf"] = None
        data["ivf_lengths"] = None
    doclens, codes, res = [], [], []
    for i in range(n_chunks):
        dl = os.path.join(index_path, f"doclens.{i}.json")
        if os.path.exists(dl):
            with open(dl) as f:
                doclens.extend(json.load(f))
        cp = os.path.join(index_path, f"{i}.codes.npy")
        rp = os.path.join(index_path, f"{i}.residuals.npy")
        if os.path.exists(cp):
            codes.append(np.load(cp, mmap_mode="r"))
            res.append(np.load(rp, mmap_mode="r"))
    dim = int(data["centroids"].shape[1])
    data["doc_lengths"] = np.asarray(doclens, dtype=np.int64)
    data["doc_codes"] = (np.concatenate(codes).astype(np.int64) if codes else np.zeros(0, np.int64))
    data["doc_residuals"] = (np.concatenate(res).astype(np.uint8) if res
                             else np.zeros((0, dim * data["nbits"] // 8), np.uint8))
    return data


def save_index_arrays(index_path: str, arr: dict, chunk_docs: int = 25_000, write_plan: bool = False) -> None:
    """Writes arrays in the reference's directory format (used by tests and by tools that
    hand an MI355X-built index back to the reference).  `cluster_threshold` (when the array set carries one) goes to
    cluster_threshold.npy as the reference's 0-dim f32; write_plan adds create.rs's plan.json."""
    os.makedirs(index_path, exist_ok=True)
    lens = np.asarray(arr["doc_lengths"], dtype=np.int64)
    T = int(lens.sum())
    codes = np.asarray(arr["doc_codes"])[:T]
    res = np.asarray(arr["doc_residuals"])[:T]
    np.save(os.path.join(index_path, "centroids.npy"), np.asarray(arr["centroids"], dtype=np.float16))
    f32 = arr.get("codec_f32") or {}   # create.py: the trained fp32 values (create.rs:380-397 writes those, not their fp16 roundings)
    for k in ("bucket_weights", "bucket_cutoffs", "avg_residual"):
        np.save(os.path.join(index_path, k + ".npy"), np.asarray(f32.get(k, arr[k])).astype(np.float32))
    n_docs = int(lens.shape[0])
    n_chunks = max(1, (n_docs + chunk_docs - 1) // chunk_docs)
    offs = np.concatenate([[0], np.cumsum(lens)])
    for i in range(n_chunks):
        d0, d1 = i * chunk_docs, min(n_docs, (i + 1) * chunk_docs)
        t0, t1 = int(offs[d0]), int(offs[d1])
        np.save(os.path.join(index_path, f"{i}.codes.npy"), codes[t0:t1].astype(np.int64))
        np.save(os.path.join(index_path, f"{i}.residuals.npy"), res[t0:t1].astype(np.uint8))
        with open(os.path.join(index_path, f"doclens.{i}.json"), "w") as f:
            json.dump(lens[d0:d1].tolist(), f)
        with open(os.path.join(index_path, f"{i}.metadata.json"), "w") as f:
            json.dump({"num_documents": d1 - d0, "num_embeddings": t1 - t0, "embedding_offset": t0}, f)
    if arr.get("cluster_threshold") is not None:
        np.save(os.path.join(index_path, "cluster_threshold.npy"), np.asarray(arr["cluster_threshold"], dtype=np.float32).reshape(()))
    if write_plan:   # create.rs:296-299 (serde_json pretty print + newline)
        with open(os.path.join(index_path, "plan.json"), "w") as f:
            f.write(json.dumps({"nbits": int(arr["nbits"]), "num_chunks": n_chunks}, indent=2) + "\n")
    compress_only = arr.get("ivf") is None
    if not compress_only:
        np.save(os.path.join(index_path, "ivf.npy"), np.asarray(arr["ivf"], dtype=np.int64))
        np.save(os.path.join(index_path, "ivf_lengths.npy"), np.asarray(arr["ivf_lengths"], dtype=np.int32))
    with open(os.path.join(index_path, "metadata.json"), "w") as f:
        json.dump({"num_chunks": n_chunks, "nbits": int(arr["nbits"]),
                   "num_partitions": int(arr["num_partitions"]) if arr.get("num_partitions") is not None
                   else (0 if compress_only else int(np.asarray(arr["ivf_lengths"]).shape[0])),
                   "num_embeddings": T, "avg_doclen": (T / n_docs) if n_docs else 0.0, "num_documents": n_docs,
                   "compress_only": bool(compress_only)}, f)
