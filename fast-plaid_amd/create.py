"""Index creation, the part the reference does in its native `create` (rust/lib.rs:132-165 ->
rust/index/create.rs:206-583) -- SURVEY.md section 8, row f1:

  codec training   create.rs:317-377   bucket cutoffs / weights = quantiles of held-out residuals  (host, numpy)
  compression      create.rs:148-184, :404-428   nearest centroid + quantised packed residuals      (device, fp_compress)
  IVF              create.rs:528-559 + optimize_ivf :55-132                                         (host, numpy)
  directory        create.rs:380-397, :476-491, :548-582                                            (search/index_io.py)

K-means is not here: the reference runs it in Python through the external `fastkmeans` package
(python/fast_plaid/search/kmeans.py), which is not part of this image; centroids are an input.
The sampling of held-out passages follows create.rs:222-281 (shuffle, take 16*sqrt(120 N) passages, walk the SHUFFLED
sample from its end until 5 % of its tokens -- at most 50 000 -- are collected) but draws the shuffle from numpy's
generator instead of Rust's StdRng, so a directory created here is a valid index for the same centroids but not
byte-identical to one the reference would create from the same seed; every deterministic step is bit-identical to
the ATen restatement (tests).  The directory holds every file the reference writes, including plan.json (:296-299) and
cluster_threshold.npy (:331-339), which the reference's update path loads unconditionally (python update.py:365-366).
"""
from __future__ import annotations

import math

import numpy as np

from . import _native as N
from . import synth
from .fast_plaid_rust import _device_id, _np, _ptr


def compress(centroids, bucket_cutoffs, embeddings, nbits: int, device: str = "cuda:0"):
    """create.rs:404-428 on the device: (codes int64 [T], packed residuals uint8 [T, dim*nbits/8])."""
    cent = _np(centroids, np.float16)
    cut = _np(bucket_cutoffs, np.float16)
    emb = _np(embeddings, np.float16)
    if emb.ndim != 2 or cent.ndim != 2 or emb.shape[1] != cent.shape[1]:
        raise ValueError("embeddings [T, dim] and centroids [C, dim] must agree on dim")
    if cut.shape[0] != (1 << nbits) - 1:
        raise ValueError("bucket_cutoffs must hold 2^nbits - 1 values")
    T, dim = emb.shape
    codes = np.zeros(T, np.int64)
    res = np.zeros((T, dim * nbits // 8), np.uint8)
    N.check(N.lib().fp_compress(_device_id(device), _ptr(cent), cent.shape[0], dim, nbits, _ptr(cut), _ptr(emb), T, _ptr(codes), _ptr(res)))
    return codes, res


def cutoffs_for_f32_compare(cutoffs_f32) -> np.ndarray:
    """create.rs:413 compares the fp16 residuals with the FLOAT cutoffs it has just computed (torch.bucketize promotes), while the
    device kernel (and update.rs, through the loader's Half cast) compares with fp16 cutoffs.  For an fp16 residual r and an fp32
    cutoff c, (c < r) == (c_dn < r) with c_dn = the largest fp16 value <= c: no fp16 value lies strictly between c_dn and c.  So
    the cutoffs rounded DOWN to fp16 reproduce the reference's first-time compression exactly."""
    c = np.asarray(cutoffs_f32, np.float32)
    h = c.astype(np.float16)
    up = h.astype(np.float32) > c          # round-to-nearest went up: step one fp16 value down
    return np.where(up, np.nextafter(h, np.float16(-np.inf)), h).astype(np.float16)


def _kth(sorted_flat: np.ndarray, k: int) -> np.float32:
    return sorted_flat[k]


def _quantile(sorted_flat: np.ndarray, q: float) -> np.float32:
    """rust/search/tensor.rs:18-34 scalar_quantile_kthvalue; lerp as ATen computes it in fp32
    (weight < 0.5 ? a + w (b - a) : b - (b - a)(1 - w))."""
    n = sorted_flat.shape[0]
    idx = q * (n - 1)
    lo, hi = math.floor(idx), math.ceil(idx)
    a = np.float32(_kth(sorted_flat, lo))
    if lo == hi:
        return a
    b = np.float32(_kth(sorted_flat, hi))
    w = np.float32(idx - lo)
    d = np.float32(b - a)
    return np.float32(a + w * d) if w < np.float32(0.5) else np.float32(b - d * np.float32(np.float32(1.0) - w))


def train_codec(heldout, centroids, nbits: int, device: str = "cuda:0", with_threshold: bool = False):
    """create.rs:317-364 -> (bucket_cutoffs f32 [2^nbits-1], bucket_weights f32 [2^nbits], avg_residual f32 [dim]);
    with_threshold: also the cluster threshold of create.rs:331-334, the 0.75 quantile of the held-out residual norms."""
    held = _np(heldout, np.float16)
    cent = _np(centroids, np.float16)
    n_opt = 1 << nbits
    codes, _ = compress(cent, np.zeros(n_opt - 1, np.float16), held, nbits, device)   # only the assignment is used
    res = (held - cent[codes]).astype(np.float32)      # fp16 subtraction (fp32 op + one rounding), then widened like .to(Float)
    flat = np.sort(res.reshape(-1), kind="stable")
    cut = np.array([_quantile(flat, i / n_opt) for i in range(1, n_opt)], np.float32)
    wts = np.array([_quantile(flat, (i + 0.5) / n_opt) for i in range(n_opt)], np.float32)
    avg = np.abs(res).mean(axis=0, dtype=np.float32)
    if with_threshold:
        return cut, wts, avg, cluster_threshold(res)
    return cut, wts, avg


def cluster_threshold(residuals_f32: np.ndarray) -> np.float32:
    """create.rs:331-334: norm_scalaropt_dim(2, [1]) of the fp32 held-out residuals, then the 0.75 quantile."""
    r = np.asarray(residuals_f32, np.float32)
    dist = np.sqrt(np.einsum("ij,ij->i", r, r, dtype=np.float32), dtype=np.float32)
    return _quantile(np.sort(dist, kind="stable"), 0.75)


def heldout_sample(docs16, rng) -> np.ndarray:
    """create.rs:222-281: shuffle the passage ids, keep the first min(1 + 16 sqrt(120 N), N); the held-out set is the LAST
    round(min(5 % of the sample's tokens, 50 000)) tokens of the sample IN ITS SHUFFLED ORDER (whole passages from the end, the
    first one taken possibly cut to its tail), so it comes from random documents, not from the end of the corpus."""
    n = len(docs16)
    k = int(min(1.0 + 16.0 * math.sqrt(120.0 * n), n))
    sample = rng.permutation(n)[:k]
    total = int(sum(docs16[int(i)].shape[0] for i in sample))
    need = int(round(min(0.05 * total, 50_000.0)))
    parts = []
    for i in sample[::-1]:
        if need <= 0:
            break
        d = docs16[int(i)]
        if d.shape[0] <= need:
            parts.append(d)
            need -= d.shape[0]
        else:
            parts.append(d[d.shape[0] - need:])
            need = 0
    parts.reverse()
    dim = docs16[0].shape[1] if n else 0
    return np.concatenate(parts) if parts else np.zeros((0, dim), np.float16)


def build_index_arrays(docs, centroids, nbits: int, device: str = "cuda:0", heldout=None, num_partitions: int | None = None) -> dict:
    """The construct_index argument set for a list of [len, dim] document embeddings (the whole corpus is the
    held-out sample unless one is given), without the reference loader's trailing padding rows; plus `cluster_threshold`."""
    cent = _np(centroids, np.float16)
    docs16 = [_np(d, np.float16) for d in docs]
    lens = np.array([d.shape[0] for d in docs16], np.int64)
    allemb = np.concatenate(docs16) if docs16 else np.zeros((0, cent.shape[1]), np.float16)
    cut, wts, avg, thr = train_codec(allemb if heldout is None else heldout, cent, nbits, device, with_threshold=True)
    codes, packed = compress(cent, cutoffs_for_f32_compare(cut), allemb, nbits, device)
    P = num_partitions if num_partitions is not None else max(cent.shape[0], 1)
    ivf, ivf_lengths = synth.build_ivf(codes, lens, P)
    return dict(nbits=nbits, centroids=cent, avg_residual=avg.astype(np.float16), bucket_cutoffs=cut.astype(np.float16),
                bucket_weights=wts.astype(np.float16), ivf=ivf, ivf_lengths=ivf_lengths, doc_codes=codes, doc_residuals=packed,
                doc_lengths=lens, cluster_threshold=np.float32(thr),
                # what create.rs writes to the directory: the fp32 values themselves (the loader casts them to fp16)
                codec_f32=dict(avg_residual=np.asarray(avg, np.float32), bucket_cutoffs=np.asarray(cut, np.float32),
                               bucket_weights=np.asarray(wts, np.float32)))


def create_index(index_path: str, documents_embeddings, centroids, nbits: int = 4, device: str = "cuda:0", seed: int | None = 42,
                 compress_only: bool = False, chunk_docs: int = 25_000, heldout=None) -> dict:
    """rust/index/create.rs:206-583 given centroids: sample held-out passages (heldout_sample), train the codec (+ the cluster
    threshold), compress every document, build the IVF, write the directory.  Returns the array set it wrote.  `heldout` (tests)
    replaces the sampled held-out embeddings."""
    from .search import index_io
    docs16 = [_np(d, np.float16) for d in documents_embeddings]
    n = len(docs16)
    if n == 0:
        raise ValueError("Cannot create an index from zero documents")
    held = heldout_sample(docs16, np.random.default_rng(seed)) if heldout is None else _np(heldout, np.float16)
    if held.shape[0] == 0:
        raise ValueError("Cannot train codec: no heldout samples were generated.")   # create.rs:301-305
    # create.rs:239-243, :292-294: the IVF's list count, from n_docs * (sum / n_docs) in doubles like the reference (a total that sits
    # on a power of four must round the way it does there)
    n_tok = float(n) * (float(sum(d.shape[0] for d in docs16)) / float(n))
    num_partitions = int(2 ** math.floor(math.log2(16.0 * math.sqrt(n_tok))))
    arr = build_index_arrays(docs16, centroids, nbits, device, heldout=held, num_partitions=num_partitions)
    # what metadata.json carries (create.rs:572) -- the estimate itself, also when the centroids given number more (bincount's
    # minlength, :542: the lists then reach the highest code) and in compress_only mode; update.rs:66 reads it back as the list count
    arr["num_partitions"] = num_partitions
    if compress_only:
        arr["ivf"] = None
        arr["ivf_lengths"] = None
    index_io.save_index_arrays(index_path, arr, chunk_docs=chunk_docs, write_plan=True)
    return arr
