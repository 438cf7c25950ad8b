"""Synthetic PLAID corpora in the compressed domain.

The benchmark corpora (1M..10M documents) are far too large to k-means / compress in a
benchmark run, so documents are generated directly as what ``create.rs`` would have
written: per-token centroid codes + packed residual bytes + doc lengths, with the IVF
derived from the codes.  Everything integer is a pure function of (seed, doc, token)
through a splitmix64 counter hash, so that

* the HIP generator in ``csrc/fp_synth.hip`` (used for the large benchmark corpora,
  which never exist on the host) and this numpy twin produce bit-identical arrays, and
* any sampled document of a 10M-document device-resident corpus can be re-materialised
  on the host for the size-independent parity checks.

Floating-point inputs (centroids, bucket weights, queries) are always generated here on
the host and uploaded; they are small.

Recipe: centroids = normalize(N(0, I)); each document draws a topic of 8 centroids from a
piecewise-uniform Zipf(1) over the (scrambled) centroid ids; each token takes a topic
centroid w.p. 205/256 (~0.8) else a uniform centroid; residual bytes are uniform;
bucket weights are the mid-bucket quantiles of N(0, 0.05^2).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from statistics import NormalDist

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)

# stream ids (must match csrc/fp_synth.hip)
S_DOCLEN, S_TOPIC, S_TOKEN, S_RESID = 1, 2, 3, 4
TOPIC_SIZE = 8
P_TOPIC_256 = 205  # P(token takes a topic centroid) = 205/256


def mix64(x):
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = (np.asarray(x, dtype=np.uint64) + _GOLD) & _M64
        z = ((z ^ (z >> np.uint64(30))) * _C1) & _M64
        z = ((z ^ (z >> np.uint64(27))) * _C2) & _M64
        return z ^ (z >> np.uint64(31))


def stream_key(seed: int, stream: int) -> np.uint64:
    with np.errstate(over="ignore"):
        return mix64(np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.uint64(stream))


def rnd(seed: int, stream: int, counter):
    with np.errstate(over="ignore"):
        return mix64(stream_key(seed, stream) + np.asarray(counter, dtype=np.uint64))


@dataclass
class SynthSpec:
    n_docs: int
    doc_len: int
    n_centroids: int  # power of two
    dim: int = 128
    nbits: int = 4
    variable_len: bool = False  # lengths ~ U[doc_len/4, doc_len]
    seed: int = 42

    @property
    def lg_c(self) -> int:
        lg = int(math.log2(self.n_centroids))
        assert 1 << lg == self.n_centroids, "n_centroids must be a power of two"
        return lg

    @property
    def packed_dim(self) -> int:
        return self.dim * self.nbits // 8


def default_num_centroids(n_tokens: float) -> int:
    """fast_plaid.py:150-154: 2^floor(log2(16*sqrt(tokens)))."""
    return int(2 ** math.floor(math.log2(16 * math.sqrt(n_tokens))))


def doc_lengths(spec: SynthSpec, pids=None) -> np.ndarray:
    pids = np.arange(spec.n_docs, dtype=np.uint64) if pids is None else np.asarray(pids, dtype=np.uint64)
    if not spec.variable_len:
        return np.full(pids.shape[0], spec.doc_len, dtype=np.int64)
    lo = max(spec.doc_len // 4, 1)
    span = np.uint64(spec.doc_len - lo + 1)
    return (lo + (rnd(spec.seed, S_DOCLEN, pids) % span).astype(np.int64)).astype(np.int64)


def _zipf_centroid(spec: SynthSpec, r):
    """Piecewise-uniform Zipf(1): octave e uniform in [0, lgC), rank uniform inside the
    octave [2^e - 1, 2^(e+1) - 1); rank scrambled to a centroid id by an odd multiplier."""
    lg = np.uint64(spec.lg_c)
    e = r % lg
    with np.errstate(over="ignore"):
        rank = ((np.uint64(1) << e) - np.uint64(1)) + ((r >> np.uint64(8)) & ((np.uint64(1) << e) - np.uint64(1)))
        cid = (rank * np.uint64(0x9E3779B1) + np.uint64(12345)) & np.uint64(spec.n_centroids - 1)
    return cid


def doc_topics(spec: SynthSpec, pids) -> np.ndarray:
    pids = np.asarray(pids, dtype=np.uint64)
    ctr = pids[:, None] * np.uint64(TOPIC_SIZE) + np.arange(TOPIC_SIZE, dtype=np.uint64)[None, :]
    return _zipf_centroid(spec, rnd(spec.seed, S_TOPIC, ctr)).astype(np.int64)  # [n, 8]


def token_codes(spec: SynthSpec, pids, tok_global) -> np.ndarray:
    """codes of tokens with global token indices `tok_global`, owned by docs `pids`
    (same length arrays)."""
    r = rnd(spec.seed, S_TOKEN, tok_global)
    topics = doc_topics(spec, pids)  # [n, 8]
    use_topic = (r & np.uint64(0xFF)) < np.uint64(P_TOPIC_256)
    slot = ((r >> np.uint64(8)) & np.uint64(TOPIC_SIZE - 1)).astype(np.int64)
    from_topic = topics[np.arange(topics.shape[0]), slot]
    uniform = ((r >> np.uint64(16)) & np.uint64(spec.n_centroids - 1)).astype(np.int64)
    return np.where(use_topic, from_topic, uniform).astype(np.int64)


def token_residuals(spec: SynthSpec, tok_global) -> np.ndarray:
    """[n, packed_dim] uint8; 8 bytes per 64-bit draw, little-endian."""
    tok = np.asarray(tok_global, dtype=np.uint64)
    words = (spec.packed_dim + 7) // 8
    ctr = tok[:, None] * np.uint64(words) + np.arange(words, dtype=np.uint64)[None, :]
    r = np.ascontiguousarray(rnd(spec.seed, S_RESID, ctr))  # [n, words] uint64
    return r.view(np.uint8).reshape(tok.shape[0], words * 8)[:, : spec.packed_dim].copy()


def centroids(spec: SynthSpec) -> np.ndarray:
    rng = np.random.default_rng(spec.seed)
    c = rng.standard_normal((spec.n_centroids, spec.dim), dtype=np.float32)
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    return c.astype(np.float16)


def bucket_weights(spec: SynthSpec, sigma: float = 0.05) -> np.ndarray:
    n = 1 << spec.nbits
    nd = NormalDist(0.0, sigma)
    return np.array([nd.inv_cdf((i + 0.5) / n) for i in range(n)], dtype=np.float32).astype(np.float16)


def bucket_cutoffs(spec: SynthSpec, sigma: float = 0.05) -> np.ndarray:
    n = 1 << spec.nbits
    nd = NormalDist(0.0, sigma)
    return np.array([nd.inv_cdf(i / n) for i in range(1, n)], dtype=np.float32).astype(np.float16)


def make_queries(spec: SynthSpec, cent: np.ndarray, n_queries: int, q_len: int, seed: int = 7,
                 noise: float = 0.3) -> np.ndarray:
    """each query token = normalize(centroid[c] + noise*N(0,I)/sqrt(D)) for a random token
    code c of a random document (so probes hit populated cells) -> [B, Q, D] f16."""
    rng = np.random.default_rng(seed)
    lens = doc_lengths(spec)
    offs = np.concatenate([[0], np.cumsum(lens)])
    out = np.zeros((n_queries, q_len, spec.dim), np.float32)
    for b in range(n_queries):
        d = int(rng.integers(0, spec.n_docs))
        toks = rng.integers(0, lens[d], size=q_len) + offs[d]
        codes = token_codes(spec, np.full(q_len, d), toks)
        v = cent[codes].astype(np.float32) + noise * rng.standard_normal((q_len, spec.dim), dtype=np.float32) / math.sqrt(spec.dim)
        out[b] = v / np.linalg.norm(v, axis=1, keepdims=True)
    return out.astype(np.float16)


def build_ivf(codes: np.ndarray, lens: np.ndarray, n_lists: int):
    """per-centroid ascending unique pids (create.rs:528-559 + optimize_ivf :55-132)."""
    pid_of_tok = np.repeat(np.arange(lens.shape[0], dtype=np.int64), lens)
    key = np.unique(codes.astype(np.int64) * np.int64(lens.shape[0]) + pid_of_tok)
    cell = key // np.int64(lens.shape[0])
    ivf = (key - cell * np.int64(lens.shape[0])).astype(np.int64)
    ivf_lengths = np.bincount(cell, minlength=n_lists).astype(np.int32)
    return ivf, ivf_lengths


def host_index_arrays(spec: SynthSpec) -> dict:
    """The full construct_index argument set on the host (small/medium corpora)."""
    lens = doc_lengths(spec)
    T = int(lens.sum())
    tok = np.arange(T, dtype=np.uint64)
    pid_of_tok = np.repeat(np.arange(spec.n_docs, dtype=np.int64), lens)
    codes = np.empty(T, np.int64)
    res = np.empty((T, spec.packed_dim), np.uint8)
    step = 1 << 20
    for s in range(0, T, step):
        e = min(T, s + step)
        codes[s:e] = token_codes(spec, pid_of_tok[s:e], tok[s:e])
        res[s:e] = token_residuals(spec, tok[s:e])
    ivf, ivf_lengths = build_ivf(codes, lens, spec.n_centroids)
    return dict(
        nbits=spec.nbits, centroids=centroids(spec), avg_residual=np.zeros(spec.dim, np.float16),
        bucket_cutoffs=bucket_cutoffs(spec), bucket_weights=bucket_weights(spec),
        ivf=ivf, ivf_lengths=ivf_lengths, doc_codes=codes, doc_residuals=res, doc_lengths=lens,
    )


def host_docs(spec: SynthSpec, pids) -> dict:
    """Re-materialise selected documents of a (possibly device-only) corpus as a compact
    host sub-corpus: codes/residuals/lengths of exactly those docs, in the given order."""
    pids = np.asarray(pids, dtype=np.int64)
    if spec.variable_len:
        all_lens = doc_lengths(spec)
        offs = np.concatenate([[0], np.cumsum(all_lens)])
        lens = all_lens[pids]
        starts = offs[pids]
    else:
        lens = np.full(pids.shape[0], spec.doc_len, np.int64)
        starts = pids * spec.doc_len
    pid_of_tok = np.repeat(pids, lens)
    within = np.arange(int(lens.sum()), dtype=np.int64) - np.repeat(np.concatenate([[0], np.cumsum(lens)[:-1]]), lens)
    tok = (np.repeat(starts, lens) + within).astype(np.uint64)
    return dict(doc_codes=token_codes(spec, pid_of_tok, tok), doc_residuals=token_residuals(spec, tok),
                doc_lengths=lens)
