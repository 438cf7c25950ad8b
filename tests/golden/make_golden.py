"""Generates tests/golden/*.npz (run once in the build container; fixtures are committed).

Producer: oracle/plaid_oracle_torch.py -- the op-for-op ATen (torch 2.10.0 CPU) restatement
of /root/reference/rust/search/search.rs.  The reference itself cannot be built/imported
in this image (Rust + PyO3, no cargo) and holds no golden vectors of its own, so these
fixtures pin the plain-C oracle and the HIP path to the reference's arithmetic dependency
(ATen), not to reference-run outputs.

Each case is checked for ambiguity before it is written: exact ties at the probe cut,
at the candidate-pruning cut or between final scores would make "the expected ids"
implementation-defined (ATen topk/sort are not stable), so such seeds are rejected.

    python tests/golden/make_golden.py [case names ...]
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import plaid_oracle_torch as OT  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = [
    # name, n_docs, (len_lo, len_hi), dim, nbits, C, B, Q, n_probe, n_full, top_k, extra
    dict(name="base_d128_nb4", n_docs=120, lens=(20, 120), dim=128, nbits=4, C=256, B=4, Q=32,
         n_probe=8, n_full=4096, top_k=10),
    dict(name="prune_d128_nb4", n_docs=160, lens=(10, 90), dim=128, nbits=4, C=256, B=4, Q=32,
         n_probe=8, n_full=64, top_k=10),
    dict(name="d64_nb2", n_docs=100, lens=(10, 60), dim=64, nbits=2, C=128, B=3, Q=20,
         n_probe=4, n_full=4096, top_k=5),
    dict(name="probe1", n_docs=120, lens=(20, 100), dim=128, nbits=4, C=256, B=3, Q=16,
         n_probe=1, n_full=32, top_k=8),
    dict(name="probe16_q50", n_docs=100, lens=(30, 110), dim=128, nbits=4, C=256, B=3, Q=50,
         n_probe=16, n_full=128, top_k=10),
    dict(name="topk_gt_ndocs", n_docs=12, lens=(5, 40), dim=128, nbits=4, C=64, B=2, Q=8,
         n_probe=8, n_full=4096, top_k=50),
    dict(name="subset", n_docs=150, lens=(10, 80), dim=128, nbits=4, C=256, B=4, Q=24,
         n_probe=8, n_full=4096, top_k=10, subset=True),
    dict(name="zero_pad_query", n_docs=100, lens=(10, 80), dim=128, nbits=4, C=128, B=2, Q=16,
         n_probe=4, n_full=4096, top_k=10, zero_pad=4),
    dict(name="empty_doc", n_docs=80, lens=(10, 60), dim=128, nbits=4, C=128, B=2, Q=16,
         n_probe=8, n_full=4096, top_k=80, empty_docs=(3, 41)),
    dict(name="unnormalised_docs", n_docs=60, lens=(20, 80), dim=128, nbits=4, C=128, B=2, Q=30,
         n_probe=8, n_full=4096, top_k=10, raw_randn=True),
    # shapes beyond dim 64/128 x nbits 2/4 (residual_codec.rs:83-140 and search.rs:53-107 are generic in both; the reference's
    # own benchmark model, answerai-colbert-small-v1, is dim 96: docs/benchmark/benchmark.py:44-45)
    dict(name="d96_nb4", n_docs=110, lens=(10, 90), dim=96, nbits=4, C=128, B=3, Q=32,
         n_probe=8, n_full=4096, top_k=10),
    dict(name="d48_nb2", n_docs=100, lens=(10, 70), dim=48, nbits=2, C=128, B=3, Q=24,
         n_probe=4, n_full=64, top_k=8),
    dict(name="d128_nb8", n_docs=80, lens=(10, 60), dim=128, nbits=8, C=128, B=2, Q=32,
         n_probe=8, n_full=4096, top_k=10),
    dict(name="d128_nb1", n_docs=80, lens=(10, 60), dim=128, nbits=1, C=128, B=2, Q=20,
         n_probe=4, n_full=4096, top_k=10),
    dict(name="d256_nb4", n_docs=60, lens=(10, 50), dim=256, nbits=4, C=64, B=2, Q=16,
         n_probe=4, n_full=4096, top_k=10),
    dict(name="d40_nb4", n_docs=80, lens=(10, 60), dim=40, nbits=4, C=64, B=2, Q=12,
         n_probe=4, n_full=32, top_k=6),
]


def ambiguous(tr, n_probe, top_k, allowed=None) -> str | None:
    S = tr.centroid_scores.float()
    if allowed is not None:
        S = S.index_select(0, allowed)
        n_probe = min(n_probe, S.shape[0])
    k = min(n_probe + 1, S.shape[0])
    if k > n_probe >= 1:
        top = S.topk(k, dim=0, largest=True, sorted=True).values
        nz = top[n_probe - 1] != 0  # all-zero (padded) columns tie by construction
        if bool(((top[n_probe - 1] == top[n_probe]) & nz).any()):
            return "probe tie"
    if tr.approx_scores is not None and tr.rerank_pids is not None:
        a = tr.approx_scores.sort(descending=True).values
        nr = tr.rerank_pids.shape[0]
        if nr < a.shape[0] and float(a[nr - 1]) == float(a[nr]):
            return "prune tie"
    if tr.exact_scores is not None:
        e = tr.exact_scores.sort(descending=True).values
        n = min(top_k + 1, e.shape[0])
        if n > 1 and bool((e[: n - 1] == e[1:n]).any()):
            return "final tie"
    return None


def make_case(cfg, seed):
    g = torch.Generator().manual_seed(seed)
    D = cfg["dim"]
    lo, hi = cfg["lens"]
    lens = torch.randint(lo, hi + 1, (cfg["n_docs"],), generator=g).tolist()
    for i in cfg.get("empty_docs", ()):
        lens[i] = 0
    docs = [torch.randn(n, D, generator=g) for n in lens]
    if not cfg.get("raw_randn"):
        docs = [torch.nn.functional.normalize(d, dim=-1) for d in docs]
    allt = torch.cat(docs)
    perm = torch.randperm(allt.shape[0], generator=g)[: cfg["C"]]
    cent = torch.nn.functional.normalize(allt[perm], dim=-1).half()
    arr = OT.build_index_arrays(docs, cent, cfg["nbits"])
    idx = OT.construct_index(**arr)
    q = torch.nn.functional.normalize(torch.randn(cfg["B"], cfg["Q"], D, generator=g), dim=-1)
    if cfg.get("zero_pad"):
        q[1, -cfg["zero_pad"]:, :] = 0.0  # what pad_sequence(padding_value=0.0) produces, fast_plaid.py:772-780
    q = q.half()
    subsets = None
    if cfg.get("subset"):
        subsets = []
        for b in range(cfg["B"]):
            n = int(torch.randint(5, 60, (1,), generator=g))
            s = torch.randint(0, cfg["n_docs"], (n,), generator=g).tolist()  # duplicates allowed
            subsets.append(s)
    out = {k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arr.items()}
    out["queries"] = q.numpy()
    out["params"] = np.array([cfg["n_probe"], cfg["n_full"], cfg["top_k"], 2000], dtype=np.int64)
    for b in range(cfg["B"]):
        sub = None if subsets is None else torch.tensor(subsets[b], dtype=torch.int64)
        tr = OT.search(q[b], idx, cfg["n_probe"], 2000, cfg["n_full"], cfg["top_k"], sub, return_token_scores=(b < 2))
        allowed = None
        if sub is not None:
            allowed = torch.unique(idx.doc_codes.lookup(sub)[0])
        why = ambiguous(tr, cfg["n_probe"], cfg["top_k"], allowed)
        if why:
            return None, why
        if b < 2:
            out[f"S_{b}"] = tr.centroid_scores.numpy()
        out[f"cells_{b}"] = tr.cells.numpy()
        zero_cols = (q[b].float().abs().sum(-1) == 0)
        if bool(zero_cols.any()):
            # all-zero query tokens (python-side padding) probe n_ivf_probe ARBITRARY centroids
            # (ATen topk over an all-equal column); only the cells of the real tokens are defined.
            S_nz = tr.centroid_scores[:, ~zero_cols]
            sel = S_nz.topk(cfg["n_probe"], dim=0, largest=True, sorted=False).indices
            out[f"cells_nz_{b}"] = torch.unique(sel.flatten()).numpy()
        out[f"cand_{b}"] = tr.candidates.numpy()
        out[f"approx_{b}"] = (tr.approx_scores.numpy() if tr.approx_scores is not None else np.zeros(0, np.float32))
        if tr.rerank_pids is not None:
            order = torch.argsort(tr.rerank_pids)
            out[f"rerank_{b}"] = tr.rerank_pids[order].numpy()
            out[f"exact_{b}"] = (tr.exact_scores[order].numpy() if tr.exact_scores is not None else np.zeros(0, np.float32))
        else:
            out[f"rerank_{b}"] = np.zeros(0, np.int64)
            out[f"exact_{b}"] = np.zeros(0, np.float32)
        out[f"pids_{b}"] = np.asarray(tr.pids, dtype=np.int64)
        out[f"scores_{b}"] = np.asarray(tr.scores, dtype=np.float32)
        if tr.token_matrices is not None:   # search.rs:668-686: [query_tokens, doc_tokens] fp16 per hit (first 3 hits kept)
            for i, m in enumerate(tr.token_matrices[:3]):
                out[f"tokmat_{b}_{i}"] = m.contiguous().numpy()
        if subsets is not None:
            out[f"subset_{b}"] = np.asarray(subsets[b], dtype=np.int64)
    # a decompression sample straight from the ATen op sequence (search.rs:53-107)
    n_s = min(64, arr["doc_codes"].shape[0])
    out["decomp_sample"] = OT.decompress_residuals(
        arr["doc_residuals"][:n_s], idx.bucket_weights, idx.rev_map, idx.idx_lookup,
        arr["doc_codes"][:n_s], idx.centroids, D, cfg["nbits"]).numpy()
    return out, None


def main():
    torch.set_num_threads(1)
    only = set(sys.argv[1:])   # optional case names: regenerate only those
    for cfg in CASES:
        if only and cfg["name"] not in only:
            continue
        seed = 1000
        while True:
            out, why = make_case(cfg, seed)
            if out is not None:
                break
            print(f"  {cfg['name']}: seed {seed} rejected ({why})")
            seed += 1
        out["seed"] = np.array([seed])
        path = os.path.join(OUT, cfg["name"] + ".npz")
        np.savez_compressed(path, **out)
        print(f"{cfg['name']}: seed {seed}, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
