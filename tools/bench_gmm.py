#!/usr/bin/env python
"""A corpus built FROM VECTORS (not from the compressed-domain generator of fast-plaid_amd/synth.py): does S4's level 0 prune as
well when codes and residuals come out of k-means + the residual codec?

  embeddings   Gaussian mixture on the unit sphere: n_centers mixture centres; a document draws 6 of them (Zipf over a scrambled
               ranking) and every token is normalize(centre + sigma * noise) of one of its centres (p = 0.85) or of a uniformly drawn
               centre; fp16
  index        kmeans.compute_kmeans (the reference's sampling protocol, Lloyd iterations with the assignment on the device) ->
               create.train_codec (bucket cutoffs / weights from held-out residuals) -> fp_compress -> IVF -> fp_index_create
  queries      32 tokens: tokens of random documents + noise, re-normalised

Reports per batch: candidates, documents rescored exactly, stage times -- for the form of S4 the engine picks and for the forced
ones (FP_APPROX_IMPL is read once per process: run this script once per form; the built arrays are cached in --cache).
With --parity: the first --parity documents as their own index against the C oracle (identical id lists modulo the usual near-ties).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_docs(n_docs, doc_len, dim, n_centers, sigma, seed):
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    centers = torch.nn.functional.normalize(torch.randn(n_centers, dim, device="cuda", generator=g), dim=1)
    # Zipf over a scrambled ranking, like synth.py's topics: octave e uniform in [0, lg), rank uniform inside the octave
    lg = int(np.log2(n_centers))
    out = np.empty((n_docs * doc_len, dim), np.float16)
    chunk = 20000
    for d0 in range(0, n_docs, chunk):
        nd = min(chunk, n_docs - d0)
        e = torch.randint(0, lg, (nd, 6), device="cuda", generator=g)
        rank = (2 ** e - 1) + (torch.rand(nd, 6, device="cuda", generator=g) * (2 ** e)).long()
        topics = (rank * 2654435761 + 12345) % n_centers                                 # [nd, 6]
        pick = torch.randint(0, 6, (nd, doc_len), device="cuda", generator=g)
        tok_c = torch.gather(topics, 1, pick)
        uni = torch.randint(0, n_centers, (nd, doc_len), device="cuda", generator=g)
        use_topic = torch.rand(nd, doc_len, device="cuda", generator=g) < 0.85
        cid = torch.where(use_topic, tok_c, uni).reshape(-1)
        x = centers[cid] + sigma * torch.randn(nd * doc_len, dim, device="cuda", generator=g) / np.sqrt(dim)
        x = torch.nn.functional.normalize(x, dim=1).half()
        out[d0 * doc_len:(d0 + nd) * doc_len] = x.cpu().numpy()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=250_000)
    ap.add_argument("--doc-len", type=int, default=128)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--centers", type=int, default=1 << 15)
    ap.add_argument("--sigma", type=float, default=0.6)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--cache", type=str, default="/tmp/gmm_corpus.npz")
    ap.add_argument("--parity", type=int, default=0, help="documents of the slice compared with the C oracle (0 = skip)")
    ap.add_argument("--tag", type=str, default="")
    a = ap.parse_args()
    import torch  # (first: torch wheels bundle their own HIP runtime)
    import fast_plaid_amd as fp
    from fast_plaid_amd import create, kmeans
    R = fp.fast_plaid_rust
    t0 = time.time()
    if os.path.exists(a.cache):
        z = np.load(a.cache)
        arr = {k: z[k] for k in z.files}
        emb_q = arr.pop("query_pool")
    else:
        emb = make_docs(a.docs, a.doc_len, a.dim, a.centers, a.sigma, 7)
        lens = np.full(a.docs, a.doc_len, np.int64)
        rng = np.random.default_rng(11)
        # k-means on a document sample (the reference's protocol), fewer points per centroid than its 256 to keep the host-side
        # Lloyd reductions short
        n_tok = a.docs * a.doc_len
        K = int(2 ** np.floor(np.log2(16 * np.sqrt(n_tok))))
        n_sample_docs = min(a.docs, max(1, (K * 48) // a.doc_len))
        picked = rng.permutation(a.docs)[:n_sample_docs]
        sample = np.concatenate([emb[i * a.doc_len:(i + 1) * a.doc_len] for i in picked])
        cent = kmeans.lloyd(sample, K, 4, rng, "cuda:0", max_points_per_centroid=48)
        cent = (cent / np.maximum(np.linalg.norm(cent, axis=1, keepdims=True), 1e-12)).astype(np.float16)
        held = emb[rng.permutation(emb.shape[0])[:50000]]
        cut, wts, avg = create.train_codec(held, cent, 4, "cuda:0")
        codes, packed = create.compress(cent, create.cutoffs_for_f32_compare(cut), emb, 4, "cuda:0")
        ivf, ivf_lengths = fp.synth.build_ivf(codes, lens, K)
        emb_q = emb[rng.permutation(emb.shape[0])[:200000]].copy()
        arr = dict(nbits=np.int64(4), centroids=cent, bucket_weights=wts.astype(np.float16), ivf=ivf, ivf_lengths=ivf_lengths, doc_codes=codes,
                   doc_residuals=packed, doc_lengths=lens)
        np.savez(a.cache, query_pool=emb_q, **arr)
    t_build = time.time() - t0
    nbits = int(arr["nbits"])
    idx = R.construct_index(nbits, arr["centroids"], None, None, arr["bucket_weights"], arr["ivf"], arr["ivf_lengths"], arr["doc_codes"],
                            arr["doc_residuals"], arr["doc_lengths"], "cuda:0", False)
    rng = np.random.default_rng(5)

    def queries(n, seed):
        r = np.random.default_rng(seed)
        q = emb_q[r.integers(0, emb_q.shape[0], n * 32)].astype(np.float32) + 0.3 * r.standard_normal((n * 32, a.dim)).astype(np.float32) / np.sqrt(a.dim)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        return q.reshape(n, 32, a.dim).astype(np.float16)

    params = R.SearchParameters(2000, 4096, 1000, 8)
    R.set_graph_replay(False)
    acc, cand, exact = {}, 0, 0
    for i in range(a.steps + 3):
        R.search_arrays(idx, queries(a.batch, 100 + i), params)
        if i >= 3:
            for k, v in R.last_search_timings().items():
                acc[k] = acc.get(k, 0.0) + v
            c = R.last_search_counts()
            cand += c["candidates"]
            exact += c["approx_exact"]
    out = {"tag": a.tag or os.environ.get("FP_APPROX_IMPL", "auto"), "docs": int(arr["doc_lengths"].shape[0]), "centroids": int(arr["centroids"].shape[0]),
           "unique_codes_per_doc": idx.n_unique_codes / max(idx.n_docs, 1), "build_s": round(t_build, 1),
           "candidates_per_query": cand / a.steps / a.batch, "rescored_exactly_per_query": exact / a.steps / a.batch,
           "survivor_fraction": exact / max(cand, 1), "stages_ms": {k: round(v / a.steps, 4) for k, v in acc.items()},
           "ms_per_batch": round(sum(acc.values()) / a.steps, 4), "hard_tokens": idx.n_hard_tokens}
    if a.parity:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import plaid_oracle as OC
        from fast_plaid_amd import sharded
        sl = sharded.shard_arrays(dict(arr, nbits=nbits), 0, a.parity)
        hip = R.construct_index(nbits, sl["centroids"], None, None, sl["bucket_weights"], sl["ivf"], sl["ivf_lengths"], sl["doc_codes"],
                                sl["doc_residuals"], sl["doc_lengths"], "cuda:0", False)
        orc = OC.OracleIndex(nbits=nbits, centroids=sl["centroids"], bucket_weights=sl["bucket_weights"], ivf=sl["ivf"], ivf_lengths=sl["ivf_lengths"],
                             doc_codes=sl["doc_codes"], doc_residuals=sl["doc_residuals"], doc_lengths=sl["doc_lengths"])
        q = queries(16, 999)
        p2 = R.SearchParameters(2000, 2048, 200, 8)
        gp, gs, gc = R.search_arrays(hip, q, p2)
        ref = orc.search(q, 200, 2048, 8, nthreads=16)
        ident = sum(int(np.array_equal(gp[b, :gc[b]], ref[b][0])) for b in range(16))
        md = max(abs(dict(zip(ref[b][0].tolist(), ref[b][1].tolist())).get(p, s) - s) for b in range(16) for p, s in zip(gp[b, :gc[b]].tolist(), gs[b, :gc[b]].tolist()))
        out["parity_slice"] = {"docs": a.parity, "queries": 16, "identical_id_lists": ident, "max_abs_score_diff": float(md)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
