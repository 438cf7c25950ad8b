#!/usr/bin/env python
"""round 6: the other entry points / corpus shapes at cfg2 size: token-score matrices of the hits, reconstruct_embeddings,
variable-length documents, nbits 2.  ms per call."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fast_plaid_amd as fp  # noqa: E402

R = fp.fast_plaid_rust


def timeit(f, n=8, warm=3):
    for _ in range(warm):
        f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    for label, kw in (("cfg2", {}), ("cfg2, variable-length documents (32..128 tokens)", {"variable_len": True}), ("cfg2, nbits 2", {"nbits": 2})):
        spec = fp.synth.SynthSpec(n_docs=1_000_000, doc_len=128, n_centroids=131072, dim=128, seed=42, **({"nbits": 4} | kw))
        cent = fp.synth.centroids(spec)
        ix = R.construct_synthetic_index(spec, "cuda:0", centroids=cent, bucket_weights=fp.synth.bucket_weights(spec))
        qs = [fp.synth.make_queries(spec, cent, 64, 32, seed=70 + i) for i in range(12)]
        it = iter(range(10 ** 9))
        p10, p1000 = R.SearchParameters(2000, 4096, 10, 8), R.SearchParameters(2000, 4096, 1000, 8)
        out = {"corpus": label}
        out["search top_k 1000 ms"] = round(timeit(lambda: R.search_arrays(ix, qs[next(it) % 12], p1000)), 3)
        out["search top_k 10 ms"] = round(timeit(lambda: R.search_arrays(ix, qs[next(it) % 12], p10)), 3)
        if not kw:
            out["pysearch_with_token_scores top_k 10 ms"] = round(timeit(lambda: R.pysearch_with_token_scores(ix, "cuda:0", qs[next(it) % 12], p10)), 3)
            out["pysearch (list results) top_k 1000 ms"] = round(timeit(lambda: R.pysearch(ix, "cuda:0", qs[next(it) % 12], p1000)), 3)
            ids = np.random.default_rng(1).integers(0, spec.n_docs, 1000).tolist()
            out["reconstruct_embeddings of 1000 documents ms"] = round(timeit(lambda: R.reconstruct_embeddings(ix, ids), n=4, warm=1), 3)
        print(json.dumps(out), flush=True)
        del ix


if __name__ == "__main__":
    main()
