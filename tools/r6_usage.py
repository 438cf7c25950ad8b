#!/usr/bin/env python
"""round 6: the cfg2 corpus under call shapes a user of FastPlaid.search makes that the headline configuration does not: top_k 10,
one query, q_len 50 (zero-padded to 64), subsets (one list for all queries / per query), zero-padded list inputs.
ms per call (mean of --steps calls after warm-up, distinct query batches), stage times of the last call."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fast_plaid_amd as fp  # noqa: E402

R = fp.fast_plaid_rust


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=12)
    a = ap.parse_args()
    spec = fp.synth.SynthSpec(n_docs=a.docs, doc_len=128, n_centroids=131072, dim=128, nbits=4, seed=42)
    cent = fp.synth.centroids(spec)
    ix = R.construct_synthetic_index(spec, "cuda:0", centroids=cent, bucket_weights=fp.synth.bucket_weights(spec))
    rng = np.random.default_rng(3)

    def run(name, B, Q, top_k, n_full=4096, subset=None, zero_rows=0, n_probe=8):
        p = R.SearchParameters(2000, n_full, top_k, n_probe)
        qs = [fp.synth.make_queries(spec, cent, B, Q, seed=50 + i) for i in range(a.steps + 4)]
        if zero_rows:
            for q in qs:
                q[:, Q - zero_rows:, :] = 0
        for i in range(4):
            R.search_arrays(ix, qs[i], p, subset)
        t0 = time.perf_counter()
        for i in range(a.steps):
            R.search_arrays(ix, qs[4 + i], p, subset)
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        R.set_graph_replay(False)
        R.search_arrays(ix, qs[4], p, subset)
        R.search_arrays(ix, qs[5], p, subset)
        st = {k: round(v, 3) for k, v in R.last_search_timings().items() if v >= 0.05}
        form = {"s1_lazy": R.last_s1_counts()["lazy"], "lazy_overflows": R.last_search_counts()["lazy_overflows"], "s4": R.last_search_counts()["s4_form"]}
        R.set_graph_replay(True)
        print(json.dumps({"case": name, "B": B, "q_len": Q, "top_k": top_k, "ms_per_call": round(ms, 3), "qps": round(B / ms * 1e3, 1), "form": form,
                          "stages_ms>=0.05": st}), flush=True)

    run("headline: B 64, q_len 32, top_k 1000", 64, 32, 1000)
    run("top_k 10", 64, 32, 10)
    run("top_k 10, n_full_scores 1024", 64, 32, 10, 1024)
    run("one query, top_k 10", 1, 32, 10)
    run("8 queries, top_k 10", 8, 32, 10)
    run("q_len 50", 64, 50, 10)
    run("q_len 32 with 12 zero-padded rows", 64, 32, 10, zero_rows=12)
    run("n_ivf_probe 16", 64, 32, 10, n_probe=16)
    run("n_ivf_probe 32", 64, 32, 10, n_probe=32)
    run("n_ivf_probe 1", 64, 32, 10, n_probe=1)
    run("256 queries", 256, 32, 10)
    run("n_full_scores 16384, top_k 100", 64, 32, 100, 16384)
    run("n_ivf_probe 64", 64, 32, 10, n_probe=64)
    run("top_k 10 again (after the n_ivf_probe 64 calls)", 64, 32, 10)
    sub_small = [rng.choice(a.docs, 10_000, replace=False).tolist()] * 64
    run("subset: one list of 10 k ids for all queries", 64, 32, 10, subset=sub_small)
    sub_big = [rng.choice(a.docs, 300_000, replace=False).tolist()] * 64
    run("subset: one list of 300 k ids for all queries", 64, 32, 10, subset=sub_big)
    sub_pq = [rng.choice(a.docs, 50_000, replace=False).tolist() for _ in range(64)]
    run("subset: 50 k ids per query", 64, 32, 10, subset=sub_pq)


if __name__ == "__main__":
    main()
