"""GPU worker for tests/test_zz_graph_replay.py: FP_GRAPH=1 (set by the test).  From the third batch of a shape on, fp_search
replays one captured HIP graph (query upload from a pinned staging buffer, every launch and fill of S1..S8, result download)
instead of enqueuing ~55 launches.  Every batch -- waited-for, speculative, captured, replayed, and after a switch to another
shape and back -- must equal fp_search_trace, which never speculates and never replays."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fptest_env import test_opt  # noqa: E402

import fast_plaid_amd as fp  # noqa: E402


def main():
    assert os.environ.get("FP_GRAPH") == "1"
    R = fp.fast_plaid_rust
    spec = fp.synth.SynthSpec(n_docs=20000, doc_len=32, n_centroids=1024, variable_len=True, seed=5)
    host = fp.synth.host_index_arrays(spec)
    dev = R.construct_synthetic_index(spec, "cuda:0", centroids=host["centroids"])
    shapes = [(6, 4, 512, 50), (1, 8, 1024, 100), (6, 4, 512, 50), (3, 16, 1024, 20)]   # (batch, n_ivf_probe, n_full_scores, top_k)
    runs = []
    for si, (nb, n_probe, n_full, top_k) in enumerate(shapes):
        params = R.SearchParameters(2000, n_full, top_k, n_probe)
        for rep in range(7):
            q = fp.synth.make_queries(spec, host["centroids"], nb, 32, seed=500 + 10 * si + (rep % 5))
            pids, scores, counts = R.search_arrays(dev, q, params)
            runs.append((params, q, pids.copy(), scores.copy(), counts.copy(), R.last_search_counts()))
    for params, q, pids, scores, counts, cnt in runs:
        assert cnt["candidates"] > 0
        for b in range(q.shape[0]):
            t = R.search_trace(dev, q[b], params)
            assert counts[b] == len(t["pids"]), (b, counts[b], len(t["pids"]))
            assert np.array_equal(pids[b, : counts[b]], t["pids"])
            assert np.array_equal(scores[b, : counts[b]], t["scores"])
    # a caller alternating between three batch sizes: the capacity and the graph are kept per shape (eight most recent shapes of
    # a scratch), so after each shape's two warm-up forms (waited-for, speculative) every call is one graph launch: 24 of these 30
    before = R.graph_replay_count()
    alt = []
    for i in range(30):
        nb = (6, 3, 1)[i % 3]
        params = R.SearchParameters(2000, 512, 50, 4)
        q = fp.synth.make_queries(spec, host["centroids"], nb, 32, seed=900 + i % 3)   # (the same batch per shape: a larger candidate total would raise the learnt capacity, a new key)
        pids, scores, counts = R.search_arrays(dev, q, params)
        alt.append((params, q, pids.copy(), scores.copy(), counts.copy()))
    replays = R.graph_replay_count() - before
    if not test_opt("spec_cap_pct"):   # (the forced-overflow run never keeps a capacity: nothing to replay)
        assert replays >= 24, replays   # 30 calls - 3 shapes x 2 warm-up forms
    for params, q, pids, scores, counts in alt:
        for b in range(q.shape[0]):
            t = R.search_trace(dev, q[b], params)
            assert counts[b] == len(t["pids"]) and np.array_equal(pids[b, : counts[b]], t["pids"]) and np.array_equal(scores[b, : counts[b]], t["scores"])
    # an overflow on a REPLAYED batch: the capacity of a new shape is learnt from batches of all-zero queries (every token probes the
    # same few lowest-numbered cells: a small candidate total), then a batch of real queries -- many times the candidates -- arrives
    # through the captured graph.  The device empties such a batch; the host must see the true total (it travels in the result
    # block: the graph has no copy node of its own for it) and run the batch again.
    params = R.SearchParameters(2000, 512, 50, 4)
    qz = np.zeros((4, 32, host["centroids"].shape[1]), np.float16)
    for _ in range(5):
        R.search_arrays(dev, qz, params)
    small = R.last_search_counts()["candidates"]
    before = R.graph_replay_count()
    R.search_arrays(dev, qz, params)
    replayed_small = R.graph_replay_count() - before
    q = fp.synth.make_queries(spec, host["centroids"], 4, 32, seed=77)
    pids, scores, counts = R.search_arrays(dev, q, params)
    big = R.last_search_counts()["candidates"]
    for b in range(4):
        t = R.search_trace(dev, q[b], params)
        assert counts[b] == len(t["pids"]) and len(t["pids"]) > 0, (b, counts[b], len(t["pids"]))
        assert np.array_equal(pids[b, : counts[b]], t["pids"]) and np.array_equal(scores[b, : counts[b]], t["scores"])
    if not test_opt("spec_cap_pct"):
        assert replayed_small == 1 and big > 2 * small, (replayed_small, small, big)
    # a DISTINCT query batch per call (what a serving loop and the bench do): the candidate totals scatter by a few per cent, and
    # the learnt capacity -- part of the graph's key -- must not follow every new maximum of them (until round 6 it did: each new
    # maximum cost two calls outside the replay).  16 calls of a fresh shape: waited-for, speculative, capturing, then replays --
    # a total that comes within 8 % of the capacity may still move it once.
    params = R.SearchParameters(2000, 512, 40, 8)
    before = R.graph_replay_count()
    for i in range(16):
        q = fp.synth.make_queries(spec, host["centroids"], 5, 32, seed=3000 + i)
        pids, scores, counts = R.search_arrays(dev, q, params)
        if i in (0, 7, 15):
            for b in range(5):
                t = R.search_trace(dev, q[b], params)
                assert counts[b] == len(t["pids"]) and np.array_equal(pids[b, : counts[b]], t["pids"]) and np.array_equal(scores[b, : counts[b]], t["scores"])
    if not test_opt("spec_cap_pct"):
        assert R.graph_replay_count() - before >= 11, R.graph_replay_count() - before
    # a shape the threshold probe does not serve (n_ivf_probe > 32 raises the probe flag on purpose, to route its select kernel)
    # through the replayed graph: one launch per call -- until round 6 the replay took the flag for a tie overflow, ran every such
    # batch twice and left the scratch on the probe fallback and the eager S1 for every later shape
    params40 = R.SearchParameters(2000, 512, 50, 40)
    q40 = fp.synth.make_queries(spec, host["centroids"], 4, 32, seed=1234)
    for _ in range(6):
        pids, scores, counts = R.search_arrays(dev, q40, params40)
    for b in range(4):
        t = R.search_trace(dev, q40[b], params40)
        assert counts[b] == len(t["pids"]) and np.array_equal(pids[b, : counts[b]], t["pids"]) and np.array_equal(scores[b, : counts[b]], t["scores"])
    before = R.graph_replay_count()
    R.search_arrays(dev, q40, params40)
    if not test_opt("spec_cap_pct"):
        assert R.graph_replay_count() - before == 1
    R.set_graph_replay(False)
    R.search_arrays(dev, fp.synth.make_queries(spec, host["centroids"], 4, 32, seed=1235), R.SearchParameters(2000, 512, 50, 4))
    assert R.last_s1_counts()["lazy"] == 1, "a later shape of the scratch lost the lazy S1"
    R.set_graph_replay(True)
    print("GRAPH_OK replays", replays, "overflow on a replayed batch:", small, "->", big)


if __name__ == "__main__":
    main()
