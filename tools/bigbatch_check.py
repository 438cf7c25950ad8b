import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, fast_plaid_amd as fp
R = fp.fast_plaid_rust
spec = fp.synth.SynthSpec(n_docs=1_000_000, doc_len=128, n_centroids=131072, seed=42)
cent = fp.synth.centroids(spec)
dev = R.construct_synthetic_index(spec, "cuda:0", centroids=cent)
q = fp.synth.make_queries(spec, cent, 4096, 32)
params = R.SearchParameters(2000, 4096, 100, 8)
t = time.perf_counter(); p, s, c = R.search_arrays(dev, q, params); dt = time.perf_counter() - t
print("B=4096: %.1f ms, %.0f qps, sub_batches=%d" % (dt * 1e3, 4096 / dt, R.last_search_counts()["sub_batches"]))
assert np.all(c == 100) and np.all(np.diff(s, axis=1) <= 0)
for b in (0, 1777, 3071, 3072, 4095):
    tr = R.search_trace(dev, q[b], params)
    assert np.array_equal(p[b], tr["pids"]) and np.array_equal(s[b], tr["scores"]), b
print("BIG_OK")
