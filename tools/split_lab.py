"""How much does running a batch as K concurrent sub-batches (own scratch / stream / graph each, one host thread each) buy?
Usage (GPU): python tools/split_lab.py [K ...]"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fast_plaid_amd as fp
R = fp.fast_plaid_rust
C = fp.synth.default_num_centroids(1_000_000 * 128)
spec = fp.synth.SynthSpec(n_docs=1_000_000, doc_len=128, n_centroids=C, seed=42, dim=128)
cent = fp.synth.centroids(spec)
index = R.construct_synthetic_index(spec, "cuda:0", centroids=cent, bucket_weights=fp.synth.bucket_weights(spec))
params = R.SearchParameters(2000, 4096, 1000, 8)
batches = [fp.synth.make_queries(spec, cent, 64, 32, seed=3000 + i) for i in range(12)]
ref = [R.search_arrays(index, q, params) for q in batches[:2]]


def run_split(q, K):
    parts = np.array_split(np.arange(q.shape[0]), K)
    out = [None] * K
    def work(i):
        out[i] = R.search_arrays(index, np.ascontiguousarray(q[parts[i]]), params)
    th = [threading.Thread(target=work, args=(i,)) for i in range(1, K)]
    for t in th: t.start()
    work(0)
    for t in th: t.join()
    return [np.concatenate([o[j] for o in out]) for j in range(3)]


for K in [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4]:
    for i in range(4):
        got = run_split(batches[i % 2], K)
    assert np.array_equal(got[0], ref[1][0]) and np.array_equal(got[1], ref[1][1])
    t0 = time.perf_counter()
    n = 0
    for rep in range(3):
        for q in batches:
            run_split(q, K)
            n += 1
    dt = (time.perf_counter() - t0) / n
    print(f"K={K}: {dt*1e3:.3f} ms per batch of 64 = {64/dt:.0f} q/s", flush=True)
