"""S1 certification counters and stage times on BASELINE cfg2 (GPU).  Usage: [FP_S1_EXACT=0|1|2|3] [FP_TEST=s1_w0_log2=-21] FP_S1_STATS=1
python tools/s1_stats_cfg2.py [batches] [docs] [dim]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fast_plaid_amd as fp
R = fp.fast_plaid_rust
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
docs = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 128
C = fp.synth.default_num_centroids(docs * 128)
spec = fp.synth.SynthSpec(n_docs=docs, doc_len=128, n_centroids=C, seed=42, dim=dim)
cent = fp.synth.centroids(spec)
index = R.construct_synthetic_index(spec, "cuda:0", centroids=cent, bucket_weights=fp.synth.bucket_weights(spec))
params = R.SearchParameters(2000, 4096, 1000, 8)
R.set_graph_replay(False) if hasattr(R, "set_graph_replay") else None
tot = dict(flagged=0, changed=0, slow_path=0, unflagged_differences=0)
s1 = []
for i in range(nb):
    q = fp.synth.make_queries(spec, cent, 64, 32, seed=2000 + i)
    t0 = time.time()
    pids, scores, counts = R.search_arrays(index, q, params)
    dt = time.time() - t0
    st = R.last_search_timings()
    c = R.last_s1_counts()
    for k in tot: tot[k] += c[k]
    s1.append(st.get("S1 centroid_gemm", 0.0))
    print(f"batch {i}: {dt*1e3:.2f} ms  S1 {s1[-1]:.4f} ms  s1 {c}", flush=True)
n_entries = nb * 64 * 32 * C
print("env", {k: v for k, v in os.environ.items() if k.startswith("FP_S1")}, "entries", n_entries, "totals", tot,
      "flag rate %.4f" % (tot["flagged"] / max(n_entries, 1)), "S1 ms min %.4f" % min(s1[1:] if len(s1) > 1 else s1), flush=True)
