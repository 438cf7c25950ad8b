#!/bin/bash
cd $GRAFT_REPO_ROOT
for d in 0 1 2 3 4; do
FP_LZ_DBG=$d FP_GRAPH=0 timeout 300 python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, ".")
import fast_plaid_amd as fp
R = fp.fast_plaid_rust
C = fp.synth.default_num_centroids(1_000_000 * 128)
spec = fp.synth.SynthSpec(n_docs=1_000_000, doc_len=128, n_centroids=C, seed=42, dim=128)
cent = fp.synth.centroids(spec)
index = R.construct_synthetic_index(spec, "cuda:0", centroids=cent, bucket_weights=fp.synth.bucket_weights(spec))
params = R.SearchParameters(2000, 4096, 1000, 8)
R.set_graph_replay(False)
acc = []
for i in range(8):
    q = fp.synth.make_queries(spec, cent, 64, 32, seed=2000 + i)
    R.search_arrays(index, q, params)
    acc.append(R.last_search_timings().get("S5 select", 0.0))
print("dbg", os.environ["FP_LZ_DBG"], "S5 select ms", np.round(acc[3:], 4).tolist(), flush=True)
PY
done
