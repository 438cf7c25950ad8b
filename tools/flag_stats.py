"""How many columns does the MFMA MaxSim flag per document?  (cfg2 corpus; the documents a search ranks at the top for each query)
usage (GPU box): python tools/flag_stats.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_plaid_amd as fp
R = fp.fast_plaid_rust
spec = fp.synth.SynthSpec(n_docs=1_000_000, doc_len=128, n_centroids=131072, dim=128, nbits=4, seed=42)
cent = fp.synth.centroids(spec)
ix = R.construct_synthetic_index(spec, "cuda:0", centroids=cent, bucket_weights=fp.synth.bucket_weights(spec))
q = fp.synth.make_queries(spec, cent, 8, 32, seed=11)
p = R.SearchParameters(2000, 4096, 1000, 8)
pids, scores, counts = R.search_arrays(ix, q, p)
tot = []
for b in range(q.shape[0]):
    d = R.maxsim_columns(ix, q[b], pids[b, : counts[b]].astype(np.int64))
    pc = np.array([bin(int(x)).count("1") for x in d["flags"][:, 0]])
    flagged_docs = (d["unc"] > 0).sum()
    tot.append((counts[b], flagged_docs, pc.mean(), pc[pc > 0].mean() if (pc > 0).any() else 0, pc.max()))
for t in tot:
    print("docs %d  with a budget %d  flagged columns/doc: mean %.2f, mean over flagged docs %.2f, max %d" % t)
