#!/usr/bin/env python
"""GPU-box diagnostic: cfg2 at full size, per-document / per-column comparison of the exact stage with the C oracle for the
documents whose final rank differs (used to find what still separates the id lists from the oracle's)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fast_plaid_amd as fp
import plaid_oracle as OC
R = fp.fast_plaid_rust
spec = fp.synth.SynthSpec(n_docs=1_000_000, doc_len=128, n_centroids=131072, seed=42)
cent = fp.synth.centroids(spec); bw = fp.synth.bucket_weights(spec)
dev = R.construct_synthetic_index(spec, "cuda:0", centroids=cent)
nq = int(os.environ.get("DBG_Q", 16))
q = fp.synth.make_queries(spec, cent, nq, 32, seed=4242)
params = R.SearchParameters(2000, 4096, 1000, 8)
pids, scores, counts = R.search_arrays(dev, q, params)
print("repaired", R.last_search_counts())
arr = R.export_index_arrays(dev, centroids=cent, bucket_weights=bw)
orc = OC.OracleIndex(nbits=4, centroids=cent, bucket_weights=bw, ivf=arr["ivf"], ivf_lengths=arr["ivf_lengths"], doc_codes=arr["doc_codes"],
                     doc_residuals=arr["doc_residuals"], doc_lengths=arr["doc_lengths"])
ref = orc.search(q, 1000, 4096, 8, nthreads=min(nq, OC.num_procs()))
for b in range(nq):
    gp, gs = pids[b, :counts[b]], scores[b, :counts[b]]
    rp, rs = ref[b]
    if np.array_equal(gp, rp):
        continue
    rmap = dict(zip(rp.tolist(), rs.tolist())); gmap = dict(zip(gp.tolist(), gs.tolist()))
    common = [p for p in gp.tolist() if p in rmap]
    diff_score = [p for p in common if rmap[p] != gmap[p]]
    only = set(rmap) ^ set(gmap)
    print(f"query {b}: sets differ by {sorted(only)}; {len(diff_score)} common docs with a different score")
    firstbad = next((i for i, (x, y) in enumerate(zip(gp.tolist(), rp.tolist())) if x != y), None)
    print("  first differing position", firstbad, gp[firstbad - 1:firstbad + 3].tolist(), rp[firstbad - 1:firstbad + 3].tolist(),
          gs[firstbad - 1:firstbad + 3].tolist(), rs[firstbad - 1:firstbad + 3].tolist())
    look = sorted(set(gp[firstbad - 1:firstbad + 3].tolist() + rp[firstbad - 1:firstbad + 3].tolist() + diff_score[:6]))
    cols = R.maxsim_columns(dev, q[b], np.array(look, np.int64))
    for i, p in enumerate(look):
        want = orc.token_scores(q[b], p).max(axis=1)
        g = cols["col_max"][i]
        bad = np.nonzero(g.view(np.uint16) != want.view(np.uint16))[0]
        fl = [c for c in range(32) if (int(cols["flags"][i, 0]) >> c) & 1]
        print(f"  doc {p}: gpu final {gmap.get(p)} oracle {rmap.get(p)} mfma-pass {cols['scores'][i]} unc {cols['unc'][i]:.2e} flagged {fl} "
              f"cols differing from oracle {bad.tolist()} gpu {g[bad].tolist()} oracle {want[bad].tolist()} oracle-colsum {float(want.astype(np.float32).sum(dtype=np.float32))}")
print("done")
