"""S1 exact-mode lab (GPU): S of fp_search_trace against the C oracle, bit for bit, on goldens and synthetic corpora; run it with
FP_S1_EXACT=0/1/2, FP_S1_STATS=1 for the counters."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import fast_plaid_amd as fp
import plaid_oracle as OC
from conftest import GOLDEN_DIR, golden_cases
R = fp.fast_plaid_rust

def oracle_of(arr):
    return OC.OracleIndex(nbits=arr["nbits"], centroids=arr["centroids"], bucket_weights=arr["bucket_weights"], ivf=arr.get("ivf"),
                          ivf_lengths=arr.get("ivf_lengths"), doc_codes=arr["doc_codes"], doc_residuals=arr["doc_residuals"], doc_lengths=arr["doc_lengths"])

def hip_of(arr):
    return R.construct_index(arr["nbits"], arr["centroids"], arr.get("avg_residual"), arr.get("bucket_cutoffs"), arr["bucket_weights"],
                             arr.get("ivf"), arr.get("ivf_lengths"), arr["doc_codes"], arr["doc_residuals"], arr["doc_lengths"], "cuda:0", False)

tot = bad = 0
for name in golden_cases():
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    arr = {k: z[k] for k in ("centroids", "avg_residual", "bucket_cutoffs", "bucket_weights", "ivf", "ivf_lengths", "doc_codes", "doc_residuals", "doc_lengths")}
    arr["nbits"] = int(z["nbits"])
    n_probe, n_full, top_k, bs = (int(x) for x in z["params"])
    hip, orc = hip_of(arr), oracle_of(arr)
    params = R.SearchParameters(bs, n_full, top_k, n_probe)
    q = z["queries"]
    nb = 0
    for b in range(q.shape[0]):
        sub = z[f"subset_{b}"] if f"subset_{b}" in z else None
        h = R.search_trace(hip, q[b], params, sub)
        o = orc.search_trace(q[b], top_k, n_full, n_probe, sub)
        d = (h["S"].view(np.uint16) != o["S"].view(np.uint16))
        nb += int(d.sum()); tot += d.size
        same = np.array_equal(h["pids"], o["pids"]) and np.array_equal(h["scores"], o["scores"]) if len(o["pids"]) else True
    bad += nb
    print(f"golden {name}: S mismatches {nb}  s1 {R.last_s1_counts()}", flush=True)
print("goldens total entries", tot, "mismatches", bad, flush=True)

for (dim, C, nd, dl, Q, seed) in [(128, 1 << 13, 4000, 64, 32, 1), (128, 1 << 15, 20000, 48, 32, 2), (64, 1 << 13, 4000, 64, 40, 3), (96, 1 << 12, 3000, 64, 32, 4),
                                  (256, 1 << 12, 2000, 48, 20, 5), (40, 1 << 11, 2000, 48, 32, 6), (128, 1 << 17, 30000, 32, 32, 7), (160, 1 << 12, 2000, 48, 32, 8), (72, 1 << 12, 2000, 48, 32, 9)]:
    spec = fp.synth.SynthSpec(n_docs=nd, doc_len=dl, n_centroids=C, dim=dim, variable_len=True, seed=seed)
    arr = fp.synth.host_index_arrays(spec)
    q = fp.synth.make_queries(spec, arr["centroids"], 3, Q, seed=seed + 50)
    if seed % 2: q[1] *= np.float16(0.37)      # an unnormalised query
    q[2, Q // 2:] = 0                           # zero rows inside a query
    hip, orc = hip_of(arr), oracle_of(arr)
    params = R.SearchParameters(1, 256, 10, 8)
    nb = n = 0
    t0 = time.time()
    for b in range(3):
        h = R.search_trace(hip, q[b], params, None)
        o = orc.search_trace(q[b], 10, 256, 8, None)
        d = (h["S"].view(np.uint16) != o["S"].view(np.uint16))
        nb += int(d.sum()); n += d.size
        ok = np.array_equal(h["pids"], o["pids"]) and np.array_equal(h["scores"], o["scores"]) and np.array_equal(h["approx"], o["approx"]) and np.array_equal(np.sort(h["cells"]), np.sort(o["cells"]))
        if not ok: print("   query", b, "downstream differs: pids", np.array_equal(h["pids"], o["pids"]), "scores", np.array_equal(h["scores"], o["scores"]), "approx", np.array_equal(h["approx"], o["approx"]))
    print(f"synth dim {dim} C {C}: S mismatches {nb} of {n}  s1 {R.last_s1_counts()}  ({time.time()-t0:.1f}s)", flush=True)
