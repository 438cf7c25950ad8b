"""Certification evidence (GPU + CPU oracle): the MaxSim kernel's column window (eps = 2^-19 |q|) and S1's score window against
the C oracle over several corpora / seeds.  For every (shape, seed): the rerank lists of nq queries, every column the kernel does
NOT flag must equal the oracle's column maximum bit for bit.  Prints one JSON line per run and a total; the summary goes to
profiles/.   python tools/cert_stats.py > gpurun_out/cert_stats.jsonl"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import fast_plaid_amd as fp
import plaid_oracle as OC
from parity import ulp_diff_f16
R = fp.fast_plaid_rust

RUNS = [  # n_docs, doc_len, centroids, nbits, dim, Q, nq, seed
    (1_000_000, 128, 131072, 4, 128, 32, 64, 777), (1_000_000, 128, 131072, 4, 128, 32, 64, 1234), (1_000_000, 128, 131072, 4, 128, 32, 64, 99),
    (100_000, 1024, 131072, 4, 128, 32, 16, 5),      # cfg4's 1024-token documents
    (2_000_000, 128, 65536, 4, 128, 32, 48, 11),     # cfg5's table
    (400_000, 128, 65536, 2, 64, 32, 32, 21), (300_000, 96, 32768, 8, 128, 50, 24, 31),
]
tot = dict(columns=0, flagged=0, flagged_and_different=0, unflagged_different=0)
for (nd, dl, C, nbits, dim, Q, nq, seed) in RUNS:
    t0 = time.time()
    spec = fp.synth.SynthSpec(n_docs=nd, doc_len=dl, n_centroids=C, dim=dim, nbits=nbits, seed=42 + seed)
    cent = fp.synth.centroids(spec)
    bw = fp.synth.bucket_weights(spec)
    dev = R.construct_synthetic_index(spec, "cuda:0", centroids=cent, bucket_weights=bw)
    q = fp.synth.make_queries(spec, cent, nq, Q, seed=seed)
    n_full = 4096
    Rr = n_full // 4
    pids, scores, counts = R.search_arrays(dev, q, R.SearchParameters(2000, n_full, Rr, 8))
    arr = R.export_index_arrays(dev, centroids=cent, bucket_weights=bw)
    orc = OC.OracleIndex(nbits=nbits, centroids=cent, bucket_weights=bw, ivf=arr["ivf"], ivf_lengths=arr["ivf_lengths"],
                         doc_codes=arr["doc_codes"], doc_residuals=arr["doc_residuals"], doc_lengths=arr["doc_lengths"])
    run = dict(shape=dict(n_docs=nd, doc_len=dl, centroids=C, nbits=nbits, dim=dim, q_len=Q, queries=nq, seed=seed), columns=0, flagged=0,
               flagged_and_different=0, unflagged_different=0)
    for b in range(nq):
        docs = np.sort(pids[b, : counts[b]])
        got = R.maxsim_columns(dev, q[b], docs)
        want = orc.column_maxima(q[b], docs)
        g = got["col_max"]
        fl = ((got["flags"][:, (np.arange(Q) // 32)] >> (np.arange(Q) % 32).astype(np.uint32)) & 1).astype(bool)
        diff = g.view(np.uint16) != want.view(np.uint16)
        run["columns"] += int(g.size); run["flagged"] += int(fl.sum()); run["flagged_and_different"] += int((diff & fl).sum())
        run["unflagged_different"] += int((diff & ~fl).sum())
        assert np.all(ulp_diff_f16(g[diff], want[diff]) <= 1)
    run["seconds"] = round(time.time() - t0, 1)
    for k in tot: tot[k] += run[k]
    print(json.dumps(run), flush=True)
    del dev
print(json.dumps(dict(total=tot)), flush=True)
assert tot["unflagged_different"] == 0
