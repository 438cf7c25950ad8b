#!/usr/bin/env python
"""GPU-box diagnostic (see debug_parity.py): is the MFMA pass deterministic across launches, and what does the repair do?
Run three times with FP_MAXSIM_REPAIR = 0 / 1 / 2 (DBG_TAG names the output), then once with DBG_TAG=compare."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
OUT = os.path.join(ROOT, "gpurun_out")
tag = os.environ.get("DBG_TAG", "r1")
nq = 8
if tag != "compare":
    import fast_plaid_amd as fp
    R = fp.fast_plaid_rust
    spec = fp.synth.SynthSpec(n_docs=1_000_000, doc_len=128, n_centroids=131072, seed=42)
    cent = fp.synth.centroids(spec); bw = fp.synth.bucket_weights(spec)
    dev = R.construct_synthetic_index(spec, "cuda:0", centroids=cent)
    q = fp.synth.make_queries(spec, cent, nq, 32, seed=4242)
    params = R.SearchParameters(2000, 4096, 1000, 8)
    pids, scores, counts = R.search_arrays(dev, q, params)
    out = dict(pids=pids, scores=scores)
    if tag == "r0":
        cols = [R.maxsim_columns(dev, q[b], pids[b]) for b in range(nq)]
        out["col_scores"] = np.stack([c["scores"] for c in cols]); out["unc"] = np.stack([c["unc"] for c in cols])
        out["flags"] = np.stack([c["flags"][:, 0] for c in cols]); out["col_max"] = np.stack([c["col_max"].view(np.uint16) for c in cols])
        import plaid_oracle as OC
        arr = R.export_index_arrays(dev, centroids=cent, bucket_weights=bw)
        orc = OC.OracleIndex(nbits=4, centroids=cent, bucket_weights=bw, ivf=arr["ivf"], ivf_lengths=arr["ivf_lengths"], doc_codes=arr["doc_codes"],
                             doc_residuals=arr["doc_residuals"], doc_lengths=arr["doc_lengths"])
        ex = np.stack([orc.exact_scores(q[b], pids[b]) for b in range(nq)])
        out["oracle_exact"] = ex
        # oracle per-column maxima for the first 2 queries
        oc = np.zeros((2, pids.shape[1], 32), np.uint16)
        for b in range(2):
            for i, p in enumerate(pids[b].tolist()):
                oc[b, i] = orc.token_scores(q[b], p).max(axis=1).view(np.uint16)
        out["oracle_cols"] = oc
    np.savez(os.path.join(OUT, f"dbg_{tag}.npz"), **out)
    print(tag, "saved", R.last_search_counts())
else:
    z0, z1, z2 = (np.load(os.path.join(OUT, f"dbg_{t}.npz")) for t in ("r0", "r1", "r2"))
    for b in range(nq):
        m0 = dict(zip(z0["pids"][b].tolist(), z0["scores"][b].tolist()))
        cs = dict(zip(z0["pids"][b].tolist(), z0["col_scores"][b].tolist()))
        un = dict(zip(z0["pids"][b].tolist(), z0["unc"][b].tolist()))
        ox = dict(zip(z0["pids"][b].tolist(), z0["oracle_exact"][b].tolist()))
        nd = sum(1 for p in m0 if m0[p] != cs[p])
        print(f"q{b}: norepair search vs separate MFMA pass: {nd} of {len(m0)} scores differ (non-determinism if > 0)")
        print(f"     MFMA pass != oracle: {sum(1 for p in m0 if cs[p] != ox[p])}; of those unflagged: {sum(1 for p in m0 if cs[p] != ox[p] and un[p] == 0)}; flagged docs {sum(1 for p in m0 if un[p] > 0)}")
        for name, z in (("near-tied repair", z1), ("repair-all", z2)):
            m = dict(zip(z["pids"][b].tolist(), z["scores"][b].tolist()))
            common = [p for p in m if p in ox]
            bad = [p for p in common if m[p] != ox[p]]
            changed = [p for p in common if p in cs and m[p] != cs[p]]
            worse = [p for p in changed if m[p] != ox[p]]
            print(f"     {name}: {len(bad)} scores != oracle; {len(changed)} changed by the repair, {len(worse)} of them to a non-oracle value; "
                  f"unrepaired-but-different {sum(1 for p in bad if p in cs and m[p] == cs[p])}")
            if name == "repair-all" and worse[:3]:
                for p in worse[:3]:
                    i = z0["pids"][b].tolist().index(p)
                    print("        doc", p, "mfma", cs[p], "repaired", m[p], "oracle", ox[p], "flags %08x" % int(z0["flags"][b][i]), "unc", un[p])
    if "oracle_cols" in z0:
        for b in range(2):
            g = z0["col_max"][b][:, :32]; o = z0["oracle_cols"][b]
            fl = z0["flags"][b]
            flagged = ((fl[:, None] >> np.arange(32)[None, :]) & 1).astype(bool)
            diff = g != o
            print(f"q{b}: columns differing from the oracle: {int(diff.sum())}, of which unflagged: {int((diff & ~flagged).sum())}; flagged columns {int(flagged.sum())} of {flagged.size}")
