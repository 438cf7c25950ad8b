"""GPU worker for tests/test_hip_parity.py::test_lazy_centroid_scores: S1's lazy form (FpLazyS1, fp_internal.h -- upper candidates
stored, the probe / the selection settle what they use) under fp_search against the EAGER form under fp_search_trace (every score
certified and repaired inside S1) and against the oracle.  LAZY_EXPECT=1: the batches must report the lazy form; 0: the eager one
(FP_S1_EXACT=1, or FP_TEST=lz_gcap=.. so small that every batch overflows its selection list and is run again eagerly)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import fast_plaid_amd as fp  # noqa: E402
import plaid_oracle as OC  # noqa: E402
from parity import check_final  # noqa: E402

R = fp.fast_plaid_rust
EXPECT = int(os.environ.get("LAZY_EXPECT", "1"))


def mk(a):
    return R.construct_index(a["nbits"], a["centroids"], None, None, a["bucket_weights"], a["ivf"], a["ivf_lengths"],
                             a["doc_codes"], a["doc_residuals"], a["doc_lengths"], "cuda:0", False)


def main():
    R.set_graph_replay(False)   # (a replayed graph reports lazy == -1; the graph path has its own test)
    seen_lazy = 0
    shapes = ((30000, 4096, 32, 512, 8, 24, 1), (20000, 2048, 20, 64, 4, 17, 2), (30000, 65536, 32, 256, 8, 20, 3),
              (20000, 2048, 70, 256, 8, 18, 4), (8000, 512, 7, 32, 2, 3, 5), (25000, 262144, 32, 256, 8, 19, 6),
              (30000, 4096, 32, 6000, 8, 9, 7))   # (the last: a rerank list of 1500, not a power of two and beyond one key per thread)
    for (n_docs, C, Q, n_full, n_probe, B, seed) in shapes:
        spec = fp.synth.SynthSpec(n_docs=n_docs, doc_len=48, n_centroids=C, variable_len=True, seed=seed)
        arr = fp.synth.host_index_arrays(spec)
        q = fp.synth.make_queries(spec, arr["centroids"], B, Q)
        expect = EXPECT
        if seed == 2:   # an unnormalised query
            q = q.astype(np.float32)
            q[1] *= 7.5
            q = q.astype(np.float16)
        if seed == 3:   # a query of tiny norm: its scores sit where fp16 is finest and S1's window spans several fp16 steps (the
            q = q.astype(np.float32)   # probe threshold and the selection's slack must follow the window, not "one step")
            q[3] *= 0.002
            q[4] *= 0.03
            q = q.astype(np.float16)
        if seed == 5:   # a query whose second half is zero rows (what list inputs of unequal lengths are padded with,
            q[2, Q // 2:] = 0   # fast_plaid.py:772-780): every centroid ties at 0 in those columns.  Round 6: the probe takes the
            # lowest-numbered cells for such a column without collecting anything and k_lz_exact takes its stored zeros as exact, so
            # the batch stays on the lazy form (until then it overflowed the probe's tie room and was run again eagerly)
        idx = mk(arr)
        # top_k = the whole rerank list (n_full / 4): a wrong document AT THE CUT of the lazy selection is then in the compared lists
        # wherever its exact score ranks it (with top_k = 25 it was visible only inside the top 25)
        K = max(n_full // 4, 1)
        params = R.SearchParameters(2000, n_full, K, n_probe)
        for rep in range(3):   # waited-for, speculative, speculative again
            pids, scores, counts = R.search_arrays(idx, q, params)
            lz = R.last_s1_counts()["lazy"]
            assert lz == expect, f"shape {seed} rep {rep}: lazy {lz}, expected {expect}"
            seen_lazy += int(lz == 1)
        for b in range(B):
            h = R.search_trace(idx, q[b], params)
            assert counts[b] == len(h["pids"]), (seed, b, counts[b], len(h["pids"]))
            assert np.array_equal(pids[b, : counts[b]], h["pids"]), (seed, b)
            assert np.array_equal(scores[b, : counts[b]], h["scores"]), (seed, b)
        orc = OC.OracleIndex(nbits=arr["nbits"], centroids=arr["centroids"], bucket_weights=arr["bucket_weights"], ivf=arr["ivf"],
                             ivf_lengths=arr["ivf_lengths"], doc_codes=arr["doc_codes"], doc_residuals=arr["doc_residuals"],
                             doc_lengths=arr["doc_lengths"])
        nb = min(B, 6)
        ref = orc.search(q[:nb], K, n_full, n_probe, nthreads=4)
        for b in range(nb):
            assert counts[b] == len(ref[b][0]), (seed, b, counts[b], len(ref[b][0]))
            check_final(pids[b, : counts[b]], scores[b, : counts[b]], ref[b][0], ref[b][1], K)
    print("LAZY_OK", seen_lazy)


if __name__ == "__main__":
    main()
