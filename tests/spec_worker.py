"""GPU worker for tests/test_hip_parity.py::test_candidate_capacity_speculation.

fp_search sizes S4 / S5 from the candidate total of EARLIER batches of the same shape instead of waiting for the current one
in the middle of the pipeline; a batch that exceeds the learnt capacity is emptied on the device and run again.  This worker
drives both outcomes: repeated batches (the capacity holds), batches whose totals grow (re-run), and -- with
FP_TEST=spec_cap_pct=50 set by the test -- a capacity that is always too small (every batch after the first is run twice).  Every
result must equal fp_search_trace, which never speculates."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import fast_plaid_amd as fp  # noqa: E402


def main():
    R = fp.fast_plaid_rust
    spec = fp.synth.SynthSpec(n_docs=20000, doc_len=32, n_centroids=1024, variable_len=True, seed=7)
    host = fp.synth.host_index_arrays(spec)
    dev = R.construct_synthetic_index(spec, "cuda:0", centroids=host["centroids"])
    for n_probe, n_full, top_k in ((2, 256, 20), (8, 1024, 50), (2, 256, 20), (16, 1024, 50)):
        params = R.SearchParameters(2000, n_full, top_k, n_probe)
        runs = []
        for rep in range(5):
            # the same shape with other queries: the candidate total moves from call to call (rep 3 repeats rep 2's batch)
            q = fp.synth.make_queries(spec, host["centroids"], 6, 32, seed=100 + min(rep, 3) - (rep == 3))
            pids, scores, counts = R.search_arrays(dev, q, params)
            runs.append((q, pids, scores, counts, R.last_search_counts()["candidates"]))
        assert runs[3][4] == runs[2][4] and np.array_equal(runs[3][1], runs[2][1]) and np.array_equal(runs[3][2], runs[2][2])
        for q, pids, scores, counts, cand in runs:
            assert cand > 0
            for b in range(q.shape[0]):
                t = R.search_trace(dev, q[b], params)   # a trace waits for the candidate total (and scores every candidate)
                assert counts[b] == len(t["pids"]), (n_probe, b, counts[b], len(t["pids"]))
                assert np.array_equal(pids[b, : counts[b]], t["pids"]), (n_probe, b)
                assert np.array_equal(scores[b, : counts[b]], t["scores"]), (n_probe, b)
    print("SPEC_OK")


if __name__ == "__main__":
    main()
