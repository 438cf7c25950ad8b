"""GPU worker for tests/test_hip_parity.py::test_index_build_with_capped_grids: runs with
FP_TEST=grid_cap=3 so every index-build kernel whose grid scales with the corpus walks its
grid-stride loop many times (the path a > 2^32 work-item launch takes at 10 M documents)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fptest_env import test_opt  # noqa: E402

import fast_plaid_amd as fp  # noqa: E402
import plaid_oracle as OC  # noqa: E402
from parity import check_trace  # noqa: E402


def main():
    assert test_opt("grid_cap") == "3"
    R = fp.fast_plaid_rust
    # device generator + sort + unique + IVF build == numpy twin
    spec = fp.synth.SynthSpec(n_docs=3000, doc_len=48, n_centroids=512, variable_len=True, seed=42)
    host = fp.synth.host_index_arrays(spec)
    dev = R.construct_synthetic_index(spec, "cuda:0", centroids=host["centroids"])
    offs = np.concatenate([[0], np.cumsum(host["doc_lengths"])])
    for pid in (0, 1, 777, 2999):
        codes, res = R.read_doc(dev, pid)
        assert np.array_equal(codes, host["doc_codes"][offs[pid]: offs[pid + 1]])
        assert np.array_equal(res, host["doc_residuals"][offs[pid]: offs[pid + 1]])
    ivf_off = np.concatenate([[0], np.cumsum(host["ivf_lengths"].astype(np.int64))])
    for cell in (0, 1, 100, 511):
        assert np.array_equal(R.read_ivf(dev, cell), host["ivf"][ivf_off[cell]: ivf_off[cell + 1]])
    # uploaded index (narrow + sort + unique) searched against the oracle
    q = fp.synth.make_queries(spec, host["centroids"], 3, 32)
    hip = R.construct_index(host["nbits"], host["centroids"], None, None, host["bucket_weights"], host["ivf"], host["ivf_lengths"],
                            host["doc_codes"], host["doc_residuals"], host["doc_lengths"], "cuda:0", False)
    orc = OC.OracleIndex(nbits=host["nbits"], centroids=host["centroids"], bucket_weights=host["bucket_weights"], ivf=host["ivf"],
                         ivf_lengths=host["ivf_lengths"], doc_codes=host["doc_codes"], doc_residuals=host["doc_residuals"],
                         doc_lengths=host["doc_lengths"])
    params = R.SearchParameters(2000, 256, 20, 4)
    for idx in (dev, hip):
        for b in range(3):
            check_trace(R.search_trace(idx, q[b], params), orc.search_trace(q[b], 20, 256, 4), 32, 4, 256, 20)
    # s_budget_kb (set by the test next to grid_cap in FP_TEST): the batch is cut into sub-batches of the centroid-score table's
    # budget; results must not depend on the cut
    if test_opt("s_budget_kb"):
        qb = fp.synth.make_queries(spec, host["centroids"], 23, 32)
        pids, scores, counts = R.search_arrays(hip, qb, params)
        assert R.last_search_counts()["sub_batches"] > 1, "the budget did not force sub-batching"
        for b in range(23):
            t = R.search_trace(hip, qb[b], params)
            assert np.array_equal(pids[b, : counts[b]], t["pids"]) and np.array_equal(scores[b, : counts[b]], t["scores"])
        subs = [[int(x) for x in np.random.default_rng(b).integers(0, 3000, 30)] for b in range(23)]
        p2, s2, c2 = R.search_arrays(hip, qb, params, subs)       # per-query subsets must follow their queries across the cut
        for b in range(23):
            t = R.search_trace(hip, qb[b], params, subs[b])
            assert np.array_equal(p2[b, : c2[b]], t["pids"])
    print("GRID_CAP_OK")


if __name__ == "__main__":
    main()
