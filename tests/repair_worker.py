"""GPU worker for tests/test_hip_parity.py::test_maxsim_repair_vs_oracle.

The MFMA MaxSim pass certifies every column maximum it cannot prove equal to the reference's (ascending-k fp32 chain, one
fp16 rounding); the repair kernel recomputes the flagged columns with the reference's chain.
  FP_MAXSIM_REPAIR=2  every flagged document is repaired: EVERY returned score must equal the oracle's exact score of that
                      document bit for bit, and the id lists must be the oracle's (documents the oracle scores identically
                      may be permuted among themselves);
  FP_MAXSIM_REPAIR=1  (the default) only near-tied documents are repaired: the id lists must still be the oracle's."""
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

R = fp.fast_plaid_rust


def mk(a):
    return R.construct_index(a["nbits"], a["centroids"], None, None, a["bucket_weights"], a["ivf"], a["ivf_lengths"],
                             a["doc_codes"], a["doc_residuals"], a["doc_lengths"], "cuda:0", False)


def oracle(a):
    return OC.OracleIndex(nbits=a["nbits"], centroids=a["centroids"], bucket_weights=a["bucket_weights"], ivf=a["ivf"],
                          ivf_lengths=a["ivf_lengths"], doc_codes=a["doc_codes"], doc_residuals=a["doc_residuals"],
                          doc_lengths=a["doc_lengths"])


def main():
    mode = int(os.environ.get("FP_MAXSIM_REPAIR", "1"))
    assert mode in (1, 2)
    from test_hip_parity import _same_order_modulo_ref_ties
    n_scores = n_lists = 0
    for (dim, nbits, n_docs, C, Q, n_full, n_probe, top_k, seed) in (
            (128, 4, 20000, 2048, 32, 4096, 8, 500, 1), (128, 2, 20000, 2048, 32, 2048, 8, 300, 2), (64, 4, 20000, 1024, 32, 2048, 8, 300, 3),
            (64, 2, 12000, 1024, 20, 1024, 4, 200, 4), (96, 4, 12000, 1024, 32, 2048, 8, 300, 5), (48, 2, 12000, 512, 32, 1024, 8, 200, 6),
            (128, 4, 12000, 1024, 50, 1024, 8, 200, 7), (48, 4, 8000, 512, 7, 512, 4, 100, 8), (96, 2, 8000, 512, 64, 1024, 8, 200, 9),
            # round 3: shapes that moved from the exact-chain kernel to the MFMA kernel (k_maxsim6)
            (128, 8, 8000, 512, 32, 1024, 8, 200, 10), (128, 1, 8000, 512, 32, 1024, 8, 200, 11), (256, 4, 6000, 512, 32, 1024, 8, 200, 12),
            (256, 2, 6000, 512, 40, 512, 4, 100, 13)):
        spec = fp.synth.SynthSpec(n_docs=n_docs, doc_len=40, n_centroids=C, dim=dim, nbits=nbits, variable_len=True, seed=seed)
        arr = fp.synth.host_index_arrays(spec)
        idx, orc = mk(arr), oracle(arr)
        q = fp.synth.make_queries(spec, arr["centroids"], 6, Q)
        params = R.SearchParameters(2000, n_full, top_k, n_probe)
        pids, scores, counts = R.search_arrays(idx, q, params)
        ref = orc.search(q, top_k, n_full, n_probe, nthreads=4)
        for b in range(q.shape[0]):
            n = int(counts[b])
            assert n == len(ref[b][0]), (dim, nbits, b, n, len(ref[b][0]))
            ex = orc.exact_scores(q[b], pids[b, :n])
            if mode == 2:
                bad = np.flatnonzero(scores[b, :n].view(np.uint32) != ex.view(np.uint32))
                assert bad.size == 0, (dim, nbits, b, bad[:5], scores[b, bad[:5]], ex[bad[:5]])
                n_scores += n
            _same_order_modulo_ref_ties(pids[b, :n], ref[b][0], dict(zip(np.asarray(ref[b][0]).tolist(), np.asarray(ref[b][1]).tolist())))
            n_lists += 1
        if test_opt("ms_rinv_hard_every") and fp.fast_plaid_rust is R and dim % 32 == 0 and not (dim == 96 and nbits == 2):
            assert idx.n_hard_tokens > 0, (dim, nbits)   # the forced exact-path tokens were really marked
    print("REPAIR_OK mode", mode, "scores", n_scores, "lists", n_lists)


if __name__ == "__main__":
    main()
