"""GPU worker for tests/test_hip_parity.py::test_bound_and_refine_forced / test_level0_forced:
FP_APPROX_IMPL=q8 forces the 8-bit bound stage of S4 on small corpora, FP_APPROX_IMPL=l0 the level-0
stage (scalar excess bound per centroid in LDS); both are otherwise chosen only for large centroid
tables / candidate sets.  fp_search (pruned path) must equal fp_search_trace (exact score of every
candidate) bit for bit, and the oracle within the usual tolerances."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import fast_plaid_amd as fp  # noqa: E402
import plaid_oracle as OC  # noqa: E402
from conftest import GOLDEN_DIR, golden_cases  # noqa: E402
from parity import check_final, check_trace  # noqa: E402

R = fp.fast_plaid_rust


def mk(a, **kw):
    return R.construct_index(a["nbits"], a["centroids"], None, None, a["bucket_weights"], a["ivf"], a["ivf_lengths"],
                             a["doc_codes"], a["doc_residuals"], a["doc_lengths"], "cuda:0", False, **kw)


def same_as_trace(idx, q, params, subs=None):
    pids, scores, counts = R.search_arrays(idx, q, params, subs)
    c = R.last_search_counts()
    for b in range(q.shape[0]):
        h = R.search_trace(idx, q[b], params, None if subs is None else subs[b])
        assert counts[b] == len(h["pids"]), (b, counts[b], len(h["pids"]))
        assert np.array_equal(pids[b, : counts[b]], h["pids"]), b
        assert np.array_equal(scores[b, : counts[b]], h["scores"]), b
    return c


def main():
    assert os.environ.get("FP_APPROX_IMPL") in ("q8", "l0", "l0h")
    # committed fixtures (subset, empty documents, zero-padded query rows, unnormalised documents, ...)
    for name in golden_cases():
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        arr = {k: z[k] for k in ("centroids", "bucket_weights", "ivf", "ivf_lengths", "doc_codes", "doc_residuals", "doc_lengths")}
        arr["nbits"] = int(z["nbits"])
        n_probe, n_full, top_k, bs = (int(x) for x in z["params"])
        q = z["queries"]
        subs = [z[f"subset_{b}"].tolist() for b in range(q.shape[0])] if "subset_0" in z else None
        same_as_trace(mk(arr), q, R.SearchParameters(bs, n_full, top_k, n_probe), subs)
    # synthetic corpora where the cut actually prunes (candidates >> R), Q < 32 exercises the column mask
    pruned = 0
    for (n_docs, C, Q, n_full, n_probe, seed) in ((20000, 2048, 32, 256, 8, 1), (20000, 2048, 20, 64, 4, 2), (8000, 512, 7, 32, 2, 3),
                                                  (30000, 4096, 32, 4096, 8, 4), (20000, 2048, 50, 256, 8, 5), (15000, 1024, 64, 128, 4, 6),
                                                  (15000, 1024, 33, 64, 8, 7),    # q_len 33..64: two 32-column chunks
                                                  (20000, 262144, 32, 256, 8, 11),   # > 2^17 centroids: level 0 walks two centroid ranges
                                                  (12000, 524288, 20, 128, 4, 12),   # four ranges
                                                  (20000, 2048, 70, 256, 8, 13), (20000, 4096, 128, 256, 8, 14),   # q_len 65..128: padded to 128 columns,
                                                  (15000, 262144, 100, 128, 4, 15)):                               # level 0 sums four column groups per tile
        spec = fp.synth.SynthSpec(n_docs=n_docs, doc_len=48, n_centroids=C, variable_len=True, seed=seed)
        arr = fp.synth.host_index_arrays(spec)
        q = fp.synth.make_queries(spec, arr["centroids"], 6, Q)
        idx = mk(arr)
        params = R.SearchParameters(2000, n_full, 25, n_probe)
        c = same_as_trace(idx, q, params)
        assert c["approx_exact"] <= c["candidates"]
        pruned += int(c["approx_exact"] < c["candidates"])
        orc = OC.OracleIndex(nbits=arr["nbits"], centroids=arr["centroids"], bucket_weights=arr["bucket_weights"], ivf=arr["ivf"],
                             ivf_lengths=arr["ivf_lengths"], doc_codes=arr["doc_codes"], doc_residuals=arr["doc_residuals"],
                             doc_lengths=arr["doc_lengths"])
        ref = orc.search(q, 25, n_full, n_probe, nthreads=4)
        pids, scores, counts = R.search_arrays(idx, q, params)
        for b in range(q.shape[0]):   # the oracle's ids in its order (modulo its exact ties), scores within 1e-3
            check_final(pids[b, : counts[b]], scores[b, : counts[b]], ref[b][0], ref[b][1], 25)
    assert pruned >= 5, "the bound stage never pruned: the test does not exercise it"
    # values outside the bin range: unnormalised queries scaled x3 (S up to 3: bin 255 voids the upper bound) and x(-3)
    spec = fp.synth.SynthSpec(n_docs=20000, doc_len=48, n_centroids=2048, variable_len=True, seed=9)
    arr = fp.synth.host_index_arrays(spec)
    idx = mk(arr)
    q = fp.synth.make_queries(spec, arr["centroids"], 4, 32).astype(np.float32)
    for scale in (3.0, -3.0, 0.01):
        same_as_trace(idx, (q * scale).astype(np.float16), R.SearchParameters(2000, 128, 25, 8))
    # the odd-shape random corpora of test_randomized_shapes_vs_oracle (those with q_len <= 64)
    from test_hip_parity import RANDOM_SHAPES, _random_arrays
    for shape in RANDOM_SHAPES:
        n_docs, max_len, C, dim, nbits, B, Q, n_probe, n_full, top_k, use_subset = shape
        if Q > 128:
            continue
        rng = np.random.default_rng(hash(shape) & 0xFFFFFFFF)
        arr = _random_arrays(rng, n_docs, max_len, C, dim, nbits)
        pick = rng.integers(0, C, (B, Q))
        q = arr["centroids"][pick].astype(np.float32) + 0.3 * rng.standard_normal((B, Q, dim), dtype=np.float32) / np.sqrt(dim)
        q /= np.linalg.norm(q, axis=2, keepdims=True)
        subs = [rng.integers(0, n_docs, 20).tolist() for _ in range(B)] if use_subset else None
        same_as_trace(mk(arr), q.astype(np.float16), R.SearchParameters(2000, n_full, top_k, n_probe), subs)
    # 4000 identical documents: every bound and every exact score ties at the cut
    from test_hip_parity import _tied_copies_arrays
    arr, q = _tied_copies_arrays(fp, np.random.default_rng(17))
    idx = mk(arr)
    for n_full, top_k in ((400, 50), (8, 5)):
        same_as_trace(idx, q, R.SearchParameters(2000, n_full, top_k, 4))
    print("Q8_OK")


if __name__ == "__main__":
    main()
