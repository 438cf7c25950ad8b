"""CPU tests: the plain-C oracle (oracle/plaid_oracle.c) against the committed golden
fixtures produced by the ATen op-for-op restatement (tests/golden/make_golden.py).
Integer/index stages must be bit-exact; fp16/fp32 values must be bit-exact too (the C
oracle reproduces ATen's CPU accumulation order)."""
import os

import numpy as np
import pytest

import plaid_oracle as OC
from conftest import GOLDEN_DIR, golden_cases


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    idx = OC.OracleIndex(
        nbits=int(z["nbits"]), centroids=z["centroids"], bucket_weights=z["bucket_weights"], ivf=z["ivf"],
        ivf_lengths=z["ivf_lengths"], doc_codes=z["doc_codes"], doc_residuals=z["doc_residuals"],
        doc_lengths=z["doc_lengths"])
    return z, idx


@pytest.mark.parametrize("name", golden_cases())
def test_c_oracle_matches_aten_golden(name):
    z, idx = load_case(name)
    n_probe, n_full, top_k, _ = (int(x) for x in z["params"])
    q = z["queries"]
    for b in range(q.shape[0]):
        sub = z[f"subset_{b}"] if f"subset_{b}" in z else None
        t = idx.search_trace(q[b], top_k, n_full, n_probe, sub)
        if f"S_{b}" in z:
            assert np.array_equal(t["S"].view(np.uint16), z[f"S_{b}"].view(np.uint16)), "S1 centroid scores"
        if f"cells_nz_{b}" in z:  # zero-padded query tokens probe arbitrary cells (implementation-defined)
            assert set(z[f"cells_nz_{b}"].tolist()) <= set(t["cells"].tolist())
            assert len(t["cells"]) <= len(z[f"cells_nz_{b}"]) + n_probe
        else:
            assert np.array_equal(t["cells"], z[f"cells_{b}"]), "S2 cells"
        assert np.array_equal(t["cand"], z[f"cand_{b}"]), "S3 candidates"
        assert np.array_equal(t["approx"], z[f"approx_{b}"]), "S4 approx scores"
        assert np.array_equal(t["rerank"], z[f"rerank_{b}"]), "S5 rerank set"
        assert np.array_equal(t["exact"], z[f"exact_{b}"]), "S7 exact scores"
        assert np.array_equal(t["pids"], z[f"pids_{b}"]), "S8 final ids"
        assert np.array_equal(t["scores"], z[f"scores_{b}"]), "S8 final scores"


@pytest.mark.parametrize("name", ["base_d128_nb4", "d64_nb2", "unnormalised_docs"])
def test_c_oracle_decompress_matches_aten(name):
    z, idx = load_case(name)
    n = z["decomp_sample"].shape[0]
    out = idx.decompress(z["doc_codes"][:n], z["doc_residuals"][:n])
    assert np.array_equal(out.view(np.uint16), z["decomp_sample"].view(np.uint16))


def test_search_many_batch_and_threads():
    z, idx = load_case("base_d128_nb4")
    n_probe, n_full, top_k, _ = (int(x) for x in z["params"])
    r1 = idx.search(z["queries"], top_k, n_full, n_probe, nthreads=1)
    r4 = idx.search(z["queries"], top_k, n_full, n_probe, nthreads=4)
    for b, ((p1, s1), (p4, s4)) in enumerate(zip(r1, r4)):
        assert np.array_equal(p1, z[f"pids_{b}"]) and np.array_equal(p4, p1)
        assert np.array_equal(s1, z[f"scores_{b}"]) and np.array_equal(s4, s1)


def test_structural_properties_like_reference_tests():
    """tests/test.py:939-974 (descending order, determinism) and :872-886 (top_k > n_docs)."""
    z, idx = load_case("topk_gt_ndocs")
    n_probe, n_full, top_k, _ = (int(x) for x in z["params"])
    res = idx.search(z["queries"], top_k, n_full, n_probe)
    for p, s in res:
        assert len(p) <= idx.n_docs
        assert np.all(np.diff(s) <= 0)
        assert len(set(p.tolist())) == len(p)
    res2 = idx.search(z["queries"], top_k, n_full, n_probe)
    assert all(np.array_equal(a[0], b[0]) for a, b in zip(res, res2))


def test_compress_only_index_raises():
    z, _ = load_case("d64_nb2")
    idx = OC.OracleIndex(nbits=int(z["nbits"]), centroids=z["centroids"], bucket_weights=z["bucket_weights"],
                         ivf=None, ivf_lengths=None, doc_codes=z["doc_codes"], doc_residuals=z["doc_residuals"],
                         doc_lengths=z["doc_lengths"])
    with pytest.raises(ValueError, match="compress_only"):
        idx.search(z["queries"], 5)


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_token_score_matrices_match_aten_goldens(name):
    """pl_token_scores == the [query_tokens, doc_tokens] fp16 matrices the ATen restatement returned
    for the first hits (search.rs:651-653, :668-686), bit for bit."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    orc = OC.OracleIndex(nbits=int(z["nbits"]), centroids=z["centroids"], bucket_weights=z["bucket_weights"], ivf=z["ivf"],
                         ivf_lengths=z["ivf_lengths"], doc_codes=z["doc_codes"], doc_residuals=z["doc_residuals"],
                         doc_lengths=z["doc_lengths"])
    checked = 0
    for b in range(2):
        for i in range(3):
            key = f"tokmat_{b}_{i}"
            if key not in z:
                continue
            pid = int(z[f"pids_{b}"][i])
            got = orc.token_scores(z["queries"][b], pid)
            assert got.shape == z[key].shape
            assert np.array_equal(got.view(np.uint16), z[key].view(np.uint16)), (name, b, i)
            checked += 1
    assert checked >= 1
