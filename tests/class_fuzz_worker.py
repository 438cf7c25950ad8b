"""GPU worker: randomized fuzz of the Python surface `fast_plaid_amd.search.FastPlaid` (the mirror of fast_plaid.search.FastPlaid,
fast_plaid.py:325-1186) through whole life cycles: create -> search -> update -> search -> delete -> search (-> a second object on the
same directory), with the query and subset forms the reference's `_prepare_search` accepts (a 3-D array, a list of 2-D arrays of
unequal lengths; no subset, one flat list, a list per query).

After every step the directory is read back (index_io.load_index_arrays) and the C oracle built from THOSE arrays must return the
same ids in the same order (exact ties aside) with scores within 1e-3: a stale index in the object after an update or a delete, a
wrong padding of ragged queries, a subset form mishandled all show here.  get_embeddings is checked against the oracle's rows.

usage: class_fuzz_worker.py <n_cases> <seed> [<first_case>]
"""
import os
import shutil
import sys
import tempfile
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import fast_plaid_amd as fp  # noqa: E402,F401
import plaid_oracle as OC  # noqa: E402
from fast_plaid_amd import search  # noqa: E402
from fast_plaid_amd.search import index_io  # noqa: E402
from parity import check_final  # noqa: E402


def oracle_of(path):
    a = index_io.load_index_arrays(path)
    return a, OC.OracleIndex(nbits=int(a["nbits"]), centroids=a["centroids"], bucket_weights=a["bucket_weights"], ivf=a["ivf"],
                             ivf_lengths=a["ivf_lengths"], doc_codes=a["doc_codes"], doc_residuals=a["doc_residuals"],
                             doc_lengths=a["doc_lengths"])


def run_case(seed, case):
    rng = np.random.default_rng([seed, case, 41])
    pick = lambda *v: v[int(rng.integers(0, len(v)))]   # noqa: E731
    dim, nbits = pick((128, 4), (128, 2), (64, 4), (96, 4), (48, 2))
    n0 = int(pick(3, 40, 250))
    # (at most as many centroids as create.rs's estimate of the list count can come to -- every document has at least one token --: the
    # regime in which the reference itself works; beyond it a probed cell can lie past the lists, where the reference's search errs and
    # this engine reads an empty list: INTEGRATION.md, deviations)
    C = min(int(pick(16, 64, 200)), int(2 ** np.floor(np.log2(16.0 * np.sqrt(float(n0))))))
    cent = rng.standard_normal((C, dim), dtype=np.float32)
    cent /= np.linalg.norm(cent, axis=1, keepdims=True)

    def mkdocs(n):
        out = []
        for _ in range(n):
            ln = int(rng.integers(1, 40))
            d = cent[rng.integers(0, C, ln)] + 0.3 * rng.standard_normal((ln, dim), dtype=np.float32) / np.sqrt(dim)
            out.append((d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float16))
        return out

    def queries(n_docs_now):
        B = int(rng.integers(1, 6))
        lens = [int(rng.integers(1, 40)) for _ in range(B)]
        qs = []
        for ln in lens:
            v = cent[rng.integers(0, C, ln)] + 0.3 * rng.standard_normal((ln, dim), dtype=np.float32) / np.sqrt(dim)
            qs.append((v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float16))
        form = int(pick(0, 1))
        if form == 0:   # one 3-D array (equal lengths)
            L = lens[0]
            qin = np.stack([np.resize(x, (L, dim)) for x in qs])   # (rows repeated cyclically up to the first query's length)
            q3 = qin
        else:           # a list of 2-D arrays: zero-padded to the longest (fast_plaid.py:772-780)
            qin = qs
            L = max(lens)
            q3 = np.stack([np.pad(x, ((0, L - x.shape[0]), (0, 0))) for x in qs])
        sform = int(pick(0, 0, 1, 2))
        if sform == 0:
            sub_in, subs = None, [None] * B
        elif sform == 1:
            one = sorted(set(rng.integers(0, n_docs_now, int(rng.integers(1, 30))).tolist()))
            sub_in, subs = one, [one] * B
        else:
            subs = [rng.integers(0, n_docs_now, int(rng.integers(1, 30))).tolist() for _ in range(B)]
            sub_in = subs
        return qin, q3, sub_in, subs

    def check(fpi, path, tag):
        arr, orc = oracle_of(path)
        n_docs_now = int(arr["doc_lengths"].shape[0])
        qin, q3, sub_in, subs = queries(n_docs_now)
        top_k, n_full, n_probe = int(pick(1, 5, 20)), int(pick(8, 64, 4096)), int(min(pick(1, 4, 8), arr["centroids"].shape[0]))
        out = fpi.search(qin, top_k=top_k, n_full_scores=n_full, n_ivf_probe=n_probe, show_progress=False, subset=sub_in)
        assert len(out) == q3.shape[0], f"{tag}: {len(out)} result rows for {q3.shape[0]} queries"
        for b, row in enumerate(out):
            ref = orc.search(q3[b: b + 1], top_k, n_full, n_probe, subset=None if subs[b] is None else [subs[b]])[0]
            check_final(np.array([p for p, _ in row], np.int64), np.array([s for _, s in row], np.float32), ref[0], ref[1], top_k)
        d = int(rng.integers(0, n_docs_now))
        emb = fpi.get_embeddings([d])[0]
        offs = np.concatenate([[0], np.cumsum(arr["doc_lengths"])])
        want = orc.decompress(arr["doc_codes"][offs[d]: offs[d + 1]], arr["doc_residuals"][offs[d]: offs[d + 1]]).astype(np.float32)
        assert emb.shape == want.shape and np.array_equal(emb, want, equal_nan=True), f"{tag}: get_embeddings({d}) differs from the oracle"
        return n_docs_now

    tmp = tempfile.mkdtemp()
    try:
        path = os.path.join(tmp, "ix")
        with search.FastPlaid(index=path, device="cuda:0") as fpi:
            own_kmeans = rng.random() < 0.25   # centroids from this repository's k-means (no oracle for those; whatever they are,
            first = mkdocs(n0)                 # the directory then defines the oracle)
            try:
                fpi.create(first, centroids=None if own_kmeans else cent, nbits=nbits)
            except ValueError as e:
                # create.rs:301-305: the held-out sample is round(0.05 x the sampled tokens) rows -- none for a corpus of under ten tokens
                assert "no heldout samples" in str(e) and sum(d.shape[0] for d in first) < 10, str(e)
                return
            n_now = check(fpi, path, "created")
            assert n_now == n0
            for op in range(int(rng.integers(1, 4))):
                if rng.random() < 0.6 or n_now <= 2:
                    new = mkdocs(int(rng.integers(1, 30)))
                    fpi.update(new, update_threshold_centroids=bool(rng.integers(0, 2)))
                    n_now += len(new)
                    tag = f"op{op}: update {len(new)}"
                else:
                    sub = sorted(set(rng.integers(0, n_now, int(rng.integers(1, max(2, n_now // 3)))).tolist()))
                    fpi.delete(sub)
                    n_now -= len(sub)
                    tag = f"op{op}: delete {len(sub)}"
                got = check(fpi, path, tag)
                assert got == n_now, f"{tag}: the directory holds {got} documents, expected {n_now}"
        with search.FastPlaid(index=path, device="cuda:0") as again:   # a fresh object on the directory
            check(again, path, "reopened")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    bad = 0
    for case in range(first, first + n):
        try:
            run_case(seed, case)
        except Exception as e:   # noqa: BLE001
            bad += 1
            print(f"FUZZ_CASE_FAILED case={case} seed={seed} {type(e).__name__}: {str(e)[:500]}", flush=True)
            if os.environ.get("FP_FUZZ_TRACEBACK"):
                traceback.print_exc()
    print(f"FUZZ_FAIL {bad}/{n}" if bad else f"FUZZ_OK {n}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
