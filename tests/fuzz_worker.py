"""GPU worker: randomized parity fuzz of fp_search / fp_search_trace against the C oracle.

Every case draws its own shape -- corpus size, document lengths (empty / one-token / repeated-code documents included), centroid
count off every tile grid, dim, nbits, batch, query length, n_ivf_probe, n_full_scores, top_k, subsets, zero-padded query rows,
unnormalised queries -- builds plain random index arrays (tests/test_hip_parity.py::_random_arrays: no corpus model) and checks

  1. fp_search_trace against the oracle's trace stage by stage (parity.check_trace: S, cells, candidates, approximate scores and
     the rerank list bit for bit; exact scores within 1e-3; the oracle's ids in its order),
  2. fp_search (the pruned / speculative / lazy path) == fp_search_trace bit for bit, on three calls in a row (the second runs on
     the learnt candidate capacity, the third may be a captured graph's replay),
  3. with a shared subset: fp_search_shared_subset == the per-query form,
  4. on a third of the cases each: pysearch_with_token_scores (same hits; the [q_len, doc_len] matrices == the oracle's bit for
     bit), fp_search_device (queries and results in HBM) == fp_search, reconstruct_embeddings of random documents == the
     oracle's decompressed rows bit for bit.

usage: fuzz_worker.py <n_cases> <seed> [<first_case>] [big]   (FP_APPROX_IMPL = q8 / l0 / l0h forces a form of S4 for the whole run)
"big": corpus-model indexes (fast-plaid_amd/synth.py, 5 k - 80 k documents, up to 2^16 centroids, batches up to 64) on which the
engine picks the bound stages / the lazy S1 / graph replay by itself: fp_search x 5 == fp_search_trace bit for bit for every query,
the oracle's ids in its order for the first four.
"stateful": see run_stateful (one index, <n_cases> calls of recurring shapes: what the engine remembers between calls).
"threads": run_threads (the same from four threads at once on one shared index, <n_cases> calls each).
"huge": run_huge (device-generated corpora of 0.1 - 1 M documents, 2^17 - 2^19 centroids).
"hostile": run_hostile (another thread of the process makes legacy-stream copies while fp_search captures its graphs).
prints one line per failing case (its number reproduces it: fuzz_worker.py 1 <seed> <case>) and FUZZ_OK <n> / FUZZ_FAIL <k>/<n>.
"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import fast_plaid_amd as fp  # noqa: E402
from parity import SCORE_TOL, check_trace  # noqa: E402
from test_hip_parity import _hip_index, _oracle, _random_arrays  # noqa: E402

R = fp.fast_plaid_rust
TALLY = {}   # what the engine ran, per fp_search call: form of S4 / S1 lazy (1) eager (0) replayed graph (-1)


def searched(*a):
    out = R.search_arrays(*a)
    k = (R.last_search_counts()["s4_form"], R.last_s1_counts()["lazy"])
    TALLY[k] = TALLY.get(k, 0) + 1
    return out


def draw_shape(rng):
    pick = lambda *v: v[int(rng.integers(0, len(v)))]   # noqa: E731
    n_docs = int(pick(1, 3, 40, 300, 900, 2500, 6000))
    n_docs = max(1, int(n_docs * rng.uniform(0.5, 1.5)))
    max_len = int(pick(1, 2, 8, 33, 70, 130, 300))
    C = int(pick(8, 37, 100, 129, 257, 513, 1000, 2048, 3001, 5000))
    dim = int(pick(128, 128, 128, 64, 64, 96, 48, 40, 256))
    nbits = int(pick(4, 4, 2, 2, 1, 8))
    B = int(rng.integers(1, 10))
    Q = int(pick(1, 2, 3, 7, 17, 31, 32, 33, 50, 64, 65, 100, 128, 130))
    n_probe = int(pick(1, 2, 4, 8, 8, 16, 32, 33, 40, 64, 100))
    n_probe = min(n_probe, C)
    n_full = int(pick(1, 4, 8, 40, 64, 128, 256, 1000, 2000, 4096))
    top_k = int(pick(1, 3, 10, 20, 100, 1000, 5000))
    subset = int(pick(0, 0, 0, 1, 2))    # 0 none, 1 per-query lists, 2 one shared list
    qkind = int(pick(0, 0, 0, 1, 2, 3))  # 0 normalised, 1 zero-padded tails, 2 scaled x3, 3 scaled x0.01
    return n_docs, max_len, C, dim, nbits, B, Q, n_probe, n_full, top_k, subset, qkind


def run_case(seed, case):
    rng = np.random.default_rng([seed, case])
    shape = draw_shape(rng)
    n_docs, max_len, C, dim, nbits, B, Q, n_probe, n_full, top_k, subset, qkind = shape
    forced = os.environ.get("FP_APPROX_IMPL")
    if forced and Q > 128:
        Q = 128   # (the forced forms of S4 serve up to 128 query columns; the engine never picks them beyond)
    arr = _random_arrays(rng, n_docs, max_len, C, dim, nbits, empty_frac=float(rng.choice([0.0, 0.1, 0.5])))
    pick = rng.integers(0, C, (B, Q))
    q = arr["centroids"][pick].astype(np.float32) + 0.3 * rng.standard_normal((B, Q, dim), dtype=np.float32) / np.sqrt(dim)
    q /= np.linalg.norm(q, axis=2, keepdims=True)
    zero_rows = np.zeros(B, bool)
    if qkind == 1 and Q > 1:
        for b in range(B):
            if rng.random() < 0.6:
                q[b, int(rng.integers(1, Q)):] = 0
                zero_rows[b] = True
    elif qkind == 2:
        q *= 3.0
    elif qkind == 3:
        q *= 0.01
    q = q.astype(np.float16)
    subs = None
    if subset == 1:
        subs = [rng.integers(0, n_docs, int(rng.integers(1, 60))).tolist() for _ in range(B)]
    elif subset == 2:
        one = rng.integers(0, n_docs, int(rng.integers(1, 200))).tolist()
        subs = [one] * B
    hip = _hip_index(fp, arr)
    orc = _oracle(arr)
    params = R.SearchParameters(2000, n_full, top_k, n_probe)
    runs = [searched(hip, q, params, subs) for _ in range(3)]
    if subset == 2:   # the shared list given as distinct (equal) lists takes the per-query entry point
        runs.append(searched(hip, q, params, [list(one) for _ in range(B)]))
    for b in range(B):
        sub = None if subs is None else subs[b]
        h = R.search_trace(hip, q[b], params, sub)
        o = orc.search_trace(q[b], top_k, n_full, n_probe, sub)
        # (a zero query token ties every centroid at the probe cut: the reference's pick is implementation-defined)
        # (the north star's 1e-3 is quoted on normalised queries: the one freedom left, one fp16 ulp of a column's maximum, scales with
        # the query; the ORDER must be the oracle's at any scale)
        # The trace's "exact" array holds the MFMA-order score of EVERY rerank document, before the order repair: a document outside
        # the emitted top_k keeps its flagged columns' ulps.  x3 queries: column values of [0.5, 1) move to [1.5, 3), and the fp16 ulp of
        # [2, 4) is FOUR times that of [0.5, 1): two flagged columns cost 2 x 1.95e-3 there.  The RETURNED scores are another
        # matter: k_final_mark re-scores every emitted document whose budget exceeds 0.00095, at any query scale (checked below).
        tol = SCORE_TOL * (4.0 if qkind == 2 else 1.0)
        try:
            check_trace(h, o, Q, n_probe, n_full, top_k, strict_cells=not zero_rows[b], tol=tol)
        except AssertionError as e:
            if "positions" not in str(e):
                raise
            # neighbours one fp32 ulp apart in the oracle: an inexact fp32 sum over the columns (parity._same_order_modulo_exact_ties)
            check_trace(h, o, Q, n_probe, n_full, top_k, strict_cells=not zero_rows[b], tol=tol, ulp_ties=1)
            TALLY["order settled by the last ulp of an inexact fp32 sum"] = TALLY.get("order settled by the last ulp of an inexact fp32 sum", 0) + 1
        if len(h["pids"]) and len(h["pids"]) == len(o["pids"]):   # returned scores: within the north star's 1e-3, whatever the query's scale
            om = dict(zip(np.asarray(o["pids"]).tolist(), np.asarray(o["scores"], np.float32).tolist()))
            worst = max(abs(float(sc) - om[p]) for p, sc in zip(np.asarray(h["pids"]).tolist(), np.asarray(h["scores"], np.float32).tolist()) if p in om)
            assert worst <= SCORE_TOL, f"query {b}: a returned score is {worst} off the oracle's (> 1e-3)"
        for i, (pids, scores, counts) in enumerate(runs):
            assert counts[b] == len(h["pids"]), f"call {i} query {b}: count {counts[b]} != trace {len(h['pids'])}"
            assert np.array_equal(pids[b, : counts[b]], h["pids"]), f"call {i} query {b}: ids differ from the trace"
            assert np.array_equal(scores[b, : counts[b]], h["scores"]), f"call {i} query {b}: scores differ from the trace"
            assert np.all(pids[b, counts[b]:] == -1), f"call {i} query {b}: unused slots not -1"
    extra = rng.random(4)
    if extra[3] < 0.2:   # fp_index_export: the arrays come back as they went in (original token order inside every document)
        back = R.export_index_arrays(hip)
        T = int(arr["doc_lengths"].sum())
        assert np.array_equal(back["doc_lengths"], arr["doc_lengths"]), "export: doc_lengths differ"
        assert np.array_equal(back["doc_codes"][:T], arr["doc_codes"][:T]), "export: doc_codes differ"
        assert np.array_equal(back["doc_residuals"][:T], arr["doc_residuals"][:T]), "export: doc_residuals differ"
        assert np.array_equal(back["ivf_lengths"][: arr["ivf_lengths"].shape[0]], arr["ivf_lengths"]) and np.array_equal(back["ivf"], arr["ivf"]), "export: IVF differs"
        TALLY["export"] = TALLY.get("export", 0) + 1
    if extra[2] < 0.3:   # reconstruct_embeddings (embeddings.rs:12-69): whole documents decompressed == the oracle's rows, bit for bit
        offs = np.concatenate([[0], np.cumsum(arr["doc_lengths"])])
        docs = rng.integers(0, n_docs, min(n_docs, 6)).tolist()
        got = R.reconstruct_embeddings(hip, docs)
        for d, g in zip(docs, got):
            want = orc.decompress(arr["doc_codes"][offs[d]: offs[d + 1]], arr["doc_residuals"][offs[d]: offs[d + 1]]).astype(np.float32)
            assert g.shape == want.shape and np.array_equal(g, want, equal_nan=True), f"reconstruct_embeddings: document {d} differs from the oracle"
        TALLY["reconstruct"] = TALLY.get("reconstruct", 0) + 1
    if extra[0] < 0.35:   # pysearch_with_token_scores: the same hits, every [q_len, doc_len] matrix == the oracle's bit for bit
        res = R.pysearch_with_token_scores(hip, "cuda:0", q, params, False, subs)
        pids, scores, counts = runs[0]
        for b, r in enumerate(res):
            assert r.passage_ids == pids[b, : counts[b]].tolist() and np.array_equal(np.asarray(r.scores, np.float32), scores[b, : counts[b]]), \
                f"token-score search, query {b}: hits differ from fp_search"
            for pid, m in list(zip(r.passage_ids, r.token_scores))[:8]:
                ref = orc.token_scores(q[b], pid)
                assert m.shape == ref.shape == (Q, int(arr["doc_lengths"][pid])), f"query {b} doc {pid}: matrix shape {m.shape}"
                assert np.array_equal(m.view(np.uint16), ref.view(np.uint16)), f"query {b} doc {pid}: token scores differ from the oracle"
        TALLY["token_scores"] = TALLY.get("token_scores", 0) + 1
    if extra[1] < 0.35 and subs is None:   # fp_search_device: queries and results in HBM
        k = max(top_k, 1)
        dq = R.DeviceBuffer(0, q.nbytes).upload(q)
        dp, ds, dc = R.DeviceBuffer(0, B * k * 8), R.DeviceBuffer(0, B * k * 4), R.DeviceBuffer(0, B * 4)
        dc.upload(np.full(B, 77, np.int32))
        R.search_device(hip, dq, B, Q, params, dp, ds, dc)
        pids, scores, counts = runs[0]
        gc, gp, gs = dc.download(np.int32, (B,)), dp.download(np.int64, (B, k)), ds.download(np.float32, (B, k))
        assert np.array_equal(gc, counts), f"device-resident search: counts {gc.tolist()} vs {counts.tolist()}"
        for b in range(B):
            assert np.array_equal(gp[b, : gc[b]], pids[b, : counts[b]]) and np.array_equal(gs[b, : gc[b]], scores[b, : counts[b]]), \
                f"device-resident search, query {b}: differs from the host-buffer search"
        TALLY["search_device"] = TALLY.get("search_device", 0) + 1
    return shape


def run_big_case(seed, case):
    rng = np.random.default_rng([seed, case, 77])
    pick = lambda *v: v[int(rng.integers(0, len(v)))]   # noqa: E731
    n_docs = int(pick(5000, 12000, 30000, 80000) * rng.uniform(0.7, 1.3))
    doc_len = int(pick(16, 48, 48, 100, 128))
    C = int(pick(1024, 2048, 8192, 16384, 65536))
    dim, nbits = pick((128, 4), (128, 4), (128, 2), (64, 4), (64, 2))
    B = int(pick(1, 3, 8, 20, 64))
    Q = int(pick(8, 20, 32, 32, 32, 33, 48, 64))
    n_probe = int(pick(1, 4, 8, 8, 16, 32))
    n_full = int(pick(32, 64, 256, 1024, 4096))
    top_k = int(pick(5, 25, 100, 1000))
    qkind = int(pick(0, 0, 0, 1))
    spec = fp.synth.SynthSpec(n_docs=n_docs, doc_len=doc_len, n_centroids=C, dim=dim, nbits=nbits, variable_len=bool(rng.integers(0, 2)),
                              seed=int(rng.integers(1, 1 << 30)))
    arr = fp.synth.host_index_arrays(spec)
    q = fp.synth.make_queries(spec, arr["centroids"], B, Q, seed=int(rng.integers(1, 1 << 30)))
    zero_rows = np.zeros(B, bool)
    if qkind == 1:
        for b in range(B):
            if rng.random() < 0.5:
                q[b, int(rng.integers(1, Q)):] = 0
                zero_rows[b] = True
    hip = _hip_index(fp, arr)
    params = R.SearchParameters(2000, n_full, top_k, n_probe)
    runs = [searched(hip, q, params) for _ in range(5)]   # (plain, learnt capacity x 2, the capturing call, a replay)
    orc = _oracle(arr)
    for b in range(B):
        h = R.search_trace(hip, q[b], params, None)
        for i, (pids, scores, counts) in enumerate(runs):
            assert counts[b] == len(h["pids"]), f"call {i} query {b}: count {counts[b]} != trace {len(h['pids'])}"
            assert np.array_equal(pids[b, : counts[b]], h["pids"]), f"call {i} query {b}: ids differ from the trace"
            assert np.array_equal(scores[b, : counts[b]], h["scores"]), f"call {i} query {b}: scores differ from the trace"
        if b < 4:
            o = orc.search_trace(q[b], top_k, n_full, n_probe, None)
            check_trace(h, o, Q, n_probe, n_full, top_k, strict_cells=not zero_rows[b])
    return (n_docs, doc_len, C, dim, nbits, B, Q, n_probe, n_full, top_k, qkind)


def run_stateful(seed, n_calls):
    """ONE index, a long sequence of calls whose shapes come from a small pool (so that shapes repeat: learnt capacities, captured
    graphs, the 8-entry shape cache and its evictions, the probe fallback's and the lazy form's switches all carry over from call
    to call), a different query batch every call, now and then one with zero-padded rows or with every query the same: every
    call's result == fp_search_trace of each of its queries, bit for bit."""
    rng = np.random.default_rng([seed, 4242])
    pick = lambda *v: v[int(rng.integers(0, len(v)))]   # noqa: E731
    spec = fp.synth.SynthSpec(n_docs=int(pick(12000, 30000, 60000)), doc_len=int(pick(32, 48, 100)), n_centroids=int(pick(2048, 8192, 32768)),
                              variable_len=True, seed=int(rng.integers(1, 1 << 30)))
    arr = fp.synth.host_index_arrays(spec)
    hip = _hip_index(fp, arr)
    pool = []
    for _ in range(12):
        pool.append((int(pick(1, 3, 8, 16)), int(pick(8, 32, 32, 48)), int(pick(1, 4, 8, 16, 40)), int(pick(64, 1024, 4096)), int(pick(10, 100, 1000))))
    bad = 0
    for call in range(n_calls):
        B, Q, n_probe, n_full, top_k = pool[int(rng.integers(0, len(pool)))] if rng.random() < 0.9 else \
            (int(pick(1, 5, 12)), int(pick(16, 32, 64)), int(pick(2, 8, 32)), int(pick(128, 2048)), int(pick(5, 50)))
        q = fp.synth.make_queries(spec, arr["centroids"], B, Q, seed=int(rng.integers(1, 1 << 30)))
        kind = rng.random()
        if kind < 0.08 and Q > 1:
            q[:, int(rng.integers(1, Q)):] = 0
        elif kind < 0.12:
            q[:] = q[0]
        params = R.SearchParameters(2000, n_full, top_k, n_probe)
        try:
            pids, scores, counts = searched(hip, q, params)
            for b in range(B):
                h = R.search_trace(hip, q[b], params, None)
                assert counts[b] == len(h["pids"]), f"query {b}: count {counts[b]} != trace {len(h['pids'])}"
                assert np.array_equal(pids[b, : counts[b]], h["pids"]), f"query {b}: ids differ from the trace"
                assert np.array_equal(scores[b, : counts[b]], h["scores"]), f"query {b}: scores differ from the trace"
        except Exception as e:   # noqa: BLE001
            bad += 1
            print(f"FUZZ_CASE_FAILED stateful seed={seed} call={call} shape={(B, Q, n_probe, n_full, top_k)} kind={kind:.3f} {type(e).__name__}: {str(e)[:300]}",
                  flush=True)
    print("calls by (S4 form, S1 lazy):", sorted(TALLY.items(), key=str), "graph replays", R.graph_replay_count() if hasattr(R, "graph_replay_count") else "?")
    print(f"FUZZ_FAIL {bad}/{n_calls}" if bad else f"FUZZ_OK {n_calls}", flush=True)
    sys.exit(1 if bad else 0)


def run_threads(seed, n_calls, n_threads=4):
    """run_stateful's sequence from several threads at once on ONE shared index (load.rs:58-59 Send + Sync; the reference's
    joblib / thread-per-device callers): every thread draws its own shapes and batches and checks each of its calls against
    fp_search_trace -- the scratch pool hands every concurrent call its own stream, capacities, graphs and switches.  A fifth
    thread builds, searches and drops small indexes of its own meanwhile."""
    import threading
    rng0 = np.random.default_rng([seed, 777])
    spec = fp.synth.SynthSpec(n_docs=30000, doc_len=48, n_centroids=8192, variable_len=True, seed=int(rng0.integers(1, 1 << 30)))
    arr = fp.synth.host_index_arrays(spec)
    hip = _hip_index(fp, arr)
    pool = [(int(rng0.choice([1, 3, 8, 16])), int(rng0.choice([8, 32, 48])), int(rng0.choice([1, 4, 8, 40])), int(rng0.choice([64, 1024, 4096])),
             int(rng0.choice([10, 100]))) for _ in range(6)]
    errors = []

    def work(t):
        rng = np.random.default_rng([seed, 777, t])
        for call in range(n_calls):
            B, Q, n_probe, n_full, top_k = pool[int(rng.integers(0, len(pool)))]
            q = fp.synth.make_queries(spec, arr["centroids"], B, Q, seed=int(rng.integers(1, 1 << 30)))
            params = R.SearchParameters(2000, n_full, top_k, n_probe)
            try:
                pids, scores, counts = R.search_arrays(hip, q, params)
                for b in range(B):
                    h = R.search_trace(hip, q[b], params, None)
                    assert counts[b] == len(h["pids"]) and np.array_equal(pids[b, : counts[b]], h["pids"]) and \
                        np.array_equal(scores[b, : counts[b]], h["scores"]), f"query {b} differs from the trace"
            except Exception as e:   # noqa: BLE001
                errors.append(f"thread {t} call {call} shape {(B, Q, n_probe, n_full, top_k)}: {type(e).__name__}: {str(e)[:300]}")

    def build(t):   # index construction and destruction (allocations, uploads, the layout kernels) next to the searching threads
        rng = np.random.default_rng([seed, 778, t])
        for i in range(max(4, n_calls // 8)):
            try:
                a = _random_arrays(rng, int(rng.integers(50, 1500)), 30, int(rng.choice([64, 257, 1000])), 128, 4)
                small = _hip_index(fp, a)
                qs = a["centroids"][rng.integers(0, a["centroids"].shape[0], (2, 16))]
                pr = R.SearchParameters(2000, 64, 5, 4)
                got = R.search_arrays(small, qs, pr)
                for b in range(2):
                    h = R.search_trace(small, qs[b], pr, None)
                    assert got[2][b] == len(h["pids"]) and np.array_equal(got[0][b, : got[2][b]], h["pids"]), "fresh index: result differs from the trace"
                del small
            except Exception as e:   # noqa: BLE001
                errors.append(f"builder {t} index {i}: {type(e).__name__}: {str(e)[:300]}")

    ths = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)] + [threading.Thread(target=build, args=(0,))]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    for e in errors[:20]:
        print("FUZZ_CASE_FAILED threads", e, flush=True)
    print(f"FUZZ_FAIL {len(errors)}/{n_calls * n_threads}" if errors else f"FUZZ_OK {n_calls}", flush=True)
    sys.exit(1 if errors else 0)


def run_hostile(seed, n_calls):
    """the application's OTHER threads: one thread keeps making synchronous legacy-stream copies through the HIP runtime itself
    (what a framework's .cpu() / .item() does) while this thread searches with recurring shapes, i.e. while fp_search captures
    graphs.  The runtime refuses such a copy while a capture is open and invalidates the capture with it: fp_search must notice,
    run the batch on the plain path and return the right result -- never an error."""
    import ctypes
    import threading
    hip = ctypes.CDLL("libamdhip64.so")
    dev = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dev), ctypes.c_size_t(64)) == 0
    stop, refused = [False], [0]

    def disturb():
        host = (ctypes.c_char * 64)()
        while not stop[0]:
            if hip.hipMemcpy(host, dev, ctypes.c_size_t(64), ctypes.c_int(2)) != 0:   # hipMemcpyDeviceToHost on the legacy stream
                refused[0] += 1
                hip.hipGetLastError()

    rng = np.random.default_rng([seed, 999])
    spec = fp.synth.SynthSpec(n_docs=20000, doc_len=48, n_centroids=4096, variable_len=True, seed=int(rng.integers(1, 1 << 30)))
    arr = fp.synth.host_index_arrays(spec)
    ix = _hip_index(fp, arr)
    pool = [(int(rng.choice([1, 4, 16])), int(rng.choice([8, 32])), int(rng.choice([4, 8])), int(rng.choice([256, 4096])), 10) for _ in range(16)]
    th = threading.Thread(target=disturb)
    th.start()
    bad, done = 0, []
    try:
        for call in range(n_calls):
            B, Q, n_probe, n_full, top_k = pool[(call // 5) % len(pool)]   # five calls in a row per shape: plain, learnt, learnt, capture, replay
            q = fp.synth.make_queries(spec, arr["centroids"], B, Q, seed=int(rng.integers(1, 1 << 30)))
            params = R.SearchParameters(2000, n_full, top_k, n_probe)
            try:
                done.append((q, params, searched(ix, q, params)))
            except Exception as e:   # noqa: BLE001
                bad += 1
                print(f"FUZZ_CASE_FAILED hostile call={call} shape={(B, Q, n_probe, n_full, top_k)} {type(e).__name__}: {str(e)[:300]}", flush=True)
    finally:
        stop[0] = True
        th.join()
    for call, (q, params, (pids, scores, counts)) in enumerate(done):   # every result against the trace, once the other thread is gone
        for b in range(q.shape[0]):
            h = R.search_trace(ix, q[b], params, None)
            if not (counts[b] == len(h["pids"]) and np.array_equal(pids[b, : counts[b]], h["pids"]) and np.array_equal(scores[b, : counts[b]], h["scores"])):
                bad += 1
                print(f"FUZZ_CASE_FAILED hostile call={call} query {b}: result differs from the trace", flush=True)
                break
    print("legacy-stream copies the runtime refused:", refused[0], " calls by (S4 form, S1 lazy):", sorted(TALLY.items(), key=str), " graph replays", R.graph_replay_count())
    print(f"FUZZ_FAIL {bad}/{n_calls}" if bad else f"FUZZ_OK {n_calls}", flush=True)
    sys.exit(1 if bad else 0)


def run_huge(seed, n_cases):
    """device-generated corpora at the sizes the multi-range forms of S4 exist for (2^17 - 2^19 centroids, 0.1 - 1 M documents; no host
    arrays): fp_search x 4 (plain, learnt capacity, capture, replay) == fp_search_trace of each query, bit for bit."""
    bad = 0
    for case in range(n_cases):
        rng = np.random.default_rng([seed, case, 31])
        pick = lambda *v: v[int(rng.integers(0, len(v)))]   # noqa: E731
        C = int(pick(131072, 262144, 524288))
        n_docs = int(pick(100000, 300000, 1000000))
        doc_len = int(pick(32, 64, 128)) if n_docs < 1000000 else int(pick(32, 64))
        B, Q = int(pick(2, 8, 24, 64)), int(pick(16, 32, 32, 48))
        n_probe, n_full, top_k = int(pick(2, 8, 16)), int(pick(256, 4096, 16384)), int(pick(10, 100, 1000))
        shape = (C, n_docs, doc_len, B, Q, n_probe, n_full, top_k)
        try:
            spec = fp.synth.SynthSpec(n_docs=n_docs, doc_len=doc_len, n_centroids=C, variable_len=bool(rng.integers(0, 2)), seed=int(rng.integers(1, 1 << 30)))
            idx = R.construct_synthetic_index(spec, "cuda:0")
            q = fp.synth.make_queries(spec, fp.synth.centroids(spec), B, Q, seed=int(rng.integers(1, 1 << 30)))
            params = R.SearchParameters(2000, n_full, top_k, n_probe)
            runs = [searched(idx, q, params) for _ in range(4)]
            for b in range(min(B, 12)):
                h = R.search_trace(idx, q[b], params, None)
                for i, (pids, scores, counts) in enumerate(runs):
                    assert counts[b] == len(h["pids"]) and np.array_equal(pids[b, : counts[b]], h["pids"]) and \
                        np.array_equal(scores[b, : counts[b]], h["scores"]), f"call {i} query {b} differs from the trace"
            del idx
        except Exception as e:   # noqa: BLE001
            bad += 1
            print(f"FUZZ_CASE_FAILED huge case={case} seed={seed} shape={shape} {type(e).__name__}: {str(e)[:300]}", flush=True)
    print("calls by (S4 form, S1 lazy):", sorted(TALLY.items(), key=str))
    print(f"FUZZ_FAIL {bad}/{n_cases}" if bad else f"FUZZ_OK {n_cases}", flush=True)
    sys.exit(1 if bad else 0)


def main():
    if len(sys.argv) > 4 and sys.argv[4] == "huge":
        run_huge(int(sys.argv[2]), int(sys.argv[1]))
    if len(sys.argv) > 4 and sys.argv[4] == "hostile":
        run_hostile(int(sys.argv[2]), int(sys.argv[1]))
    if len(sys.argv) > 4 and sys.argv[4] == "threads":
        run_threads(int(sys.argv[2]), int(sys.argv[1]))
    if len(sys.argv) > 4 and sys.argv[4] == "stateful":
        run_stateful(int(sys.argv[2]), int(sys.argv[1]))
    if len(sys.argv) > 4 and sys.argv[4] == "big":
        n, seed, first, bad = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 0
        for case in range(first, first + n):
            try:
                run_big_case(seed, case)
            except Exception as e:   # noqa: BLE001
                bad += 1
                print(f"FUZZ_CASE_FAILED big case={case} seed={seed} {type(e).__name__}: {str(e)[:400]}", flush=True)
                if os.environ.get("FP_FUZZ_TRACEBACK"):
                    traceback.print_exc()
        print("calls by (S4 form, S1 lazy):", sorted(TALLY.items(), key=str))
        print(f"FUZZ_FAIL {bad}/{n}" if bad else f"FUZZ_OK {n}", flush=True)
        sys.exit(1 if bad else 0)
    n = int(sys.argv[1])
    seed = int(sys.argv[2])
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    bad = 0
    for case in range(first, first + n):
        try:
            run_case(seed, case)
        except Exception as e:   # noqa: BLE001  (every failure is reported with its case number; the run goes on)
            bad += 1
            rng = np.random.default_rng([seed, case])
            print(f"FUZZ_CASE_FAILED case={case} seed={seed} shape={draw_shape(rng)} {type(e).__name__}: {str(e)[:400]}", flush=True)
            if os.environ.get("FP_FUZZ_TRACEBACK"):
                traceback.print_exc()
    print("calls by (S4 form, S1 lazy):", sorted(TALLY.items(), key=str))
    print(f"FUZZ_FAIL {bad}/{n}" if bad else f"FUZZ_OK {n}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
