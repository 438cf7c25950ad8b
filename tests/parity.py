"""Parity comparators shared by the GPU tests.

Bar (BASELINE.json north_star): identical top-k doc ids, MaxSim scores within 1e-3 fp32.

Since round 4 every stage up to the rerank list is compared BIT FOR BIT, unconditionally: the centroid scores S are exact (S1
certifies every fp32 MFMA result against the fp16 rounding boundaries and re-evaluates the flagged ones with the reference's
ascending chain, fp_internal.h FpS1Exact), and everything between S and the rerank list is integer work or fp16 max / fp32
ascending sums of those scores.  The one floating-point stage left with a tolerance is the exact MaxSim score (S7): its MFMA
accumulation order can move a returned score by one fp16 ulp of one column (<= 4.9e-4, inside the north star's 1e-3); the
ORDER of the returned documents is the oracle's because near-tied scores are re-evaluated in the reference's order
(k_final_mark / k_maxsim_repair).  The only other freedom is what the reference itself leaves undefined: the order among
EXACTLY tied scores (ATen's topk / sort tie order is implementation-defined; ours is id ascending).
"""
from __future__ import annotations

import numpy as np

SCORE_TOL = 1e-3          # north-star tolerance on final MaxSim scores


def ulp_diff_f16(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """distance in representable fp16 steps (monotone integer map)."""
    def mono(x):
        u = x.view(np.uint16).astype(np.int32)
        return np.where(u & 0x8000, 0x8000 - (u & 0x7FFF), 0x8000 + u)
    return np.abs(mono(a) - mono(b))


def check_centroid_scores(S_hip: np.ndarray, S_ref: np.ndarray):
    """S1: bit-identical fp16 scores (-0 and +0 are the same score)."""
    a = np.ascontiguousarray(S_hip).view(np.uint16).copy()
    b = np.ascontiguousarray(S_ref).view(np.uint16).copy()
    a[a == 0x8000] = 0
    b[b == 0x8000] = 0
    bad = a != b
    if bad.any():
        i = np.argwhere(bad)[0]
        raise AssertionError(f"{int(bad.sum())} of {bad.size} centroid scores differ from the reference's matmul; first at {tuple(i)}: "
                             f"{S_hip[tuple(i)]!r} vs {S_ref[tuple(i)]!r}")


def check_cells(cells_hip, cells_ref, S_ref, n_probe, exact_ties_ok=False):
    """S2: the same probed cells.  exact_ties_ok: a cell may differ only where its score EXACTLY equals the column's
    n_probe-th best score (an all-zero query token ties every centroid; the reference's choice among exact ties is
    implementation-defined)."""
    a, b = np.sort(np.asarray(cells_hip)), np.sort(np.asarray(cells_ref))
    if np.array_equal(a, b):
        return True
    assert exact_ties_ok, f"probed cells differ: {sorted(set(a.tolist()) ^ set(b.tolist()))}"
    Sf = np.asarray(S_ref, np.float32)
    k = min(n_probe, Sf.shape[0])
    kth = np.sort(Sf, axis=0)[::-1][k - 1]  # per column
    for c in set(a.tolist()) ^ set(b.tolist()):
        assert np.any(Sf[c] == kth), f"cell {c} differs and is not an exact tie at a column's probe cut"
    return False


def _same_order_modulo_exact_ties(ids_hip, ids_ref, scores_ref, ulp_ties=0):
    """identical sequences, except that documents whose REFERENCE scores are exactly equal may be permuted among themselves.
    ulp_ties = 1 (tests/fuzz_worker.py only, and only after the strict comparison has failed): neighbours whose reference scores
    differ by one fp32 ulp count as tied too.  The final score is an fp32 sum of the query columns' fp16 maxima (search.rs:401).
    That sum is exact -- hence order-independent -- unless it reaches 16 while some column maximum is below 2^-8, which takes a long
    query and documents of one or two tokens; then the last ulp depends on the ORDER of the additions: ascending in
    oracle/plaid_oracle.c, ATen's vectorised reduction (whose lane count follows the host's ISA) in the reference.  Round 6's
    fuzz met one such pair in ~45 000 cases (seed 11, case 1763: q_len 130, one-token documents): ATen, the C oracle and the
    device each rounded it their own way, one ulp apart."""
    ids_hip, ids_ref = list(ids_hip), list(ids_ref)
    assert len(ids_hip) == len(ids_ref), f"result count {len(ids_hip)} != {len(ids_ref)}"
    i = 0
    while i < len(ids_ref):
        j = i + 1
        while j < len(ids_ref) and (scores_ref[j] == scores_ref[i] or
                                    (ulp_ties and abs(float(scores_ref[j - 1]) - float(scores_ref[j])) <= ulp_ties * float(np.spacing(np.float32(abs(scores_ref[j - 1])))))):
            j += 1
        assert set(ids_hip[i:j]) == set(ids_ref[i:j]), f"positions {i}..{j - 1}: {ids_hip[i:j]} vs {ids_ref[i:j]} (reference scores {scores_ref[i:j]})"
        i = j


def check_final(pids_hip, scores_hip, pids_ref, scores_ref, top_k=None, cut_tie_ok=True, tol=SCORE_TOL, ulp_ties=0):
    """the returned ids ARE the reference's, in the reference's order (modulo its exact ties); scores within 1e-3,
    descending, no duplicates.  cut_tie_ok: when the reference's scores tie exactly ACROSS the top_k cut the tied group's
    members inside the list may differ (the reference's pick among them is implementation-defined)."""
    pids_hip, pids_ref = np.asarray(pids_hip), np.asarray(pids_ref)
    scores_hip, scores_ref = np.asarray(scores_hip, np.float32), np.asarray(scores_ref, np.float32)
    assert len(pids_hip) == len(pids_ref), f"result count {len(pids_hip)} != {len(pids_ref)}"
    assert len(set(pids_hip.tolist())) == len(pids_hip), "duplicate ids"
    assert np.all(np.diff(scores_hip) <= 0), "scores not in descending order"
    if len(pids_ref) == 0:
        return True
    assert np.abs(scores_hip - scores_ref).max() <= tol, f"scores differ by {np.abs(scores_hip - scores_ref).max()} position-wise"
    if np.array_equal(pids_hip, pids_ref):
        return True
    n = len(pids_ref)
    last = n
    if cut_tie_ok:   # the trailing group of exactly tied reference scores may extend beyond the list
        last = n - 1
        while last > 0 and scores_ref[last - 1] == scores_ref[n - 1]:
            last -= 1
    _same_order_modulo_exact_ties(pids_hip[:last], pids_ref[:last], scores_ref[:last], ulp_ties)
    return False


def check_trace(hip: dict, ref: dict, Q: int, n_probe: int, n_full: int, top_k: int, strict_cells=True, tol=SCORE_TOL, ulp_ties=0):
    """Stage-by-stage comparison of fp_search_trace against an oracle trace: S, cells, candidates, approximate scores and
    the rerank list bit for bit; exact scores within 1e-3; final ids identical.  strict_cells=False admits other cells only at
    EXACT ties of the probe cut (a zero query token) -- everything downstream is then compared only if the cells agree."""
    diag = {}
    if ref.get("S") is not None and hip.get("S") is not None:
        check_centroid_scores(hip["S"], ref["S"])
    cells_same = check_cells(hip["cells"], ref["cells"], ref.get("S"), n_probe, exact_ties_ok=not strict_cells)
    diag["cells_same"] = cells_same
    if not cells_same:
        return diag
    assert np.array_equal(hip["cand"], ref["cand"]), "candidate doc ids differ (integer stage)"
    assert np.array_equal(hip["approx"], ref["approx"]), \
        f"approximate scores differ (max |d| {np.abs(hip['approx'] - ref['approx']).max()}) although S is identical"
    assert np.all(np.diff(hip["rerank"]) > 0), "rerank ids not in ascending order"
    if not np.array_equal(hip["rerank"], ref["rerank"]):
        # the pruning cut (approx desc; ties: ours id ascending, the reference's implementation-defined): only EXACT ties of
        # the approximate score at the cut may differ
        amap = dict(zip(ref["cand"].tolist(), ref["approx"].tolist()))
        cut = min(amap[p] for p in ref["rerank"].tolist())
        for p in set(hip["rerank"].tolist()) ^ set(ref["rerank"].tolist()):
            assert amap[p] == cut, f"doc {p}: approx {amap[p]} is not an exact tie at the pruning cut {cut}"
        diag["rerank_same"] = False
        return diag
    diag["rerank_same"] = True
    d = np.abs(np.asarray(hip["exact"], np.float32) - np.asarray(ref["exact"], np.float32))
    assert d.max(initial=0.0) <= tol, f"exact MaxSim differs by {d.max()} (> {tol})"
    diag["exact_max_abs_diff"] = float(d.max(initial=0.0))
    diag["exact_equal_frac"] = float((d == 0).mean()) if d.size else 1.0
    diag["ids_identical"] = check_final(hip["pids"], hip["scores"], ref["pids"], ref["scores"], top_k, tol=tol, ulp_ties=ulp_ties)
    return diag
