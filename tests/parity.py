"""Parity comparators shared by the GPU tests.

Bar (BASELINE.json north_star): identical top-k doc ids, MaxSim scores within 1e-3 fp32.
Integer / index stages are compared bit-exactly whenever their float inputs are identical.
The only admissible float difference is the fp32 ACCUMULATION ORDER inside the two matmuls
(MFMA vs the CPU's ascending-k chain), which can move a stored fp16 value by one ulp in rare
cases; every relaxation below is tied to that mechanism and is bounded, never a blanket skip.
"""
from __future__ import annotations

import numpy as np

SCORE_TOL = 1e-3          # north-star tolerance on final MaxSim scores
FP16_ULP_AT_1 = 2.0 ** -10  # spacing of fp16 in [1, 2)


def ulp_diff_f16(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """distance in representable fp16 steps (monotone integer map)."""
    def mono(x):
        u = x.view(np.uint16).astype(np.int32)
        return np.where(u & 0x8000, 0x8000 - (u & 0x7FFF), 0x8000 + u)
    return np.abs(mono(a) - mono(b))


def check_centroid_scores(S_hip: np.ndarray, S_ref: np.ndarray):
    d = ulp_diff_f16(S_hip, S_ref)
    assert d.max() <= 1, f"centroid score differs by {d.max()} fp16 ulps"
    frac = float((d > 0).mean())
    assert frac < 5e-3, f"{frac:.4%} of centroid scores differ (expected ~0.05% from accumulation order)"
    return frac


def cells_explainable(cells_hip, cells_ref, S_ref, n_probe) -> bool:
    """A probed-cell difference is admissible only when the cell's score is within one fp16
    ulp of the column's n_probe-th best score in the oracle (a rounding-order near-tie)."""
    a, b = set(cells_hip.tolist()), set(cells_ref.tolist())
    if a == b:
        return True
    Sf = S_ref.astype(np.float32)
    k = min(n_probe, Sf.shape[0])
    kth = np.sort(Sf, axis=0)[::-1][k - 1]  # per column
    for c in (a ^ b):
        gap = np.abs(Sf[c] - kth)
        tol = np.maximum(np.abs(kth), 2.0 ** -14) * 2.0 ** -10 * 2
        if not np.any(gap <= tol):
            return False
    return True


def check_final(pids_hip, scores_hip, pids_ref, scores_ref, top_k, exact_ref_by_pid=None, excused=frozenset(),
                cut_slack=SCORE_TOL):
    """ids identical except for near-ties at the top_k cut; matched scores within 1e-3;
    descending order; no duplicates.  `excused` = docs that only one side exact-scored because
    of an (already verified) approximate-score near-tie at the pruning cut: they may appear on
    one side only, and each of them may displace one doc at the bottom of the other list."""
    pids_hip, pids_ref = np.asarray(pids_hip), np.asarray(pids_ref)
    scores_hip, scores_ref = np.asarray(scores_hip, np.float32), np.asarray(scores_ref, np.float32)
    assert len(pids_hip) == len(pids_ref), f"result count {len(pids_hip)} != {len(pids_ref)}"
    assert len(set(pids_hip.tolist())) == len(pids_hip), "duplicate ids"
    assert np.all(np.diff(scores_hip) <= 0), "scores not in descending order"
    ref = dict(zip(pids_ref.tolist(), scores_ref.tolist()))
    hip = dict(zip(pids_hip.tolist(), scores_hip.tolist()))
    for p in set(ref) & set(hip):
        assert abs(ref[p] - hip[p]) <= SCORE_TOL, f"doc {p}: score {hip[p]} vs {ref[p]}"
    only_h, only_r = set(hip) - set(ref), set(ref) - set(hip)
    if only_h or only_r:
        n_exc = len((only_h | only_r) & set(excused))
        k = len(pids_ref)
        # the cut may legitimately sit up to n_exc positions higher
        kth_r = float(scores_ref[max(k - 1 - n_exc, 0)])
        kth_h = float(scores_hip[max(k - 1 - n_exc, 0)])
        for p in only_r - set(excused):
            assert ref[p] - kth_r <= 2 * cut_slack, f"doc {p} (score {ref[p]}) missing and not a near-tie at the cut {kth_r}"
        for p in only_h - set(excused):
            assert hip[p] - kth_h <= 2 * cut_slack, f"doc {p} (score {hip[p]}) extra and not a near-tie at the cut {kth_h}"
            if exact_ref_by_pid is not None and p in exact_ref_by_pid:
                assert abs(exact_ref_by_pid[p] - hip[p]) <= SCORE_TOL
    return len(only_h)


def check_trace(hip: dict, ref: dict, Q: int, n_probe: int, n_full: int, top_k: int, strict_cells=True):
    """Stage-by-stage comparison of fp_search_trace against an oracle trace.  Returns a dict
    of diagnostics.  `ref` may come from the C oracle or from a golden fixture."""
    diag = {}
    excused = set()
    S_same = False
    if ref.get("S") is not None and hip.get("S") is not None:
        diag["S_mismatch_frac"] = check_centroid_scores(hip["S"], ref["S"])
        S_same = diag["S_mismatch_frac"] == 0.0
    cells_same = set(hip["cells"].tolist()) == set(ref["cells"].tolist())
    if not cells_same:
        assert not S_same or not strict_cells, "probed cells differ although centroid scores are identical"
        if ref.get("S") is not None:
            assert cells_explainable(hip["cells"], ref["cells"], ref["S"], n_probe), "probed cells differ beyond near-ties"
    else:
        assert np.array_equal(np.sort(hip["cells"]), np.sort(ref["cells"]))
    diag["cells_same"] = cells_same
    if cells_same:
        # S3 is pure integer work: must be bit-exact
        assert np.array_equal(hip["cand"], ref["cand"]), "candidate doc ids differ (integer stage)"
        # S4: fp16 max over identical integer gathers; differs only where S differs by an ulp
        d = np.abs(hip["approx"] - ref["approx"])
        lim = Q * FP16_ULP_AT_1
        assert d.max(initial=0.0) <= lim, f"approx score differs by {d.max()}"
        if S_same:
            assert np.array_equal(hip["approx"], ref["approx"]), "approx scores differ although S is identical"
        diag["approx_mismatch_frac"] = float((d > 0).mean()) if d.size else 0.0
        # S5: rerank set equal except docs within the approx perturbation of the cut
        a, b = set(hip["rerank"].tolist()), set(ref["rerank"].tolist())
        excused = a ^ b
        if a != b:
            assert not S_same, "rerank set differs although S is identical"
            amap = dict(zip(ref["cand"].tolist(), ref["approx"].tolist()))
            cut = min(amap[p] for p in b)
            for p in (a ^ b):
                assert abs(amap[p] - cut) <= 2 * lim, f"doc {p}: approx {amap[p]} is not near the cut {cut}"
        diag["rerank_same"] = a == b
    # S7: exact scores of the docs both sides scored
    rmap = dict(zip(ref["rerank"].tolist(), ref["exact"].tolist()))
    hmap = dict(zip(hip["rerank"].tolist(), hip["exact"].tolist()))
    common = sorted(set(rmap) & set(hmap))
    if common:
        dd = np.array([abs(rmap[p] - hmap[p]) for p in common])
        assert dd.max() <= SCORE_TOL, f"exact MaxSim differs by {dd.max()} (> 1e-3)"
        diag["exact_max_abs_diff"] = float(dd.max())
        diag["exact_equal_frac"] = float((dd == 0).mean())
    assert np.all(np.diff(hip["rerank"]) > 0), "rerank ids not in ascending order"
    if not cells_same:  # candidate sets may differ: excuse docs only one side considered
        excused = set(hip["rerank"].tolist()) ^ set(ref["rerank"].tolist())
    diag["final_swaps"] = check_final(hip["pids"], hip["scores"], ref["pids"], ref["scores"], top_k, rmap, excused)
    return diag
