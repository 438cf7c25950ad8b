"""CPU model of the lazy S1's SELECTION argument ("S1 lazy form"; fast-plaid_amd/csrc/fp_kernels.hip:
k_probe_tau's lz_tight, lz_delta, k_sel_gather_lz, k_lz_exact, k_sel_finish_lz) -- no GPU needed.

The device stores upper candidates s = h(x + u) of the centroid scores; the reference's value is t = h(chain) with the chain
result inside [x - u, x + u].  The approximate score summed over stored column maxima, A_up, is then an upper bound of the
reference's A, and the selection claims:  with U = the R-th largest A_up and D = lz_delta(query slack, U),
    A_up > U + D  => the document is in the reference's top R        ("certain")
    A_up < U - D  => it is not
and the reference's top R = the certain ones + the best of the rest ("maybes") by exact score (desc, id asc).
This test builds t, x, s for a small random corpus with numpy (x = the chain value moved by up to half the window: what another
summation order may do), runs that classification with the device's formulas, and checks the claim against the exact selection."""
import numpy as np

from test_lazy_bounds_cpu import KAPPA, s1_u2

W0 = np.float32(2.0 ** -21.5)


def _chain(cent, q):
    """ascending-k fp32 chain of every (centroid, query token) pair: [C, Q] float32"""
    acc = np.zeros((cent.shape[0], q.shape[0]), np.float32)
    cf, qf = cent.astype(np.float32), q.astype(np.float32)
    for k in range(cent.shape[1]):
        acc = (cf[:, k : k + 1] * qf[None, :, k] + acc).astype(np.float32)
    return acc


def _asc_sum(m):
    """fp32 sum over the last axis in ascending column order (search.rs:401 as the oracle fixes it)"""
    tot = np.zeros(m.shape[:-1], np.float32)
    for j in range(m.shape[-1]):
        tot = (tot + m[..., j]).astype(np.float32)
    return tot


def _run(seed, scale):
    rng = np.random.default_rng(seed)
    C, D, N, L, Q, R = 2048, 64, 12000, 24, 32, 256
    cent = rng.standard_normal((C, D)).astype(np.float32)
    cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    cent = cent.astype(np.float16)
    # queries near a handful of centroids ("topics"), so that the top of the ranking is crowded
    topics = rng.choice(C, 6, replace=False)
    q = (cent[rng.choice(topics, Q)].astype(np.float32) + 0.35 * rng.standard_normal((Q, D)).astype(np.float32))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True) * scale).astype(np.float16)
    codes = rng.integers(0, C, (N, L))
    hot = rng.random((N, L)) < 0.08
    codes = np.where(hot, rng.choice(topics, (N, L)), codes)          # documents touching the topics
    codes[N // 2 :] = codes[: N - N // 2]                              # ... and exact duplicates: ties at every rank
    chain = _chain(cent, q)                                            # [C, Q]
    t = chain.astype(np.float16)                                       # the reference's S
    qn = np.linalg.norm(q.astype(np.float32), axis=1).astype(np.float32)
    cmax = np.float32(np.linalg.norm(cent.astype(np.float32), axis=1).max())
    w = (W0 * qn * cmax).astype(np.float32)                            # per column
    u_c = (w[None, :] + KAPPA * np.abs(chain)).astype(np.float32)
    x = (chain + (rng.uniform(-0.5, 0.5, chain.shape) * u_c).astype(np.float32)).astype(np.float32)
    u = (w[None, :] + KAPPA * np.abs(x)).astype(np.float32)
    assert np.all(np.abs(chain.astype(np.float64) - x) <= u)           # the window holds the chain value
    s = (x + u).astype(np.float32).astype(np.float16)                  # stored upper candidates
    assert np.all(s.astype(np.float32) >= t.astype(np.float32))
    a_true = _asc_sum(t[codes].max(axis=1).astype(np.float32))         # [N]
    colmax_docs = s[codes].max(axis=1)                                 # [N, Q] stored column maxima
    a_up = _asc_sum(colmax_docs.astype(np.float32))
    assert np.all(a_up >= a_true)
    # the query's slack (k_probe_tau: tight form from the column's overall stored maximum; valid while no scored document has a
    # negative column maximum -- k_approx's negflag; the model's documents all do better than that or the loose form applies)
    smax = s.max(axis=0)
    bits = smax.view(np.uint16)
    e = (bits & 0x7C00).astype(np.uint32)
    e = np.where(e < 0x2C00, 0x2C00, e) - 0x2800
    ulp = e.astype(np.uint16).view(np.float16).astype(np.float32)
    tight = (ulp + np.float32(2.0) * s1_u2(np.abs(smax.astype(np.float32)), w, KAPPA)).astype(np.float32)
    if (colmax_docs.astype(np.float32) < 0).any():
        bq = (w / W0 * np.float32(1.002)).astype(np.float32)
        ex = np.maximum(np.frexp(bq)[1] - 11, -14)
        loose = (np.ldexp(np.float32(1.0), ex) + np.float32(2.0) * s1_u2(bq, w, KAPPA)).astype(np.float32)
        slack = np.float32(np.maximum(loose, tight).sum())
    else:
        slack = np.float32(tight.sum())
    assert np.all(a_up.astype(np.float64) - a_true <= slack * 1.0000001)   # what the slack is for
    order_up = np.sort(a_up)[::-1]
    U = order_up[R - 1]
    dlt = np.float32(slack * np.float32(1.01) + np.float32(Q) * np.float32(2.4e-7) * (abs(U) + np.float32(1.0)))
    certain = a_up > U + dlt
    maybe = (a_up >= U - dlt) & ~certain
    ids = np.arange(N)
    exact_rank = np.lexsort((ids, -a_true.astype(np.float64)))
    want = set(exact_rank[:R].tolist())
    assert set(ids[certain].tolist()) <= want, "a 'certain' document outside the reference's selection"
    assert want <= set(ids[certain | maybe].tolist()), "a selected document classified as out"
    mids = ids[maybe]
    morder = np.lexsort((mids, -a_true[mids].astype(np.float64)))
    take = R - int(certain.sum())
    got = set(ids[certain].tolist()) | set(mids[morder[:take]].tolist())
    assert got == want
    return int(certain.sum()), int(maybe.sum())


def test_lazy_selection_equals_the_exact_one_unit_queries():
    nc, nm = _run(1, 1.0)
    assert nc + nm >= 256 and nm < 4000, (nc, nm)


def test_lazy_selection_equals_the_exact_one_scaled_queries():
    for seed, scale in ((2, 6.0), (3, 0.004)):   # an unnormalised query batch; a tiny-norm one (scores where fp16 is finest)
        _run(seed, scale)


def test_lazy_probe_threshold_collects_the_reference_top_n():
    """the probe under the lazy form (k_probe_tau / k_probe_collect / k_probe_merge): tau = the n_probe-th largest 128-centroid
    chunk maximum of the STORED column, lowered by s1_lower16; everything stored at or above it is collected and re-evaluated.
    Claim: the reference's top n_probe of the column (value desc, centroid id asc) is among the collected."""
    from test_lazy_bounds_cpu import f16_bits_to_f32, s1_lower16
    rng = np.random.default_rng(5)
    C, D, Q, n_probe = 16384, 64, 16, 8
    cent = rng.standard_normal((C, D)).astype(np.float32)
    cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    cent = cent.astype(np.float16)
    for scale in (1.0, 0.003, 9.0):
        q = rng.standard_normal((Q, D)).astype(np.float32)
        q = (q / np.linalg.norm(q, axis=1, keepdims=True) * scale).astype(np.float16)
        q[3] = 0                                                         # a zero row: exact zeros, window 0
        chain = _chain(cent, q)
        t = chain.astype(np.float16)
        qn = np.linalg.norm(q.astype(np.float32), axis=1).astype(np.float32)
        w = (W0 * qn * np.float32(np.linalg.norm(cent.astype(np.float32), axis=1).max())).astype(np.float32)
        u_c = (w[None, :] + KAPPA * np.abs(chain)).astype(np.float32)
        x = (chain + (rng.uniform(-0.5, 0.5, chain.shape) * u_c).astype(np.float32)).astype(np.float32)
        s = (x + (w[None, :] + KAPPA * np.abs(x)).astype(np.float32)).astype(np.float32).astype(np.float16)
        for col in range(Q):
            sc, tc = s[:, col], t[:, col].astype(np.float64)
            chunk_max = sc.reshape(C // 128, 128).max(axis=1)
            tau = np.sort(chunk_max)[::-1][n_probe - 1]
            low = f16_bits_to_f32(s1_lower16(np.array([tau]).view(np.uint16), np.array([w[col]], np.float32), KAPPA))[0]
            collected = np.nonzero(sc.astype(np.float32) >= low)[0]
            ref = np.lexsort((np.arange(C), -tc))[:n_probe]
            assert set(ref.tolist()) <= set(collected.tolist()), (scale, col, tau, low)
            if w[col] > 0:
                assert len(collected) < 4096, (scale, col, len(collected))   # ... and it stays a short list
