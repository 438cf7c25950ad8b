#!/usr/bin/env python
"""CPU simulation (numpy): would a second, cheap bound between level 0 and the exact rescoring pay?  (Design aid, no GPU.)

Level 0:  UB0(d) = F + sum_{c in codes(d)} e(c),  e(c) = sum_q ex[c][q],  ex = max(0, bin - floor).
Level 1a: split every centroid's excess into its LARGEST entry (column qmax(c), value emax(c)) and the rest:
          UB1(d) = F + sum_c (e(c) - emax(c)) + sum_q max_{c: qmax(c) = q} emax(c)       (4 bytes per centroid, L2-resident table)
Level 1b: the same with the TWO largest entries split off.
Survivor rule as in the product: bound + Q > T4, T4 = R-th largest exact K among the top-4R by UB0.
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_plaid_amd as fp
S = fp.synth

N = int(os.environ.get("SIM_DOCS", 1_000_000)); L = 128; Q = 32; NPROBE = 8; R = 1024
spec = S.SynthSpec(n_docs=N, doc_len=L, n_centroids=S.default_num_centroids(N * L), seed=42)
C = spec.n_centroids
cache = f"/tmp/sim/codes_{N}.npy"
t0 = time.time()
if os.path.exists(cache):
    codes = np.load(cache)
else:
    codes = np.empty((N, L), np.int32)
    step = 8192
    for d0 in range(0, N, step):
        d1 = min(N, d0 + step)
        pids = np.repeat(np.arange(d0, d1, dtype=np.int64), L)
        tok = np.arange(d0 * L, d1 * L, dtype=np.uint64)
        codes[d0:d1] = S.token_codes(spec, pids, tok).reshape(d1 - d0, L)
    codes.sort(axis=1)
    np.save(cache, codes)
print("codes", codes.shape, "C", C, "%.1fs" % (time.time() - t0), flush=True)
uniq = np.ones(codes.shape, bool); uniq[:, 1:] = codes[:, 1:] != codes[:, :-1]
cent = S.centroids(spec)
cent32 = cent.astype(np.float32)
nq = int(os.environ.get("SIM_QUERIES", 4))
qs = S.make_queries(spec, cent, nq, Q, seed=1000)
pct = float(os.environ.get("SIM_FLOOR", 97.5))
for b in range(nq):
    q = qs[b].astype(np.float32)
    Sc = (cent32 @ q.T).astype(np.float16)
    S8 = np.clip(np.floor(Sc.astype(np.float32) * 128.0) + 100, 0, 255).astype(np.int32)
    cells = np.unique(np.argpartition(-Sc.astype(np.float32), NPROBE, axis=0)[:NPROBE].ravel())
    iscell = np.zeros(C, bool); iscell[cells] = True
    cand = np.nonzero((iscell[codes]).any(1))[0]
    cc = codes[cand]; um = uniq[cand]; n = len(cand)
    colmax = np.zeros((n, Q), np.int32)
    for j in range(L):
        np.maximum(colmax, S8[cc[:, j]], out=colmax)
    K = colmax.sum(1)
    f = np.percentile(S8, pct, axis=0).astype(np.int32)
    F = int(f.sum())
    exc = np.maximum(S8 - f[None, :], 0)
    e = exc.sum(1)
    ub0 = F + (e[cc] * um).sum(1)
    top = np.argsort(-ub0)[: 4 * R]
    T4 = np.sort(K[top])[-R]
    surv = np.nonzero(ub0 + Q > T4)[0]
    # level 1 on the survivors
    order = np.argsort(-exc, axis=1)
    q1 = order[:, 0]; e1 = np.take_along_axis(exc, order[:, :1], 1)[:, 0]
    q2 = order[:, 1]; e2 = np.take_along_axis(exc, order[:, 1:2], 1)[:, 0]
    sc = cc[surv]; su = um[surv]; ns = len(surv)
    rest1 = ((e - e1)[sc] * su).sum(1)
    rest2 = ((e - e1 - e2)[sc] * su).sum(1)
    m1 = np.zeros((ns, Q), np.int32); m2 = np.zeros((ns, Q), np.int32)
    rows = np.arange(ns)
    for j in range(L):
        cj = sc[:, j]
        np.maximum.at(m1, (rows, q1[cj]), e1[cj])
        np.maximum.at(m2, (rows, q1[cj]), e1[cj])
        np.maximum.at(m2, (rows, q2[cj]), e2[cj])
    ub1 = F + rest1 + m1.sum(1)
    ub2 = F + rest2 + m2.sum(1)
    Ks = K[surv]
    assert (ub1 >= Ks).all() and (ub2 >= Ks).all() and (ub1 <= ub0[surv]).all()
    s1 = int((ub1 + Q > T4).sum()); s2 = int((ub2 + Q > T4).sum())
    need = int((Ks + Q > T4).sum())
    print(f"q{b}: cand {n}  survivors of level 0 {ns} | level 1a (1 entry split off) {s1} | level 1b (2 entries) {s2} | exact bins would keep {need}", flush=True)
    # pilot size: exactly scored documents = pilot U survivors(T_pilot)
    line = []
    srt = np.argsort(-ub0)
    for mult in (1.0, 1.5, 2.0, 3.0, 4.0, 6.0):
        p = int(mult * R)
        Tp = np.sort(K[srt[:p]])[-R]
        sv = ub0 + Q > Tp
        tot = int(sv.sum()) + int((~sv[srt[:p]]).sum())
        line.append(f"{mult}R: T {Tp} scored {tot}")
    print("     pilot size -> exactly scored:", " | ".join(line), flush=True)
