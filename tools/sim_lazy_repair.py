"""How many chains would a LAZY repair of the centroid scores need?    CPU model on a small synthetic
corpus: the certification flags of S1 (same window as the kernel, the chain value standing in for the MFMA accumulator), and the
fraction of per-(document, query column) maxima of the refine whose largest STORED value is a flagged one -- the only
scores a lazy repair has to re-evaluate.  python tools/sim_lazy_repair.py -> flagged fraction 0.052 (the device counts 0.052),
flagged column maxima 0.0076 over all documents, 0.0056 over the 400 best per query."""
import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_plaid_amd as fp
s = fp.synth
C = 16384; ND = 60000
spec = s.SynthSpec(n_docs=ND, doc_len=128, n_centroids=C, seed=42)
cent = s.centroids(spec).astype(np.float16)          # [C,128]
q = s.make_queries(spec, cent, 8, 32, seed=2000)     # [8,32,128] f16
q = np.asarray(q, np.float16)
# ascending fp32 chain
cf = cent.astype(np.float32); 
res = []
for b in range(q.shape[0]):
    qf = q[b].astype(np.float32)                      # [32,128]
    acc = np.zeros((C, 32), np.float32)
    for k in range(128):
        acc = (cf[:, k:k+1] * qf[None, :, k] + acc).astype(np.float32)   # not fused, close enough for statistics
    res.append(acc)
X = np.stack(res)                                     # [8,C,32] fp32
qn = np.linalg.norm(q.astype(np.float32), axis=-1)    # [8,32]
cmax = np.linalg.norm(cf, axis=1).max()
u = (2.0**-21.5) * qn[:, None, :] * cmax + (2.0**-20) * np.abs(X)
hu = (X + u).astype(np.float16); hl = (X - u).astype(np.float16)
flag = hu != hl
print("flagged fraction", flag.mean())
# documents: codes
pids = np.arange(ND)
tok = (pids[:, None] * 128 + np.arange(128)[None, :]).reshape(-1)
codes = s.token_codes(spec, np.repeat(pids, 128), tok).reshape(ND, 128)
tot = 0; fl = 0; topfl = 0; toptot = 0
for b in range(q.shape[0]):
    Sb = hu[b].astype(np.float32)                     # stored upper candidates [C,32]
    Fb = flag[b]
    sc = Sb[codes]                                    # [ND,128,32]
    am = sc.argmax(axis=1)                            # [ND,32] token index of the max
    mx = np.take_along_axis(sc, am[:, None, :], 1)[:, 0, :]
    cm = np.take_along_axis(codes[:, :, None].repeat(32, 2), am[:, None, :], 1)[:, 0, :]   # code of the max
    f = Fb[cm, np.arange(32)[None, :]]
    approx = mx.sum(1)
    top = np.argsort(-approx)[:400]                   # the documents a refine would score (top by approx)
    tot += f.size; fl += f.sum(); toptot += f[top].size; topfl += f[top].sum()
print("flagged column maxima: all docs %.4f, top docs %.4f" % (fl / tot, topfl / toptot))
