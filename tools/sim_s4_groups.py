#!/usr/bin/env python
"""round 6, simulation (torch on the GPU box): would a first bound over centroid GROUPS prune the build-from-vectors corpus?
The centroids are clustered into K groups (k-means over the centroids); S'[g][q] = max over the group's members of S[c][q];
UB'(d) = sum_q max over d's codes of S'[group(code)][q] >= the approximate score.  With the best threshold there is (the R-th best
approximate score of the query's candidates) -- how many candidates have UB' >= it?  (An LDS-resident table: K x 32 bytes.)"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    z = np.load(sys.argv[1] if len(sys.argv) > 1 else "/tmp/gmm_corpus.npz")
    dev = "cuda"
    cent = torch.from_numpy(z["centroids"].astype(np.float32)).to(dev)
    codes = torch.from_numpy(z["doc_codes"].astype(np.int64)).to(dev)
    lens = z["doc_lengths"]
    N, L = lens.shape[0], int(lens[0])
    assert (lens == L).all()
    codes = codes[: N * L].view(N, L)
    C = cent.shape[0]
    pool = torch.from_numpy(z["query_pool"].astype(np.float32)).to(dev)
    g = torch.Generator(device=dev).manual_seed(5)
    R = 1024
    out = {}
    for K in (2048, 4096, 8192):
        # k-means over the centroids
        ctr = cent[torch.randperm(C, device=dev, generator=g)[:K]].clone()
        for _ in range(8):
            a = torch.empty(C, dtype=torch.long, device=dev)
            for s in range(0, C, 8192):
                a[s:s + 8192] = (cent[s:s + 8192] @ ctr.T).argmax(1)
            sums = torch.zeros_like(ctr).index_add_(0, a, cent)
            cnt = torch.bincount(a, minlength=K).clamp(min=1).unsqueeze(1)
            ctr = torch.nn.functional.normalize(sums / cnt, dim=1)
        grp = a
        gcodes = grp[codes]   # [N, L]
        res = []
        for qi in range(6):
            q = pool[torch.randint(0, pool.shape[0], (32,), device=dev, generator=g)] + 0.3 * torch.randn(32, cent.shape[1], device=dev, generator=g) / np.sqrt(cent.shape[1])
            q = torch.nn.functional.normalize(q, dim=1)
            S = (cent @ q.T).half().float()            # [C, 32]
            Sg = torch.full((K, 32), -1e9, device=dev).scatter_reduce_(0, grp.unsqueeze(1).expand(C, 32), S, "amax")
            # candidates: documents holding one of a token's 8 best centroids
            top = S.topk(8, dim=0).indices.flatten().unique()
            mark = torch.zeros(C, dtype=torch.bool, device=dev)
            mark[top] = True
            cand = mark[codes].any(1).nonzero().flatten()
            approx = torch.empty(cand.shape[0], device=dev)
            ub = torch.empty(cand.shape[0], device=dev)
            for s in range(0, cand.shape[0], 4096):
                c = codes[cand[s:s + 4096]]
                approx[s:s + 4096] = S[c].amax(1).sum(1)
                ub[s:s + 4096] = Sg[gcodes[cand[s:s + 4096]]].amax(1).sum(1)
            assert (ub >= approx - 1e-4).all()
            thr = approx.topk(min(R, approx.shape[0])).values[-1]
            res.append((int(cand.shape[0]), int((ub >= thr).sum()), float((ub - approx).mean()), float(approx.std())))
        out[K] = {"candidates": int(np.mean([r[0] for r in res])), "survivors": int(np.mean([r[1] for r in res])),
                  "survivor_fraction": round(float(np.mean([r[1] / r[0] for r in res])), 4),
                  "mean_looseness": round(float(np.mean([r[2] for r in res])), 3), "std_of_the_scores": round(float(np.mean([r[3] for r in res])), 3),
                  "distinct_groups_per_doc": round(float(torch.stack([torch.unique(gcodes[i]).numel() * torch.ones(()) for i in range(0, N, N // 200)]).mean()), 1)}
        print(K, json.dumps(out[K]), flush=True)


if __name__ == "__main__":
    main()
