"""One rank of the world_size-2 gloo protocol test (spawned by tests/test_host_cpu.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fast_plaid_amd as fp  # noqa: E402
import plaid_oracle as OC  # noqa: E402
from fast_plaid_amd import sharded  # noqa: E402


class OracleShardEngine:
    """Stand-in for HipShardEngine built on the CPU oracle (TEST ONLY): same stage contract
    as include/fastplaid.h fp_shard_stage1/2/3/4."""

    def __init__(self, arr, begin, end):
        self.begin_, self.end_ = begin, end
        sh = sharded.shard_arrays(arr, begin, end)
        self.idx = OC.OracleIndex(nbits=sh["nbits"], centroids=sh["centroids"], bucket_weights=sh["bucket_weights"], ivf=sh["ivf"],
                                  ivf_lengths=sh["ivf_lengths"], doc_codes=sh["doc_codes"], doc_residuals=sh["doc_residuals"],
                                  doc_lengths=sh["doc_lengths"])

    def begin(self, q, params):
        return dict(q=np.ascontiguousarray(q, np.float16), B=q.shape[0], R=max(params.n_full_scores // 4, 1), p=params,
                    top_k=params.top_k)

    def stage1(self, st):
        B, R, p = st["B"], st["R"], st["p"]
        rec = np.zeros((B, R), sharded.REC1_DTYPE)
        rec["pid"] = -1
        rec["approx"] = -np.inf
        for b in range(B):
            t = self.idx.search_trace(st["q"][b], p.top_k, p.n_full_scores, p.n_ivf_probe)
            amap = dict(zip(t["cand"].tolist(), t["approx"].tolist()))
            for i, d in enumerate(t["rerank"].tolist()):
                rec[b, i]["approx"] = amap[d]
                rec[b, i]["pid"] = d + self.begin_
        return torch.from_numpy(rec.view(np.uint8).reshape(B, R * sharded.REC1_BYTES))

    def stage2(self, st, all_rec1, world):
        B, R = st["B"], st["R"]
        all1 = all_rec1.numpy().reshape(world, B, R * sharded.REC1_BYTES).view(sharded.REC1_DTYPE).reshape(world, B, R)
        rec = np.zeros((B, R), sharded.REC2_DTYPE)
        rec["pid"] = -1
        rec["score"] = -np.inf
        for b in range(B):
            a = all1[:, b, :]["approx"].reshape(-1)
            p = all1[:, b, :]["pid"].reshape(-1)
            ok = p >= 0
            a, p = a[ok], p[ok]
            order = np.lexsort((p, -a))[:R]  # (approx desc, id asc)
            mine = np.sort(p[order][(p[order] >= self.begin_) & (p[order] < self.end_)])
            if len(mine):
                ex = self.idx.exact_scores(st["q"][b], mine - self.begin_)   # the oracle's scores are exact: budget 0
                rec[b, : len(mine)]["score"] = ex
                rec[b, : len(mine)]["pid"] = mine
        return torch.from_numpy(rec.view(np.uint8).reshape(B, R * sharded.REC2_BYTES))

    def stage3(self, st, all_rec2, world, rank):
        # the oracle's scores are exact (budget 0): nothing is marked, the third message carries no information
        st["all2"] = all_rec2.numpy().reshape(world, st["B"], st["R"] * sharded.REC2_BYTES).view(sharded.REC2_DTYPE).reshape(world, st["B"], st["R"])
        return torch.zeros((st["B"], st["R"]), dtype=torch.float32)

    def stage4(self, st, all_x, world):
        B, k = st["B"], st["top_k"]
        assert tuple(all_x.shape) == (world, B, st["R"])
        all2 = st["all2"]
        pids = np.full((B, k), -1, np.int64)
        scores = np.zeros((B, k), np.float32)
        counts = np.zeros(B, np.int32)
        for b in range(B):
            s = all2[:, b, :]["score"].reshape(-1)
            p = all2[:, b, :]["pid"].reshape(-1)
            ok = p >= 0
            s, p = s[ok], p[ok]
            order = np.lexsort((p, -s))[:k]
            counts[b] = len(order)
            pids[b, : len(order)] = p[order]
            scores[b, : len(order)] = s[order]
        return pids, scores, counts

    def end(self, st):
        pass


def main():
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    spec = fp.synth.SynthSpec(n_docs=600, doc_len=40, n_centroids=256, variable_len=True, seed=11)
    arr = fp.synth.host_index_arrays(spec)
    q = fp.synth.make_queries(spec, arr["centroids"], 5, 16)
    params = fp.fast_plaid_rust.SearchParameters(2000, 128, 20, 4)
    b, e = sharded.plan_shards(arr["doc_lengths"], world)[rank]
    eng = OracleShardEngine(arr, b, e)
    pids, scores, counts = sharded.sharded_search(eng, q, params, dist=dist)
    whole = OC.OracleIndex(nbits=arr["nbits"], centroids=arr["centroids"], bucket_weights=arr["bucket_weights"], ivf=arr["ivf"],
                           ivf_lengths=arr["ivf_lengths"], doc_codes=arr["doc_codes"], doc_residuals=arr["doc_residuals"],
                           doc_lengths=arr["doc_lengths"])
    ref = whole.search(q, 20, 128, 4)
    for i in range(5):
        assert counts[i] == len(ref[i][0]), (counts[i], len(ref[i][0]))
        assert np.array_equal(pids[i, : counts[i]], ref[i][0]), (rank, i, pids[i], ref[i][0])
        assert np.array_equal(scores[i, : counts[i]], ref[i][1])
    # replicated index, batch split across ranks (uneven: 5 queries over `world` ranks)
    def search_local(qs):
        r = whole.search(qs, 20, 128, 4)
        P = np.full((len(r), 20), -1, np.int64)
        S = np.zeros((len(r), 20), np.float32)
        Cn = np.zeros(len(r), np.int32)
        for i, (pp, ss) in enumerate(r):
            P[i, : len(pp)], S[i, : len(pp)], Cn[i] = pp, ss, len(pp)
        return P, S, Cn
    rp, rs, rc = sharded.replicated_search(search_local, q, 20, dist=dist)
    assert rp.shape == (5, 20)
    for i in range(5):
        assert rc[i] == len(ref[i][0])
        assert np.array_equal(rp[i, : rc[i]], ref[i][0]) and np.array_equal(rs[i, : rc[i]], ref[i][1])
    # more ranks than queries: some ranks hold an empty slice
    rp1, rs1, rc1 = sharded.replicated_search(search_local, q[:1], 20, dist=dist)
    assert rp1.shape == (1, 20) and np.array_equal(rp1[0, : rc1[0]], ref[0][0])
    # 2-D layout (world 4: 2 document shards x 2 query groups): the group's document-sharded search over its own process
    # group, the batch cut into one slice per group, one result all-gather over the world
    if world % 2 == 0 and world >= 4:
        D = 2
        grid = sharded.plan_grid(world, D)
        d, g = grid[rank]
        groups = [dist.new_group([gg * D + dd for dd in range(D)]) for gg in range(world // D)]   # (every rank creates every group)
        # the communicator bootstrap of the native path (NativeComm.from_torch_dist) over each sub-group: the group's first
        # member's 128 bytes must reach its members -- also in the groups that do not contain global rank 0
        mine = bytes([g * 16 + 1] * 128) if dist.get_rank(groups[g]) == 0 else None
        got = sharded.group_broadcast_bytes(dist, mine, groups[g])
        assert got == bytes([g * 16 + 1] * 128), (rank, g, got[:4])
        gb, ge = sharded.plan_shards(arr["doc_lengths"], D)[d]
        geng = OracleShardEngine(arr, gb, ge)
        gp, gs, gc = sharded.replicated_search(lambda qs: sharded.sharded_search(geng, qs, params, dist=dist, group=groups[g]), q, 20,
                                               dist=dist, group_size=D)
        assert gp.shape == (5, 20)
        for i in range(5):
            assert gc[i] == len(ref[i][0])
            assert np.array_equal(gp[i, : gc[i]], ref[i][0]) and np.array_equal(gs[i, : gc[i]], ref[i][1]), (rank, i)
        # one query over two groups: the second group's slice is empty
        gp1, gs1, gc1 = sharded.replicated_search(lambda qs: sharded.sharded_search(geng, qs, params, dist=dist, group=groups[g]), q[:1], 20,
                                                  dist=dist, group_size=D)
        assert gp1.shape == (1, 20) and np.array_equal(gp1[0, : gc1[0]], ref[0][0])
        print("GRID_OK rank", rank, "shard", d, "group", g)
    dist.barrier()
    print("SHARDED_OK rank", rank)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
