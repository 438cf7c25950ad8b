"""One rank of tests/test_hip_parity.py::test_sharded_and_replicated_multiprocess: two real processes, each with its own
document shard (and a full replica) on cuda:0, exchanging through gloo -- the production code path of
fast_plaid_amd.sharded with the HIP engine, only the transport differs from RCCL."""
import os
import sys

import torch  # FIRST (torch wheels bundle their own HIP runtime)
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import fast_plaid_amd as fp  # noqa: E402
from fast_plaid_amd import sharded  # noqa: E402


def main():
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    R = fp.fast_plaid_rust
    spec = fp.synth.SynthSpec(n_docs=12000, doc_len=48, n_centroids=1024, variable_len=True, seed=13)
    arr = fp.synth.host_index_arrays(spec)
    q = fp.synth.make_queries(spec, arr["centroids"], 7, 32)     # 7 queries over 2 ranks: uneven split
    params = R.SearchParameters(2000, 512, 100, 8)

    def mk(a, **kw):
        return R.construct_index(a["nbits"], a["centroids"], None, None, a["bucket_weights"], a["ivf"], a["ivf_lengths"],
                                 a["doc_codes"], a["doc_residuals"], a["doc_lengths"], "cuda:0", False, **kw)

    whole = mk(arr)
    want = R.search_arrays(whole, q, params)
    b, e = sharded.plan_shards(arr["doc_lengths"], world)[rank]
    eng = sharded.HipShardEngine(mk(sharded.shard_arrays(arr, b, e), pid_offset=b), "cuda:0")
    got = sharded.sharded_search(eng, q, params, dist=dist)
    assert all(np.array_equal(x, y) for x, y in zip(got, want)), f"rank {rank}: sharded != unsharded"
    rep = sharded.replicated_search(lambda qs: R.search_arrays(whole, qs, params), q, params.top_k, dist=dist, device="cpu")
    assert all(np.array_equal(x, y) for x, y in zip(rep, want)), f"rank {rank}: replicated != unsharded"
    if world == 4:
        # 2-D layout: 2 document shards x 2 query groups; the group's sharded search runs over its own process group
        D = 2
        d, g = sharded.plan_grid(world, D)[rank]
        groups = [dist.new_group([gg * D + dd for dd in range(D)]) for gg in range(world // D)]
        gb, ge = sharded.plan_shards(arr["doc_lengths"], D)[d]
        geng = sharded.HipShardEngine(mk(sharded.shard_arrays(arr, gb, ge), pid_offset=gb), "cuda:0")
        grid = sharded.replicated_search(lambda qs: sharded.sharded_search(geng, qs, params, dist=dist, group=groups[g]), q, params.top_k,
                                         dist=dist, device="cpu", group_size=D)
        assert all(np.array_equal(x, y) for x, y in zip(grid, want)), f"rank {rank}: 2 x 2 grid != unsharded"
        print("GRID_MP_OK rank", rank)
    dist.barrier()
    print("SHARD_MP_OK rank", rank)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
