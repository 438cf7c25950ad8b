"""GPU worker for tests/test_hip_parity.py::test_sharded_equals_unsharded."""
import os
import sys

import torch  # FIRST: see the test's docstring

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import fast_plaid_amd as fp  # noqa: E402
from fast_plaid_amd import sharded  # noqa: E402


def main():
    assert torch.cuda.is_available()
    R = fp.fast_plaid_rust
    spec = fp.synth.SynthSpec(n_docs=9000, doc_len=64, n_centroids=2048, variable_len=True, seed=7)
    arr = fp.synth.host_index_arrays(spec)
    q = fp.synth.make_queries(spec, arr["centroids"], 12, 32)
    params = R.SearchParameters(2000, 512, 100, 8)

    def mk(a, **kw):
        return R.construct_index(a["nbits"], a["centroids"], None, None, a["bucket_weights"], a["ivf"], a["ivf_lengths"],
                                 a["doc_codes"], a["doc_residuals"], a["doc_lengths"], "cuda:0", False, **kw)

    whole = mk(arr)
    pids, scores, counts = R.search_arrays(whole, q, params)
    G = 3
    ranges = sharded.plan_shards(arr["doc_lengths"], G)
    engines = [sharded.HipShardEngine(mk(sharded.shard_arrays(arr, b, e), pid_offset=b), "cuda:0") for (b, e) in ranges]
    sts = [en.begin(q, params) for en in engines]
    all1 = torch.stack([en.stage1(st) for en, st in zip(engines, sts)]).contiguous()    # [G, B, R*16] bytes == an all-gather's layout
    torch.cuda.synchronize()
    all2 = torch.stack([en.stage2(st, all1, G) for en, st in zip(engines, sts)]).contiguous()
    torch.cuda.synchronize()
    allx = torch.stack([en.stage3(st, all2, G, r) for r, (en, st) in enumerate(zip(engines, sts))]).contiguous()   # [G, B, R] f32
    torch.cuda.synchronize()
    for en, st in zip(engines, sts):
        p3, s3, c3 = en.stage4(st, allx, G)
        assert np.array_equal(c3, counts), (c3, counts)
        for b in range(q.shape[0]):
            assert np.array_equal(p3[b, : c3[b]], pids[b, : counts[b]]), b
            assert np.array_equal(s3[b, : c3[b]], scores[b, : counts[b]]), b
        en.end(st)
    # single-rank sharded_search (world 1, no process group) == fp_search
    p1, s1_, c1 = sharded.sharded_search(sharded.HipShardEngine(whole, "cuda:0"), q, params)
    assert np.array_equal(c1, counts) and np.array_equal(p1, pids) and np.array_equal(s1_, scores)
    print("SHARDED_GPU_OK")


if __name__ == "__main__":
    main()
