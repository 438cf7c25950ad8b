"""GPU worker: randomized fuzz of the document-sharded search (fp_shard_stage1..4, the exchanges done by concatenation in one
process) against the unsharded fp_search on the same arrays: identical ids, scores and counts on every shard's rank, bit for bit.

Every case draws the number of shards (2 - 5), the corpus (plain random arrays -- empty documents, shards that hold next to
nothing -- or the corpus model), batch, query length, n_ivf_probe, n_full_scores, top_k.  FP_TEST=shard_big=1 in the environment
forces the sort-free cut / union of large unions.

usage: shard_fuzz_worker.py <n_cases> <seed> [<first_case>] [native]      native: see run_native (fp_shard_search, one rank over RCCL)
"""
import os
import sys
import traceback

import torch  # FIRST (torch wheels bundle their own HIP runtime)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import fast_plaid_amd as fp  # noqa: E402
from fast_plaid_amd import sharded  # noqa: E402
from test_hip_parity import _random_arrays  # noqa: E402

R = fp.fast_plaid_rust


def mk(a, **kw):
    return R.construct_index(a["nbits"], a["centroids"], None, None, a["bucket_weights"], a["ivf"], a["ivf_lengths"],
                             a["doc_codes"], a["doc_residuals"], a["doc_lengths"], "cuda:0", False, **kw)


def draw(rng):
    pick = lambda *v: v[int(rng.integers(0, len(v)))]   # noqa: E731
    G = int(rng.integers(2, 6))
    model = int(pick(0, 0, 1))
    B = int(pick(1, 2, 5, 12, 33))
    Q = int(pick(1, 7, 20, 32, 32, 33, 64, 100))
    n_probe = int(pick(1, 2, 4, 8, 8, 16, 32, 40))
    n_full = int(pick(1, 4, 8, 64, 256, 512, 1000, 4096))
    top_k = int(pick(1, 10, 100, 1000))
    if model:
        shape = (int(pick(3000, 9000, 25000) * rng.uniform(0.7, 1.3)), int(pick(16, 48, 64, 128)), int(pick(512, 2048, 8192)),
                 *pick((128, 4), (128, 2), (64, 4)), int(rng.integers(1, 1 << 30)))
    else:
        shape = (max(1, int(pick(1, 3, 40, 300, 900, 2500) * rng.uniform(0.5, 1.5))), int(pick(1, 2, 8, 33, 70, 130)),
                 int(pick(8, 37, 129, 257, 1000, 3001)), int(pick(128, 128, 64, 96, 48)), int(pick(4, 4, 2, 1, 8)), 0)
    return G, model, B, Q, n_probe, n_full, top_k, shape


def run_case(seed, case):
    rng = np.random.default_rng([seed, case, 5])
    G, model, B, Q, n_probe, n_full, top_k, shape = draw(rng)
    if model:
        n_docs, doc_len, C, dim, nbits, sseed = shape
        spec = fp.synth.SynthSpec(n_docs=n_docs, doc_len=doc_len, n_centroids=C, dim=dim, nbits=nbits, variable_len=True, seed=sseed)
        arr = fp.synth.host_index_arrays(spec)
        q = fp.synth.make_queries(spec, arr["centroids"], B, Q, seed=int(rng.integers(1, 1 << 30)))
    else:
        n_docs, max_len, C, dim, nbits, _ = shape
        arr = _random_arrays(rng, n_docs, max_len, C, dim, nbits, empty_frac=float(rng.choice([0.0, 0.1, 0.5])))
        pk = rng.integers(0, C, (B, Q))
        q = arr["centroids"][pk].astype(np.float32) + 0.3 * rng.standard_normal((B, Q, dim), dtype=np.float32) / np.sqrt(dim)
        q /= np.linalg.norm(q, axis=2, keepdims=True)
        q = q.astype(np.float16)
    n_probe = min(n_probe, C)
    params = R.SearchParameters(2000, n_full, top_k, n_probe)
    whole = mk(arr)
    pids, scores, counts = R.search_arrays(whole, q, params)
    ranges = sharded.plan_shards(arr["doc_lengths"], G)
    engines = [sharded.HipShardEngine(mk(sharded.shard_arrays(arr, b, e), pid_offset=b), "cuda:0") for (b, e) in ranges]
    sts = [en.begin(q, params) for en in engines]
    all1 = torch.stack([en.stage1(st) for en, st in zip(engines, sts)]).contiguous()
    torch.cuda.synchronize()
    all2 = torch.stack([en.stage2(st, all1, G) for en, st in zip(engines, sts)]).contiguous()
    torch.cuda.synchronize()
    allx = torch.stack([en.stage3(st, all2, G, r) for r, (en, st) in enumerate(zip(engines, sts))]).contiguous()
    torch.cuda.synchronize()
    for r, (en, st) in enumerate(zip(engines, sts)):
        p3, s3, c3 = en.stage4(st, allx, G)
        assert np.array_equal(c3, counts), f"rank {r}: counts {c3.tolist()} vs {counts.tolist()}"
        for b in range(B):
            assert np.array_equal(p3[b, : c3[b]], pids[b, : counts[b]]), f"rank {r} query {b}: ids differ"
            assert np.array_equal(s3[b, : c3[b]], scores[b, : counts[b]]), f"rank {r} query {b}: scores differ"
        en.end(st)


def run_native(seed, n_cases):
    """fp_shard_search -- the three all-gathers issued by the library itself on its search stream -- with ONE rank over RCCL (all a
    one-GPU box allows): drawn corpora, batches and parameters, every result == fp_search on the same index, with fp_search
    calls (graph capture and replay on the same scratch pool) in between."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    bad, comm = 0, None
    for case in range(n_cases):
        rng = np.random.default_rng([seed, case, 6])
        G, model, B, Q, n_probe, n_full, top_k, shape = draw(rng)
        try:
            if model:
                n_docs, doc_len, C, dim, nbits, sseed = shape
                spec = fp.synth.SynthSpec(n_docs=n_docs, doc_len=doc_len, n_centroids=C, dim=dim, nbits=nbits, variable_len=True, seed=sseed)
                arr = fp.synth.host_index_arrays(spec)
                q = fp.synth.make_queries(spec, arr["centroids"], B, Q, seed=int(rng.integers(1, 1 << 30)))
            else:
                n_docs, max_len, C, dim, nbits, _ = shape
                arr = _random_arrays(rng, n_docs, max_len, C, dim, nbits, empty_frac=float(rng.choice([0.0, 0.1, 0.5])))
                pk = rng.integers(0, C, (B, Q))
                q = arr["centroids"][pk].astype(np.float32) + 0.3 * rng.standard_normal((B, Q, dim), dtype=np.float32) / np.sqrt(dim)
                q = (q / np.linalg.norm(q, axis=2, keepdims=True)).astype(np.float16)
            params = R.SearchParameters(2000, n_full, top_k, min(n_probe, C))
            index = mk(arr)
            if comm is None:
                comm = sharded.NativeComm.from_torch_dist(index.device_id, dist)
            for rep in range(3):
                want = R.search_arrays(index, q, params)
                got = sharded.native_sharded_search(index, comm, q, params)
                for x, y in zip(got, want):
                    assert np.array_equal(x, y), f"rep {rep}: fp_shard_search differs from fp_search"
        except Exception as e:   # noqa: BLE001
            bad += 1
            print(f"FUZZ_CASE_FAILED native case={case} seed={seed} draw={draw(np.random.default_rng([seed, case, 6]))} {type(e).__name__}: {str(e)[:400]}",
                  flush=True)
            if os.environ.get("FP_FUZZ_TRACEBACK"):
                traceback.print_exc()
    if comm is not None:
        comm.close()
    dist.barrier()
    dist.destroy_process_group()
    print(f"FUZZ_FAIL {bad}/{n_cases}" if bad else f"FUZZ_OK {n_cases}", flush=True)
    sys.exit(1 if bad else 0)


def main():
    if len(sys.argv) > 4 and sys.argv[4] == "native":
        run_native(int(sys.argv[2]), int(sys.argv[1]))
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    bad = 0
    for case in range(first, first + n):
        try:
            run_case(seed, case)
        except Exception as e:   # noqa: BLE001
            bad += 1
            print(f"FUZZ_CASE_FAILED case={case} seed={seed} draw={draw(np.random.default_rng([seed, case, 5]))} {type(e).__name__}: {str(e)[:400]}",
                  flush=True)
            if os.environ.get("FP_FUZZ_TRACEBACK"):
                traceback.print_exc()
    print(f"FUZZ_FAIL {bad}/{n}" if bad else f"FUZZ_OK {n}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
