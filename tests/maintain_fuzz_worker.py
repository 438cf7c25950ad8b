"""GPU worker: randomized fuzz of the index-directory writers -- create.py / maintain.py (device compression through fp_compress)
against the ATen restatement of rust/index/create.rs, update.rs, delete.rs in oracle/plaid_index_oracle_torch.py, run LIVE on
drawn inputs (tests/golden/maintain/snapshots.npz pins one fixed sequence).

Every case draws dim, nbits, the centroid count (create.rs's own estimate, half of it, a handful, twice it), the chunk size, the
documents, and a sequence of 2 - 6 operations (update with or without the threshold refresh, delete of random positions --
duplicates included); after every operation the two directories must hold the same files: integer arrays and packed bytes
identical, fp16 centroids identical, codec floats / thresholds / json numbers within an ulp-level tolerance.  (Twice the estimate:
the created directory must still be the reference's; the operations after it are checked against a one-shot rebuild, see run_case.)

usage: maintain_fuzz_worker.py <n_cases> <seed> [<first_case>]
"""
import json
import os
import shutil
import sys
import tempfile
import traceback

import torch  # FIRST (torch wheels bundle their own HIP runtime)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import fast_plaid_amd as fp  # noqa: E402,F401
import plaid_index_oracle_torch as IO  # noqa: E402
from fast_plaid_amd import create as CR, maintain as MT  # noqa: E402


def same_directory(tag, ours, theirs):
    want = IO.read_directory(theirs)
    have = set(os.listdir(ours))
    assert set(want) <= have, f"{tag}: missing files {sorted(set(want) - have)}"
    for fn, w in want.items():
        if fn.endswith(".npy"):
            g = np.load(os.path.join(ours, fn))
            assert g.shape == w.shape, f"{tag}/{fn}: shape {g.shape} vs {w.shape}"
            if w.dtype.kind in "iu":
                assert g.dtype.kind in "iu" and np.array_equal(g, w), f"{tag}/{fn} differs ({int((g != w).sum())} of {w.size})"
            elif fn == "centroids.npy":
                assert np.array_equal(g.astype(np.float16), w.astype(np.float16)), f"{tag}/{fn} differs"
            else:
                assert np.allclose(g.astype(np.float64), w.astype(np.float64), rtol=2e-6, atol=1e-9), f"{tag}/{fn}: {g} vs {w}"
        else:
            with open(os.path.join(ours, fn)) as f:
                gj = json.load(f)
            if isinstance(w, dict):
                for k, v in w.items():
                    assert k in gj, f"{tag}/{fn}: key {k} missing"
                    if isinstance(v, float):
                        assert abs(gj[k] - v) <= 1e-9 * max(1.0, abs(v)), f"{tag}/{fn}[{k}]: {gj[k]} vs {v}"
                    else:
                        assert gj[k] == v, f"{tag}/{fn}[{k}]: {gj[k]} vs {v}"
            else:
                assert gj == w, f"{tag}/{fn} differs"


def draw(rng):
    pick = lambda *v: v[int(rng.integers(0, len(v)))]   # noqa: E731
    dim = int(pick(128, 64, 96, 48, 32))
    nbits = int(pick(4, 4, 2, 1, 8))
    c_rule = int(pick(0, 0, 0, 1, 2, 3))   # centroid count: the reference's estimate / half of it / a handful / TWICE it (see below)
    n0 = int(pick(1, 3, 20, 60, 150))
    chunk = int(pick(1, 7, 25, 1000))
    max_len = int(pick(1, 5, 40))
    n_ops = int(rng.integers(2, 7))
    compress_only = bool(rng.random() < 0.15)
    return dim, nbits, c_rule, n0, chunk, max_len, n_ops, compress_only


def one_shot_ivf(path):
    """the lists a rebuild from every chunk's codes gives (create.rs:528-559)"""
    from fast_plaid_amd import synth
    meta = json.load(open(os.path.join(path, "metadata.json")))
    codes, lens = [], []
    for i in range(int(meta["num_chunks"])):
        codes.append(np.load(os.path.join(path, f"{i}.codes.npy")))
        lens += json.load(open(os.path.join(path, f"doclens.{i}.json")))
    return synth.build_ivf(np.concatenate(codes), np.asarray(lens, np.int64), int(meta["num_partitions"]))


def run_case(seed, case):
    rng = np.random.default_rng([seed, case, 13])
    dim, nbits, c_rule, n0, chunk, max_len, n_ops, co = draw(rng)
    g = torch.Generator().manual_seed(int(rng.integers(1, 1 << 30)))
    lens0 = [int(torch.randint(1, max_len + 1, (1,), generator=g)) for _ in range(n0)]
    T0 = sum(lens0)
    est = int(2 ** np.floor(np.log2(16.0 * np.sqrt(float(n0) * (float(T0) / float(n0))))))   # create.rs:292-294
    C = (est, max(1, est // 2), 5, 2 * est)[c_rule]
    cent = torch.nn.functional.normalize(torch.randn(C, dim, generator=g), dim=-1).half()

    def mkdocs(n, lens=None):
        out = []
        for i in range(n):
            ln = lens[i] if lens is not None else int(torch.randint(1, max_len + 1, (1,), generator=g))
            pk = torch.randint(0, C, (ln,), generator=g)
            d = cent[pk].float() + 0.4 * torch.randn(ln, dim, generator=g) / dim ** 0.5
            out.append(torch.nn.functional.normalize(d, dim=-1).half())
        return out

    docs = mkdocs(n0, lens0)
    allt = torch.cat(docs)
    heldout = allt[torch.randperm(allt.shape[0], generator=g)[: max(1, min(150, allt.shape[0]))]]
    tmp = tempfile.mkdtemp()
    try:
        ours, theirs = os.path.join(tmp, "ours"), os.path.join(tmp, "theirs")
        IO.create_index(docs, theirs, cent, nbits, heldout, batch_size=chunk, compress_only=co)
        CR.create_index(ours, [d.numpy() for d in docs], cent.numpy(), nbits=nbits, device="cuda:0", heldout=heldout.numpy(), chunk_docs=chunk,
                        compress_only=co)
        same_directory("created", ours, theirs)
        # More centroids than create.rs's own estimate (the k-means driver sizes k from a SAMPLE's average length, fast_plaid.py:150-154,
        # so the two can sit either side of a power of two): update.rs then walks `0..num_partitions` lists (:384-420) -- the old
        # lists beyond the estimate are cut off and new tokens on those centroids are dropped (:362-366), and it returns before
        # the metadata update when nothing is left (:371-373).  maintain.py keeps every list (deviations): in that
        # regime the sequence runs on our directory alone and is checked against a one-shot rebuild from its own chunks.
        mirrored = C <= est
        n_now = n0
        for op in range(n_ops):
            if rng.random() < 0.6 or n_now <= 1:
                new = mkdocs(int(rng.integers(1, 40)))
                thr = bool(rng.integers(0, 2))
                if mirrored:
                    IO.update_index(new, theirs, batch_size=chunk, update_threshold=thr)
                MT.update_index(ours, [d.numpy() for d in new], device="cuda:0", update_threshold=thr)
                n_now += len(new)
                tag = f"op{op}:update({len(new)},thr={thr})"
            else:
                k = int(rng.integers(1, max(2, n_now // 2)))
                sub = rng.integers(0, n_now, k).tolist()   # (duplicates on purpose)
                if mirrored:
                    IO.delete_from_index(sub, theirs)
                MT.delete_from_index(ours, sub)
                n_now -= len(set(sub))
                tag = f"op{op}:delete({k})"
            if mirrored:
                same_directory(tag, ours, theirs)
            else:
                if not os.path.exists(os.path.join(ours, "ivf.npy")):
                    continue   # (still compress_only: no lists yet)
                ivf, ivfl = one_shot_ivf(ours)
                assert np.array_equal(np.load(os.path.join(ours, "ivf.npy")), ivf), f"{tag}: ivf.npy is not the one-shot rebuild"
                assert np.array_equal(np.load(os.path.join(ours, "ivf_lengths.npy")), ivfl), f"{tag}: ivf_lengths.npy is not the one-shot rebuild"
                meta = json.load(open(os.path.join(ours, "metadata.json")))
                assert meta["num_documents"] == n_now and meta["num_partitions"] == est, f"{tag}: metadata {meta}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    torch.set_num_threads(2)
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    bad = 0
    for case in range(first, first + n):
        try:
            run_case(seed, case)
        except Exception as e:   # noqa: BLE001
            bad += 1
            print(f"FUZZ_CASE_FAILED case={case} seed={seed} draw={draw(np.random.default_rng([seed, case, 13]))} {type(e).__name__}: {str(e)[:500]}",
                  flush=True)
            if os.environ.get("FP_FUZZ_TRACEBACK"):
                traceback.print_exc()
    print(f"FUZZ_FAIL {bad}/{n}" if bad else f"FUZZ_OK {n}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
