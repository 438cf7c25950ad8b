"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/fastplaid.h
declares, host-side logic (shard planning, subset flattening, index directory I/O, synthetic
corpus twin), and the world_size-2 gloo test of the sharded-search protocol."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ctypes
    import fast_plaid_amd  # noqa: F401
    from fast_plaid_amd import _native
    hdr = open(os.path.join(ROOT, "include", "fastplaid.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fp_[A-Za-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    L = ctypes.CDLL(_native.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in fastplaid.h but not exported"
    assert declared == set(_native.SYMBOLS), f"binding table out of sync: {declared ^ set(_native.SYMBOLS)}"
    _native.lib()
    assert b"gfx950" in _native.lib().fp_version()


def test_product_has_no_cpu_fallback():
    """the product must fail loudly without a device, never route through the oracle."""
    import fast_plaid_amd as fp
    from fast_plaid_amd import _native
    if _native.lib().fp_device_count() > 0:
        pytest.skip("a GPU is visible")
    spec = fp.synth.SynthSpec(n_docs=50, doc_len=16, n_centroids=64)
    arr = fp.synth.host_index_arrays(spec)
    with pytest.raises(ValueError):
        fp.fast_plaid_rust.construct_index(arr["nbits"], arr["centroids"], None, None, arr["bucket_weights"], arr["ivf"],
                                           arr["ivf_lengths"], arr["doc_codes"], arr["doc_residuals"], arr["doc_lengths"], "cuda:0")
    with pytest.raises(ValueError, match="cpu"):
        fp.fast_plaid_rust.construct_index(arr["nbits"], arr["centroids"], None, None, arr["bucket_weights"], arr["ivf"],
                                           arr["ivf_lengths"], arr["doc_codes"], arr["doc_residuals"], arr["doc_lengths"], "cpu")
    for root, _, files in os.walk(os.path.join(ROOT, "fast-plaid_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"import\s+plaid_oracle|libplaid_oracle|from\s+oracle|oracle/_ref", src), \
                    f"{f} links/imports the oracle"
                # the only library the product binds at run time is RCCL (native collectives of the sharded search)
                for m in re.finditer(r"dlopen\(([^)]*)\)", src):
                    assert f == "fp_engine.cpp" and m.group(1).split(",")[0].strip() == "n", f"{f}: unexpected dlopen({m.group(1)})"
                if "dlopen" in src:
                    names = re.search(r"const char\* names\[\] = \{([^}]*)\}", src).group(1)
                    lits = re.findall(r'"([^"]*)"', names)
                    assert lits and all("rccl" in x for x in lits), lits


def test_device_string_parsing():
    from fast_plaid_amd.fast_plaid_rust import _device_id
    assert _device_id("cuda") == 0 and _device_id("cuda:3") == 3
    with pytest.raises(ValueError):
        _device_id("cuda:x")
    with pytest.raises(ValueError):
        _device_id("tpu")


def test_synth_twin_is_deterministic_and_consistent():
    import fast_plaid_amd as fp
    spec = fp.synth.SynthSpec(n_docs=400, doc_len=40, n_centroids=256, variable_len=True, seed=5)
    a = fp.synth.host_index_arrays(spec)
    b = fp.synth.host_index_arrays(spec)
    assert all(np.array_equal(a[k], b[k]) for k in ("doc_codes", "doc_residuals", "doc_lengths", "ivf", "ivf_lengths"))
    assert a["doc_lengths"].min() >= 10 and a["doc_lengths"].max() <= 40
    offs = np.concatenate([[0], np.cumsum(a["doc_lengths"])])
    sub = fp.synth.host_docs(spec, [7, 3, 399])
    assert np.array_equal(sub["doc_codes"][: a["doc_lengths"][7]], a["doc_codes"][offs[7]: offs[8]])
    assert np.array_equal(sub["doc_residuals"][-a["doc_lengths"][399]:], a["doc_residuals"][offs[399]: offs[400]])
    # IVF invariants: every (cell, pid) pair is a real token code, lists ascending unique
    ioff = np.concatenate([[0], np.cumsum(a["ivf_lengths"].astype(np.int64))])
    for c in range(0, 256, 37):
        lst = a["ivf"][ioff[c]: ioff[c + 1]]
        assert np.all(np.diff(lst) > 0)
        for p in lst[:5]:
            assert c in a["doc_codes"][offs[p]: offs[p + 1]]
    assert a["centroids"].dtype == np.float16 and np.allclose(np.linalg.norm(a["centroids"].astype(np.float32), axis=1), 1, atol=2e-3)


def test_index_directory_roundtrip(tmp_path):
    import fast_plaid_amd as fp
    from fast_plaid_amd.search import index_io
    spec = fp.synth.SynthSpec(n_docs=120, doc_len=30, n_centroids=64, variable_len=True)
    arr = fp.synth.host_index_arrays(spec)
    index_io.save_index_arrays(str(tmp_path), arr, chunk_docs=50)
    back = index_io.load_index_arrays(str(tmp_path))
    for k in ("doc_codes", "doc_residuals", "doc_lengths", "ivf", "ivf_lengths", "centroids", "bucket_weights"):
        assert np.array_equal(np.asarray(back[k]), np.asarray(arr[k])), k
    assert back["nbits"] == 4
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith(".codes.npy")) == ["0.codes.npy", "1.codes.npy", "2.codes.npy"]


def test_shard_planning_and_slicing():
    import fast_plaid_amd as fp
    from fast_plaid_amd import sharded
    spec = fp.synth.SynthSpec(n_docs=1000, doc_len=60, n_centroids=128, variable_len=True)
    arr = fp.synth.host_index_arrays(spec)
    for w in (1, 2, 3, 8):
        rs = sharded.plan_shards(arr["doc_lengths"], w)
        assert rs[0][0] == 0 and rs[-1][1] == 1000 and all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
        toks = [int(arr["doc_lengths"][b:e].sum()) for b, e in rs]
        assert max(toks) - min(toks) <= 2 * 60
    b, e = sharded.plan_shards(arr["doc_lengths"], 3)[1]
    sh = sharded.shard_arrays(arr, b, e)
    assert sh["doc_lengths"].shape[0] == e - b and sh["ivf"].max() < e - b
    assert int(sh["ivf_lengths"].sum()) == sh["ivf"].shape[0]
    assert sharded.plan_shards(10, 4) == [(0, 2), (2, 5), (5, 8), (8, 10)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_protocol_gloo(world):
    """world_size-2/3/4 gloo runs of fast_plaid_amd.sharded.sharded_search (document shards),
    replicated_search (batch split) and, with four ranks, the 2-D layout (2 document shards x 2 query groups: the sharded
    search over a sub-group, the batch split over the groups) with an oracle-backed engine standing in for the HIP stages
    (test infrastructure): the protocols (fixed-size all-gathers, global cut, merge, result
    gather) must reproduce the unsharded oracle exactly."""
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
    procs = []
    for rank in range(world):
        e = dict(env, RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gloo_shard_worker.py")], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert "SHARDED_OK" in o, o
        assert world < 4 or "GRID_OK" in o, o


def test_index_directory_matches_the_reference_loader():
    """tests/golden/refloader/index_dir was written by this repo's writer; index_dir_expected.npz is what the REFERENCE's own
    python/fast_plaid/search/load.py::_load_index_tensors_cpu returned for it (tests/golden/make_index_dir_golden.py).
    So (a) the writer emits a directory the reference reads, and (b) our reader returns the same construct_index
    argument set, bit for bit, up to the max_len - last_len padding rows the reference appends (load.py:298-320),
    which construct_index ignores."""
    from fast_plaid_amd.search import index_io
    gdir = os.path.join(ROOT, "tests", "golden", "refloader")
    exp = np.load(os.path.join(gdir, "index_dir_expected.npz"))
    got = index_io.load_index_arrays(os.path.join(gdir, "index_dir"))
    assert got["nbits"] == int(exp["nbits"])
    for k in ("centroids", "avg_residual", "bucket_cutoffs", "bucket_weights", "ivf", "ivf_lengths", "doc_lengths"):
        assert got[k].dtype == exp[k].dtype, k
        assert np.array_equal(got[k].view(np.uint16) if got[k].dtype == np.float16 else got[k],
                              exp[k].view(np.uint16) if exp[k].dtype == np.float16 else exp[k]), k
    T = int(got["doc_lengths"].sum())
    lens = got["doc_lengths"]
    pad = int(lens.max() - lens[-1])
    assert exp["doc_codes"].shape[0] == T + pad and exp["doc_residuals"].shape[0] == T + pad
    assert got["doc_codes"].shape[0] == T
    assert np.array_equal(got["doc_codes"], exp["doc_codes"][:T]) and np.array_equal(got["doc_residuals"], exp["doc_residuals"][:T])
    # the reference's loader leaves nothing behind in the committed directory (it ran on a scratch copy)
    assert not [f for f in os.listdir(os.path.join(gdir, "index_dir")) if f.startswith("merged_")]


def test_delete_from_index_matches_rebuild(tmp_path):
    """maintain.delete_from_index (rust/index/delete.rs:26-145) on a copy of the committed directory: what loads back
    equals the arrays of the surviving documents (renumbered by position), IVF rebuilt; pure host work."""
    import shutil
    from fast_plaid_amd import maintain, synth
    from fast_plaid_amd.search import index_io
    src = os.path.join(ROOT, "tests", "golden", "refloader", "index_dir")
    dst = str(tmp_path / "idx")
    shutil.copytree(src, dst)
    before = index_io.load_index_arrays(dst)
    drop = [0, 3, 49, 50, 51, 129, 129, 100000]          # first, a chunk boundary, last, a duplicate, an id that does not exist
    maintain.delete_from_index(dst, drop)
    after = index_io.load_index_arrays(dst)
    keep = np.array([d not in set(drop) for d in range(before["doc_lengths"].shape[0])])
    lens = before["doc_lengths"]
    keep_tok = np.repeat(keep, lens)
    assert np.array_equal(after["doc_lengths"], lens[keep])
    assert np.array_equal(after["doc_codes"], before["doc_codes"][keep_tok])
    assert np.array_equal(after["doc_residuals"], before["doc_residuals"][keep_tok])
    ivf, ivf_lengths = synth.build_ivf(after["doc_codes"], after["doc_lengths"], before["ivf_lengths"].shape[0])
    assert np.array_equal(after["ivf"], ivf) and np.array_equal(after["ivf_lengths"], ivf_lengths)
    meta = __import__("json").load(open(os.path.join(dst, "metadata.json")))
    assert meta["num_documents"] == int(keep.sum()) and meta["num_embeddings"] == int(keep_tok.sum()) and meta["num_chunks"] == 3
    for k in ("centroids", "bucket_weights", "bucket_cutoffs"):
        assert np.array_equal(after[k], before[k])


def test_heldout_sample_follows_the_shuffled_sample_not_the_corpus_tail():
    """create.rs:222-281: the held-out embeddings are the last 5 % (<= 50 000) tokens of the SHUFFLED passage sample, walked from
    its end -- whole passages, the first one cut to its tail -- not the tail of the corpus in input order."""
    import math
    from fast_plaid_amd import create as CR
    rng = np.random.default_rng(5)
    n, dim = 400, 8
    # every token carries its document id in dim 0 and its position in dim 1
    docs = []
    for i in range(n):
        ln = int(rng.integers(3, 30))
        d = np.zeros((ln, dim), np.float16)
        d[:, 0] = i
        d[:, 1] = np.arange(ln)
        docs.append(d)
    held = CR.heldout_sample(docs, np.random.default_rng(42))
    perm = np.random.default_rng(42).permutation(n)
    k = int(min(1.0 + 16.0 * math.sqrt(120.0 * n), n))
    sample = perm[:k]
    total = sum(docs[int(i)].shape[0] for i in sample)
    want_n = int(round(min(0.05 * total, 50_000.0)))
    assert held.shape == (want_n, dim)
    # the held-out rows are the tail of the sample's concatenation in SHUFFLED order
    cat = np.concatenate([docs[int(i)] for i in sample])
    assert np.array_equal(held, cat[-want_n:])
    ids = held[:, 0].astype(int)
    assert len(set(ids.tolist())) > 1 and not np.all(ids >= n - len(set(ids.tolist())) - 1), "held-out rows are the corpus tail"
    # documents appear whole except possibly the first, which is cut to its tail
    first = ids[0]
    assert held[0, 1] == docs[first].shape[0] - (ids == first).sum()


def test_cutoffs_rounded_down_reproduce_the_fp32_comparison():
    """create.rs:413 buckets fp16 residuals against fp32 cutoffs; the device kernel compares with fp16 cutoffs.  Rounding the cutoffs
    DOWN to fp16 gives the same buckets for every fp16 residual (exhaustive over all finite fp16 values)."""
    from fast_plaid_amd import create as CR
    rng = np.random.default_rng(0)
    cut32 = np.sort(rng.normal(0, 0.05, 15).astype(np.float32))
    cut16 = CR.cutoffs_for_f32_compare(cut32)
    allh = np.arange(65536, dtype=np.uint16).view(np.float16)
    allh = allh[np.isfinite(allh)]
    want = (cut32[None, :] < allh.astype(np.float32)[:, None]).sum(1)     # torch.bucketize(right=False) after promotion to fp32
    got = (cut16.astype(np.float32)[None, :] < allh.astype(np.float32)[:, None]).sum(1)
    assert np.array_equal(want, got)
    assert np.all(cut16.astype(np.float32) <= cut32)


def test_index_directory_oracle_roundtrip():
    """the ATen restatement of create / update / delete (oracle/plaid_index_oracle_torch.py) reproduces the committed snapshots
    (tests/golden/maintain/): guards the fixture and its generator against drift."""
    import json
    import shutil
    import tempfile
    import torch
    import plaid_index_oracle_torch as IO
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "maintain", "snapshots.npz"))

    def docs_of(name):
        lens = z[name + "_lens"]
        offs = np.concatenate([[0], np.cumsum(lens)])
        return [torch.from_numpy(z[name][offs[i]: offs[i + 1]]) for i in range(len(lens))]

    tmp = tempfile.mkdtemp()
    try:
        d = os.path.join(tmp, "ix")
        IO.create_index(docs_of("docs"), d, torch.from_numpy(z["centroids"]), int(z["nbits"]), torch.from_numpy(z["heldout"]), batch_size=25)
        IO.update_index(docs_of("new1"), d, batch_size=25, update_threshold=True)
        snap = IO.read_directory(d)
        for fn, v in snap.items():
            w = z["updated/" + fn]
            if fn.endswith(".npy"):
                assert np.array_equal(v, w), fn
            else:
                assert v == json.loads(bytes(w).decode()), fn
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_delete_and_metadata_follow_the_reference_writers():
    """Host-side halves of what tests/maintain_fuzz_worker.py found on the GPU box (round 6), checked here without a device against
    the ATen restatement run live: (1) maintain.delete_from_index on directories the oracle's create wrote -- plain and
    compress_only (delete.rs:105-143 rebuilds the lists and writes the metadata WITHOUT the compress_only key), duplicated and
    repeated positions -- leaves the files the oracle's delete leaves; (2) index_io.save_index_arrays writes the `num_partitions`
    it is handed (create.rs:572: the estimate 2^floor(log2(16 sqrt(tokens))), not the list count)."""
    import json
    import shutil
    import tempfile
    import torch
    import plaid_index_oracle_torch as IO
    from fast_plaid_amd import maintain as MT
    from fast_plaid_amd.search import index_io
    g = torch.Generator().manual_seed(5)
    tmp = tempfile.mkdtemp()
    try:
        for case, (n_docs, C, chunk, co) in enumerate(((40, 64, 7, False), (25, 16, 1000, True), (3, 8, 1, False), (60, 128, 25, True))):
            cent = torch.nn.functional.normalize(torch.randn(C, 32, generator=g), dim=-1).half()
            docs = [torch.nn.functional.normalize(torch.randn(int(torch.randint(1, 9, (1,), generator=g)), 32, generator=g), dim=-1).half()
                    for _ in range(n_docs)]
            held = torch.cat(docs)[:50]
            a, b = os.path.join(tmp, f"a{case}"), os.path.join(tmp, f"b{case}")
            IO.create_index(docs, a, cent, 4, held, batch_size=chunk, compress_only=co)
            shutil.copytree(a, b)
            sub = [0, n_docs - 1, n_docs // 2, n_docs // 2, 0]
            IO.delete_from_index(sub, a)
            MT.delete_from_index(b, sub)
            want = IO.read_directory(a)
            assert set(want) <= set(os.listdir(b)), sorted(set(want) - set(os.listdir(b)))
            for fn, w in want.items():
                if fn.endswith(".npy"):
                    got = np.load(os.path.join(b, fn))
                    assert got.shape == w.shape and (np.array_equal(got, w) if w.dtype.kind in "iu" else np.allclose(got, w, rtol=2e-6)), (case, fn)
                else:
                    with open(os.path.join(b, fn)) as f:
                        gj = json.load(f)
                    if isinstance(w, dict):
                        assert set(gj) == set(w), (case, fn, sorted(gj), sorted(w))     # (no compress_only key after a delete)
                        for k, v in w.items():
                            assert (abs(gj[k] - v) <= 1e-9 * max(1.0, abs(v))) if isinstance(v, float) else gj[k] == v, (case, fn, k)
                    else:
                        assert gj == w, (case, fn)
        # (2)
        arr = dict(nbits=4, centroids=np.zeros((8, 16), np.float16), bucket_cutoffs=np.zeros(15, np.float16), bucket_weights=np.zeros(16, np.float16),
                   avg_residual=np.zeros(16, np.float16), ivf=np.zeros(3, np.int64), ivf_lengths=np.array([1, 0, 2, 0, 0, 0, 0, 0], np.int32),
                   doc_codes=np.array([0, 2, 2], np.int64), doc_residuals=np.zeros((3, 8), np.uint8), doc_lengths=np.array([1, 2], np.int64),
                   num_partitions=4)
        d = os.path.join(tmp, "meta")
        index_io.save_index_arrays(d, arr)
        assert json.load(open(os.path.join(d, "metadata.json")))["num_partitions"] == 4
        del arr["num_partitions"]
        index_io.save_index_arrays(d, arr)
        assert json.load(open(os.path.join(d, "metadata.json")))["num_partitions"] == 8
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_comparators_admit_exact_ties_only_unless_told():
    """tests/parity.py: two documents may swap only where the REFERENCE's scores are exactly equal; with ulp_ties=1 (the fuzz worker's
    second look after a strict failure) also neighbours one fp32 ulp apart -- and nothing wider."""
    import parity as P
    s = np.array([20.47344017, 20.47343826, 20.47343826, 19.0], np.float32)
    assert float(s[0]) - float(s[1]) == float(np.spacing(np.float32(s[1])))          # one ulp apart
    P._same_order_modulo_exact_ties([1, 3, 2, 4], [1, 2, 3, 4], s)                    # exact tie at positions 1, 2
    with pytest.raises(AssertionError):
        P._same_order_modulo_exact_ties([2, 1, 3, 4], [1, 2, 3, 4], s)                # one ulp is not a tie ...
    P._same_order_modulo_exact_ties([2, 1, 3, 4], [1, 2, 3, 4], s, ulp_ties=1)        # ... unless told
    s2 = np.array([20.47344398, 20.47343826, 19.0, 18.0], np.float32)                 # three ulps apart
    with pytest.raises(AssertionError):
        P._same_order_modulo_exact_ties([2, 1, 3, 4], [1, 2, 3, 4], s2, ulp_ties=1)


def test_prepare_search_matches_the_reference_python():
    """tests/golden/pyboundary/prepare_search.npz holds what the REFERENCE's own FastPlaid._prepare_search (python/fast_plaid/search/
    fast_plaid.py:743-795, imported from /root/reference in the build container by tests/golden/pyboundary/make_prepare_search_golden.py)
    returned for seeded inputs: a list of 2-D / 3-D query tensors zero-padded to the longest, a 3-D tensor passed through, and
    every subset form (flat list, per-query lists, an int, the empty list, a list of the wrong length).  This repository's
    FastPlaid._prepare_search must hand the same padded queries (cast to fp16, as fast_plaid.py:241 does before the native
    boundary) and the same normalised subsets to the native search."""
    import types
    from fast_plaid_amd import search
    z = np.load(os.path.join(ROOT, "tests", "golden", "pyboundary", "prepare_search.npz"))
    me = types.SimpleNamespace(index=None, devices=["cpu"], indices={"cpu": object()})
    prep = lambda q, s: search.FastPlaid._prepare_search(me, q, s)   # noqa: E731
    qa = [z[f"a_in_{i}"] for i in range(4)]
    q3, sub = prep(qa, None)
    assert sub is None and q3.dtype == np.float16
    assert np.array_equal(q3, z["a_out"].astype(np.float16))
    q3, _ = prep(z["b_in"], None)
    assert np.array_equal(q3, z["b_out"].astype(np.float16))
    forms = {"flat": [7, 3, 3, 11], "perq": [[1], [2, 2, 5], [], [9, 8]], "int": 6, "empty": []}
    for name, sv in forms.items():
        _, sub = prep(qa, sv)
        assert (sub is None) == bool(z[f"sub_{name}_none"]), name
        if sub is not None:
            assert [len(x) for x in sub] == z[f"sub_{name}_lens"].tolist(), name
            assert [int(v) for x in sub for v in x] == z[f"sub_{name}_flat"].tolist(), name
    assert bool(z["sub_badlen_raises"])
    with pytest.raises(ValueError):
        prep(qa, [[1], [2]])
