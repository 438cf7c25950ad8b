"""Generates tests/golden/pyboundary/prepare_search.npz  (run in the build container only; reads /root/reference).

What is pinned: the Python-side normalisation in front of the native boundary ("Python caller to reproduce"):
the REFERENCE's own `FastPlaid._prepare_search` (python/fast_plaid/search/fast_plaid.py:743-795) -- zero-padding of a list of
2-D / 3-D query tensors to the longest (`pad_sequence`, :772-780) and the subset normalisation (:782-793: int -> per-query list,
flat list -> broadcast, [] -> None, length check) -- executed here on seeded inputs.  fast_plaid.py is imported from
/root/reference by file path inside a throw-away package; the modules it imports but `_prepare_search` never touches (the native
`fast_plaid_rust`, the `fastkmeans` / `usearch` users kmeans.py / update.py, the sqlite filtering) are satisfied by empty
placeholders.  The method runs on a stand-in `self` (index directory with a metadata.json, one loaded-index token per device).
Only its OUTPUTS are stored; nothing of the reference's source is copied."""
import importlib.util
import json
import os
import sys
import tempfile
import threading
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/python/fast_plaid/search/fast_plaid.py"


def reference_module():
    def placeholder(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    pkg = placeholder("fast_plaid")
    pkg.__path__ = []
    pkg.fast_plaid_rust = placeholder("fast_plaid.fast_plaid_rust")
    placeholder("fast_plaid.filtering", create=None, delete=None)
    sp = placeholder("fast_plaid.search")
    sp.__path__ = []
    placeholder("fast_plaid.search.kmeans", FastKMeans=None)
    placeholder("fast_plaid.search.load", _reload_index=None, save_list_tensors_on_disk=None)
    placeholder("fast_plaid.search.update", process_update=None)
    spec = importlib.util.spec_from_file_location("fast_plaid.search.fast_plaid", REF)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = "fast_plaid.search"
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = reference_module()
    g = torch.Generator().manual_seed(11)
    dim = 16
    cases = {}
    with tempfile.TemporaryDirectory() as tmp:
        json.dump({}, open(os.path.join(tmp, "metadata.json"), "w"))
        me = types.SimpleNamespace(index=tmp, devices=["cpu"], indices={"cpu": object()}, _index_swap_lock=threading.Lock(),
                                   _check_and_reload_index=lambda blocking=False: None)

        def run(q, subset):
            _, q3, sub = ref.FastPlaid._prepare_search(me, q, subset)
            return q3, sub

        # (a) list of 2-D tensors of different lengths, a 3-D [1, L, D] member among them
        qa = [torch.randn(5, dim, generator=g), torch.randn(1, 9, dim, generator=g), torch.randn(2, dim, generator=g), torch.randn(9, dim, generator=g)]
        # (b) a 3-D tensor goes through untouched
        qb = torch.randn(3, 4, dim, generator=g)
        out = {}
        q3, sub = run(qa, None)
        out["a_in_lens"] = np.array([5, 9, 2, 9])
        for i, t in enumerate(qa):
            out[f"a_in_{i}"] = t.numpy()
        out["a_out"] = q3.numpy()
        q3, sub = run(qb, None)
        out["b_in"] = qb.numpy()
        out["b_out"] = q3.numpy()
        # subset forms (on the 4-query list): flat list -> broadcast, per-query lists kept, int -> broadcast, [] -> None
        subs = {"flat": [7, 3, 3, 11], "perq": [[1], [2, 2, 5], [], [9, 8]], "int": 6, "empty": []}
        for name, sv in subs.items():
            _, sub = run(qa, sv)
            out[f"sub_{name}_none"] = np.array(sub is None)
            if sub is not None:
                out[f"sub_{name}_lens"] = np.array([len(x) if isinstance(x, list) else -1 for x in sub])
                out[f"sub_{name}_flat"] = np.array([v for x in sub for v in (x if isinstance(x, list) else [x])], dtype=np.int64)
        # a subset of the wrong length raises ValueError
        try:
            run(qa, [[1], [2]])
            out["sub_badlen_raises"] = np.array(False)
        except ValueError:
            out["sub_badlen_raises"] = np.array(True)
        cases = out
    np.savez_compressed(os.path.join(HERE, "prepare_search.npz"), **cases)
    print("wrote", os.path.join(HERE, "prepare_search.npz"), {k: (v.shape if hasattr(v, "shape") else v) for k, v in cases.items()})


if __name__ == "__main__":
    main()
