"""Generates tests/golden/refloader/index_dir/ and tests/golden/refloader/index_dir_expected.npz  (run in the build container only).

What is pinned: the on-disk index directory format.  The directory is written by THIS
repository's writer (fast_plaid_amd.search.index_io.save_index_arrays); the expected arrays are what the REFERENCE's own
loader, python/fast_plaid/search/load.py::_load_index_tensors_cpu (:220-322, the function whose output is handed to
construct_index), returns for that directory.  load.py is imported from /root/reference by file path; its top-level
`from fast_plaid import fast_plaid_rust` (the native module that cannot be built here) is satisfied by an empty
placeholder module -- _load_index_tensors_cpu never touches it.  The loader runs on a scratch copy because it writes
merged_*.npy caches next to the chunks.  Nothing of the reference's source is copied: only its outputs are stored.
"""
import importlib.util
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import plaid_oracle_torch as OT  # noqa: E402
from fast_plaid_amd.search import index_io  # noqa: E402

REF_LOAD = "/root/reference/python/fast_plaid/search/load.py"


def reference_loader():
    pkg = types.ModuleType("fast_plaid")
    pkg.fast_plaid_rust = types.ModuleType("fast_plaid.fast_plaid_rust")   # placeholder, never called
    sys.modules.setdefault("fast_plaid", pkg)
    sys.modules.setdefault("fast_plaid.fast_plaid_rust", pkg.fast_plaid_rust)
    spec = importlib.util.spec_from_file_location("_ref_load", REF_LOAD)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    g = torch.Generator().manual_seed(7)
    dim, nbits, C = 64, 4, 48
    cent = torch.nn.functional.normalize(torch.randn(C, dim, generator=g), dim=-1).to(torch.float16)
    lens = torch.randint(1, 40, (130,), generator=g).tolist()
    lens[-1] = 3      # last document shorter than the longest: the reference appends max_len - last_len padding rows
    docs = []
    for n in lens:
        pick = torch.randint(0, C, (n,), generator=g)
        d = cent[pick].float() + 0.25 * torch.randn(n, dim, generator=g) / dim ** 0.5
        docs.append(torch.nn.functional.normalize(d, dim=-1).to(torch.float16))
    arr = OT.build_index_arrays(docs, cent, nbits)
    arr = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in arr.items()}
    out_dir = os.path.join(HERE, "refloader", "index_dir")
    shutil.rmtree(out_dir, ignore_errors=True)
    index_io.save_index_arrays(out_dir, arr, chunk_docs=50)      # 3 chunks
    ref = reference_loader()
    with tempfile.TemporaryDirectory() as tmp:
        scratch = os.path.join(tmp, "idx")
        shutil.copytree(out_dir, scratch)
        data = ref._load_index_tensors_cpu(scratch)
        exp = {}
        for k, v in data.items():
            if v is None:
                continue
            exp[k] = np.array(v.numpy() if isinstance(v, torch.Tensor) else v)   # copy out of the mmap before the dir goes away
    np.savez_compressed(os.path.join(HERE, "refloader", "index_dir_expected.npz"), **exp)
    print({k: (v.shape, v.dtype) for k, v in exp.items()})
    print("files:", sorted(os.listdir(out_dir)))


if __name__ == "__main__":
    main()
