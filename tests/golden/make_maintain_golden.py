"""Generates tests/golden/maintain/snapshots.npz (run once in the build container; the fixture is committed).

Producer: oracle/plaid_index_oracle_torch.py -- the op-for-op ATen restatement of the reference's directory writers
(rust/index/create.rs, update.rs, delete.rs).  Three directory snapshots of one small corpus:

    created   create_index(60 documents, held-out sample given, chunks of 25 documents)
    updated   + update_index(15 new documents, update_threshold=True)      (the last chunk holds < 2000 documents: it is extended)
    updated2  + update_index(9 more documents, update_threshold=False)
    deleted   + delete_from_index([3, 17, 61, 70, 83])

Every .npy / .json of each snapshot is stored (json as text); the GPU test rebuilds the same directories with
fast-plaid_amd/create.py + maintain.py (device compression) and compares file by file.

    python tests/golden/make_maintain_golden.py
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import plaid_index_oracle_torch as IO  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "maintain", "snapshots.npz")


def snapshot(path, tag, store):
    for fn, v in IO.read_directory(path).items():
        if fn.endswith(".npy"):
            store[f"{tag}/{fn}"] = v
        else:
            store[f"{tag}/{fn}"] = np.frombuffer(json.dumps(v).encode(), dtype=np.uint8)


def main():
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(2024)
    dim, nbits, C = 128, 4, 64

    def mkdocs(n):
        lens = torch.randint(5, 41, (n,), generator=g).tolist()
        return [torch.nn.functional.normalize(torch.randn(l, dim, generator=g), dim=-1).half() for l in lens]

    docs = mkdocs(60)
    allt = torch.cat(docs)
    cent = torch.nn.functional.normalize(allt[torch.randperm(allt.shape[0], generator=g)[:C]].float(), dim=-1).half()
    heldout = torch.cat(docs[41:52])[-150:]
    new1, new2 = mkdocs(15), mkdocs(9)
    subset = [3, 17, 61, 70, 83]
    store = dict(dim=np.array(dim), nbits=np.array(nbits), centroids=cent.numpy(), heldout=heldout.numpy(), subset=np.array(subset, np.int64))
    for name, dl in (("docs", docs), ("new1", new1), ("new2", new2)):
        store[name] = torch.cat(dl).numpy()
        store[name + "_lens"] = np.array([d.shape[0] for d in dl], np.int64)
    tmp = tempfile.mkdtemp()
    try:
        d = os.path.join(tmp, "ix")
        IO.create_index(docs, d, cent, nbits, heldout, batch_size=25)
        snapshot(d, "created", store)
        IO.update_index(new1, d, batch_size=25, update_threshold=True)
        snapshot(d, "updated", store)
        IO.update_index(new2, d, batch_size=25, update_threshold=False)
        snapshot(d, "updated2", store)
        IO.delete_from_index(subset, d)
        snapshot(d, "deleted", store)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    np.savez_compressed(OUT, **store)
    print(OUT, os.path.getsize(OUT) // 1024, "KiB;", len(store), "entries")
    for tag in ("created", "updated", "updated2", "deleted"):
        m = json.loads(bytes(store[f"{tag}/metadata.json"]).decode())
        print(tag, m)


if __name__ == "__main__":
    main()
