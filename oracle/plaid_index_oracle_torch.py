"""ORACLE (test infrastructure; nothing under fast-plaid_amd/ imports this): op-for-op ATen (torch 2.10 CPU) restatement of the
reference's index-directory writers

    create_index        rust/index/create.rs:206-583   (given the held-out sample: the RNG-dependent sampling :222-281 is an input)
    update_index        rust/index/update.rs:30-473
    delete_from_index   rust/index/delete.rs:26-145

on top of the array-level helpers of plaid_oracle_torch.py (compress_into_codes, packbits, scalar_quantile_kthvalue, ...).
Each function cites the lines it follows.  Parity pinning: like the search oracle, this is pinned to the reference's arithmetic
dependency (the same ATen CPU kernels), not to reference-run outputs -- the reference is Rust + tch-rs and cannot be built here.
tests/golden/make_maintain_golden.py runs these functions and stores the resulting directories as fixtures (tests/golden/maintain/); the GPU tests compare
fast-plaid_amd/create.py and maintain.py with them array by array.
"""
from __future__ import annotations

import json
import math
import os
import re

import numpy as np
import torch

import plaid_oracle_torch as OT

DEFAULT_PROC_CHUNK_SIZE = 25_000   # update.rs:28


def _write_npy(path, t: torch.Tensor):
    np.save(path, t.detach().cpu().numpy())


def _read_json(p):
    with open(p) as f:
        return json.load(f)


def _process_batch(batch: torch.Tensor, centroids: torch.Tensor, cutoffs: torch.Tensor, nbits: int, batch_size: int, want_norms=False):
    """create.rs:404-428 / update.rs:122-176: codes, packed residual bytes, (fp32 residual norms)."""
    dim = centroids.shape[1]
    codes_l, packed_l, norms_l = [], [], []
    for micro in batch.split(batch_size, 0):                     # update.rs:136 (create.rs processes the whole batch at once: same values)
        codes = OT.compress_into_codes(micro, centroids)
        res = micro - centroids.index_select(0, codes)
        if want_norms:
            norms_l.append(res.to(torch.float32).norm(2, dim=1))   # update.rs:144-147
        # create.rs:413 hands bucketize the FLOAT cutoffs it just computed (ATen promotes the Half residuals: the comparison is made
        # in fp32 against unrounded cutoffs); update.rs:149 hands it the loaded codec's cutoffs, which load.py:255-258 cast to Half
        b = torch.bucketize(res, cutoffs, out_int32=True, right=False)
        b = b.unsqueeze(-1).expand(*b.shape, nbits)
        b = b.bitwise_right_shift(torch.arange(nbits, dtype=torch.int8))                # bit_helper
        b = b.bitwise_and(torch.ones_like(b))
        packed_l.append(OT.packbits(b.flatten()).reshape(micro.shape[0], dim // 8 * nbits))
        codes_l.append(codes)
    codes = torch.cat(codes_l) if codes_l else torch.empty((0,), dtype=torch.int64)
    packed = torch.cat(packed_l) if packed_l else torch.empty((0, dim // 8 * nbits), dtype=torch.uint8)
    norms = torch.cat(norms_l) if (want_norms and norms_l) else None
    return codes, packed, norms


def _compress_chunk(docs, centroids, cutoffs, nbits, batch_size, want_norms=False):
    """the document loop of create.rs:430-470 / update.rs:178-228: documents accumulated until `batch_size` rows, then one process_batch."""
    codes_l, res_l, norms_l, doclens = [], [], [], []
    acc, rows = [], 0
    for d in docs:
        doclens.append(int(d.shape[0]))
        acc.append(d.to(torch.float16))
        rows += int(d.shape[0])
        if rows >= batch_size:
            c, r, n = _process_batch(torch.cat(acc), centroids, cutoffs, nbits, batch_size, want_norms)
            acc, rows = [], 0
            codes_l.append(c); res_l.append(r)
            if n is not None:
                norms_l.append(n)
    if acc:
        c, r, n = _process_batch(torch.cat(acc), centroids, cutoffs, nbits, batch_size, want_norms)
        codes_l.append(c); res_l.append(r)
        if n is not None:
            norms_l.append(n)
    return torch.cat(codes_l), torch.cat(res_l), doclens, norms_l


def optimize_ivf(sorted_indices: torch.Tensor, code_counts: torch.Tensor, index_path: str):
    """create.rs:55-132: embedding ids -> passage ids through the doclens files, unique per list."""
    files = {}
    for fn in os.listdir(index_path):
        m = re.fullmatch(r"doclens\.(\d+)\.json", fn)
        if m:
            files[int(m.group(1))] = os.path.join(index_path, fn)
    all_doclens = []
    for k in sorted(files):
        all_doclens.extend(_read_json(files[k]))
    emb_to_pid = torch.repeat_interleave(torch.arange(len(all_doclens), dtype=torch.int64), torch.tensor(all_doclens, dtype=torch.int64))
    pids = emb_to_pid.index_select(0, sorted_indices)
    out, lens, off = [], [], 0
    for ln in code_counts.tolist():
        u = torch.unique(pids.narrow(0, off, ln), sorted=True)   # unique_dim(0, sorted=true)
        out.append(u)
        lens.append(int(u.shape[0]))
        off += ln
    ivf = torch.cat(out) if out else torch.empty((0,), dtype=torch.int64)
    return ivf, torch.tensor(lens, dtype=torch.int64)


def _rebuild_ivf(index_path, n_chunks, num_embeddings, num_partitions):
    """create.rs:528-559 == delete.rs:105-123."""
    all_codes = torch.zeros(num_embeddings, dtype=torch.int64)
    off = 0
    for i in range(n_chunks):
        c = torch.from_numpy(np.load(os.path.join(index_path, f"{i}.codes.npy")))
        all_codes[off: off + c.shape[0]] = c
        off += c.shape[0]
    sorted_codes, sorted_idx = all_codes.sort()
    counts = torch.bincount(sorted_codes, minlength=num_partitions)
    ivf, lens = optimize_ivf(sorted_idx, counts, index_path)
    _write_npy(os.path.join(index_path, "ivf.npy"), ivf.to(torch.int64))
    _write_npy(os.path.join(index_path, "ivf_lengths.npy"), lens.to(torch.int32))


def create_index(docs, index_path: str, centroids: torch.Tensor, nbits: int, heldout: torch.Tensor, batch_size: int = 25_000,
                 compress_only: bool = False):
    """create.rs:283-583 with the held-out sample given (the shuffle of :222-281 draws from Rust's StdRng)."""
    os.makedirs(index_path, exist_ok=True)
    centroids = centroids.to(torch.float16)
    n_docs = len(docs)
    n_chunks = math.ceil(n_docs / min(batch_size, 1 + n_docs))                       # :220
    avg_doc_len = sum(int(d.shape[0]) for d in docs) / n_docs
    est = int(2 ** math.floor(math.log2(16.0 * math.sqrt(n_docs * avg_doc_len))))    # :292-294
    with open(os.path.join(index_path, "plan.json"), "w") as f:                      # :296-299
        f.write(json.dumps({"nbits": nbits, "num_chunks": n_chunks}, indent=2) + "\n")
    held = heldout.to(torch.float16)
    codes = OT.compress_into_codes(held, centroids)                                  # :317
    res = (held - centroids.index_select(0, codes)).to(torch.float32)                # :325-326
    thr = OT.scalar_quantile_kthvalue(res.norm(2, dim=1), 0.75)                      # :331-334
    _write_npy(os.path.join(index_path, "cluster_threshold.npy"), thr.reshape(()))
    avg = res.abs().mean(0)                                                          # :341-344
    flat = res.flatten()
    n_opt = 2 ** nbits
    cut = torch.cat([OT.scalar_quantile_kthvalue(flat, i / n_opt) for i in range(1, n_opt)])            # :351-356
    wts = torch.cat([OT.scalar_quantile_kthvalue(flat, (i + 0.5) / n_opt) for i in range(n_opt)])       # :358-363
    _write_npy(os.path.join(index_path, "centroids.npy"), centroids)                 # :380-397
    _write_npy(os.path.join(index_path, "bucket_cutoffs.npy"), cut)
    _write_npy(os.path.join(index_path, "bucket_weights.npy"), wts)
    _write_npy(os.path.join(index_path, "avg_residual.npy"), avg)                    # ResidualCodec keeps avg_residual as given (Float)
    chunk = min(batch_size, 1 + n_docs)
    off = 0
    for i in range(n_chunks):                                                        # :430-497
        part = docs[i * chunk: min((i + 1) * chunk, n_docs)]
        c, r, dl, _ = _compress_chunk(part, centroids, cut, nbits, batch_size)
        _write_npy(os.path.join(index_path, f"{i}.codes.npy"), c)
        _write_npy(os.path.join(index_path, f"{i}.residuals.npy"), r)
        with open(os.path.join(index_path, f"doclens.{i}.json"), "w") as f:
            json.dump(dl, f)
        with open(os.path.join(index_path, f"{i}.metadata.json"), "w") as f:          # :500-524 (offset added in a second pass)
            json.dump({"num_documents": len(dl), "num_embeddings": int(c.shape[0]), "embedding_offset": off}, f, indent=2)
        off += int(c.shape[0])
    if not compress_only:
        _rebuild_ivf(index_path, n_chunks, off, est)
    with open(os.path.join(index_path, "metadata.json"), "w") as f:                  # :561-581
        json.dump({"num_chunks": n_chunks, "nbits": nbits, "num_partitions": est, "num_embeddings": off, "avg_doclen": off / n_docs,
                   "num_documents": n_docs, "compress_only": compress_only}, f, indent=2)


def update_index(docs, index_path: str, batch_size: int = 25_000, update_threshold: bool = False):
    """update.rs:30-473.  The loaded index it reads (codec, old IVF) is what load.py would hand over: the directory's own files."""
    meta = _read_json(os.path.join(index_path, "metadata.json"))
    n_existing = int(meta["num_chunks"])
    ivf_lengths_path = os.path.join(index_path, "ivf_lengths.npy")
    old_num_documents = int(meta["num_documents"])                                   # :51-54
    est = int(meta["num_partitions"])
    old_total = int(meta.get("num_embeddings", 0))
    compress_only = bool(meta.get("compress_only", False))
    nbits = int(meta["nbits"])
    centroids = torch.from_numpy(np.load(os.path.join(index_path, "centroids.npy"))).to(torch.float16)
    cutoffs = torch.from_numpy(np.load(os.path.join(index_path, "bucket_cutoffs.npy"))).to(torch.float16)   # load.py casts the codec to Half
    start, append_to_last, emb_off = n_existing, False, old_total                   # :75-108
    if start > 0:
        lp = os.path.join(index_path, f"{start - 1}.metadata.json")
        if os.path.exists(lp):
            lm = _read_json(lp)
            nd = lm.get("num_documents")
            if nd is not None and nd < 2000:
                start, append_to_last = start - 1, True
                emb_off = int(lm["embedding_offset"]) if "embedding_offset" in lm else old_total - int(lm.get("num_embeddings", 0))
    n_new = len(docs)
    chunk = min(DEFAULT_PROC_CHUNK_SIZE, 1 + n_new)                                  # :112-113
    n_new_chunks = math.ceil(n_new / chunk)
    new_codes_acc, new_doclens_acc, all_norms = [], [], []
    for i in range(n_new_chunks):                                                    # :178-279
        g = start + i
        part = docs[i * chunk: min((i + 1) * chunk, n_new)]
        c, r, dl, norms = _compress_chunk(part, centroids, cutoffs, nbits, batch_size, want_norms=update_threshold)
        all_norms.extend(norms)
        new_codes_acc.append(c)
        new_doclens_acc.extend(dl)
        if i == 0 and append_to_last and os.path.exists(os.path.join(index_path, f"{g}.codes.npy")):
            c = torch.cat([torch.from_numpy(np.load(os.path.join(index_path, f"{g}.codes.npy"))), c])
            r = torch.cat([torch.from_numpy(np.load(os.path.join(index_path, f"{g}.residuals.npy"))), r])
            dl = list(_read_json(os.path.join(index_path, f"doclens.{g}.json"))) + dl
        _write_npy(os.path.join(index_path, f"{g}.codes.npy"), c)
        _write_npy(os.path.join(index_path, f"{g}.residuals.npy"), r)
        with open(os.path.join(index_path, f"doclens.{g}.json"), "w") as f:
            json.dump(dl, f)
        with open(os.path.join(index_path, f"{g}.metadata.json"), "w") as f:
            json.dump({"num_documents": len(dl), "num_embeddings": int(c.shape[0])}, f)
    if update_threshold and all_norms:                                               # :282-306
        new_norms = torch.cat(all_norms)
        new_count = int(new_norms.shape[0])
        new_thr = float(OT.scalar_quantile_kthvalue(new_norms, 0.75))
        tp = os.path.join(index_path, "cluster_threshold.npy")
        if os.path.exists(tp):
            old_thr = float(np.load(tp))
            final = (old_thr * old_total + new_thr * new_count) / (old_total + new_count)
        else:
            final = new_thr
        np.save(tp, np.asarray(final, dtype=np.float64))                             # Tensor::from(f64) -> a Double scalar
    total_chunks = start + n_new_chunks
    for k in range(start, total_chunks):                                             # :310-322 embedding offsets
        mp = os.path.join(index_path, f"{k}.metadata.json")
        m = _read_json(mp)
        m["embedding_offset"] = emb_off
        emb_off += int(m["num_embeddings"])
        with open(mp, "w") as f:
            json.dump(m, f, indent=2)
    if not compress_only:                                                            # :325-444 partial IVF merged into the old lists
        new_codes = torch.cat(new_codes_acc).tolist()
        part_map = {}
        pid, ci = old_num_documents, 0
        for dl in new_doclens_acc:
            for _ in range(dl):
                part_map.setdefault(new_codes[ci], []).append(pid)
                ci += 1
            pid += 1
        new_part = {c: sorted(set(p)) for c, p in part_map.items() if 0 <= c < est}
        old_ivf = torch.from_numpy(np.load(os.path.join(index_path, "ivf.npy"))).to(torch.int64)
        old_len = torch.from_numpy(np.load(ivf_lengths_path)).to(torch.int64).tolist()
        if not new_part:
            return                                                                   # :371-373 (the reference returns before the metadata update)
        offs, cur = [], 0
        for l in old_len:
            offs.append(cur)
            cur += l
        parts, lens = [], []
        for i in range(est):
            ol = old_len[i] if i < len(old_len) else 0
            if ol > 0:
                parts.append(old_ivf.narrow(0, offs[i], ol))
            if i in new_part:
                parts.append(torch.tensor(new_part[i], dtype=torch.int64))
                lens.append(ol + len(new_part[i]))
            else:
                lens.append(ol)
        _write_npy(os.path.join(index_path, "ivf.npy"), torch.cat(parts).to(torch.int64))
        _write_npy(ivf_lengths_path, torch.tensor(lens, dtype=torch.int32))
    new_tokens = sum(new_doclens_acc)                                                # :447-471
    total_docs = old_num_documents + n_new
    old_avg = float(meta.get("avg_doclen", 0.0))
    with open(os.path.join(index_path, "metadata.json"), "w") as f:
        json.dump({"num_chunks": total_chunks, "nbits": nbits, "num_partitions": est, "num_embeddings": old_total + new_tokens,
                   "num_documents": total_docs, "avg_doclen": (old_avg * old_num_documents + new_tokens) / total_docs if total_docs else 0.0,
                   "compress_only": compress_only}, f, indent=2)


def delete_from_index(subset, index_path: str):
    """delete.rs:26-145: documents addressed by position; survivors renumbered; IVF rebuilt from scratch."""
    meta = _read_json(os.path.join(index_path, "metadata.json"))
    n_chunks, nbits, est = int(meta["num_chunks"]), int(meta["nbits"]), int(meta["num_partitions"])
    drop = set(int(x) for x in subset)
    final_docs, doc0, num_emb = 0, 0, 0
    for i in range(n_chunks):
        dlp = os.path.join(index_path, f"doclens.{i}.json")
        doclens = list(_read_json(dlp))
        new_doclens, mask = [], []
        for j, ln in enumerate(doclens):
            keep = (doc0 + j) not in drop
            if keep:
                new_doclens.append(ln)
            mask.extend([keep] * ln)
        final_docs += len(new_doclens)
        if len(new_doclens) < len(doclens):
            with open(dlp, "w") as f:
                json.dump(new_doclens, f)
            m = torch.tensor(mask, dtype=torch.bool)
            cp = os.path.join(index_path, f"{i}.codes.npy")
            codes = torch.from_numpy(np.load(cp))
            new_codes = codes.masked_select(m)
            _write_npy(cp, new_codes)
            rp = os.path.join(index_path, f"{i}.residuals.npy")
            res = torch.from_numpy(np.load(rp))
            _write_npy(rp, res.masked_select(m.unsqueeze(-1)).reshape(-1, res.shape[1]))
            mp = os.path.join(index_path, f"{i}.metadata.json")
            cm = _read_json(mp)
            cm["num_documents"] = len(new_doclens)
            cm["num_embeddings"] = int(new_codes.shape[0])
            with open(mp, "w") as f:
                json.dump(cm, f, indent=2)
        num_emb += sum(new_doclens)
        doc0 += len(doclens)
    _rebuild_ivf(index_path, n_chunks, num_emb, est)
    with open(os.path.join(index_path, "metadata.json"), "w") as f:
        json.dump({"num_chunks": n_chunks, "nbits": nbits, "num_partitions": est, "num_embeddings": num_emb,
                   "avg_doclen": (num_emb / final_docs) if final_docs else 0.0, "num_documents": final_docs}, f, indent=2)


def read_directory(index_path: str) -> dict:
    """every array / json of an index directory as numpy / python values (for fixtures and comparisons)."""
    out = {}
    for fn in sorted(os.listdir(index_path)):
        p = os.path.join(index_path, fn)
        if fn.endswith(".npy"):
            out[fn] = np.load(p)
        elif fn.endswith(".json"):
            out[fn] = _read_json(p)
    return out
