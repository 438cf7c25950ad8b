"""ORACLE (test infrastructure, never shipped, never timed as the product).

Op-for-op torch restatement of the fast-plaid *search* hot path.

Why torch: every arithmetic operation of the reference's hot path is a libtorch
ATen op called through the third-party crate ``tch = "0.20.0"`` (reference
``Cargo.toml:14``; wheel variant ``1.4.6.2100`` pins torch 2.10.0,
``ci-2100.toml:6``).  That dependency is NOT under /root/reference, and the
reference itself (Rust + PyO3) can neither be built nor imported in this image
(no cargo/rustc/maturin, no wheel).  tch-rs methods map 1:1 onto ``torch.*`` ops,
and torch 2.10.0 is installed here, so the functions below execute the same ATen
CPU kernels the reference's ``device="cpu"`` path executes, in the same order,
with the same dtypes.  Each function cites the reference lines it restates.

PARITY PINNING STATUS: the reference's own tests (tests/test.py) hold no golden
vectors, no seeded inputs and no known-answer values for this path (SURVEY.md
section 4 / 8c) and the reference cannot be executed here, so numeric parity is
"unpinned" against reference *outputs*; it is pinned (a) to the reference's
arithmetic dependency (ATen 2.10.0 CPU kernels) through this file and (b) to the
structural properties tests/test.py asserts (counts, ordering, determinism,
subset membership), which tests/ re-express.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch

# --------------------------------------------------------------------------- #
# ResidualCodec::load  (rust/utils/residual_codec.rs:72-152)
# --------------------------------------------------------------------------- #


def byte_reversed_bits_map(nbits: int) -> torch.Tensor:
    """256-entry uint8 table: each nbits-wide segment of the byte (MSB first) is
    bit-reversed in place.  residual_codec.rs:83-114."""
    out_tab = [0] * 256
    nbits_mask = (1 << nbits) - 1
    for i in range(256):
        val = i
        out = 0
        pos = 8
        while pos >= nbits:
            segment = (val >> (pos - nbits)) & nbits_mask
            rev_segment = 0
            for k in range(nbits):
                if segment & (1 << k):
                    rev_segment |= 1 << (nbits - 1 - k)
            out |= rev_segment
            if pos > nbits:
                out <<= nbits
            pos -= nbits
        out_tab[i] = out & 0xFF
    return torch.tensor(out_tab, dtype=torch.uint8)


def bucket_weight_indices_lookup(nbits: int) -> torch.Tensor:
    """[256, 8/nbits] int64: byte -> bucket indices, most-significant segment first.
    residual_codec.rs:117-140."""
    keys_per_byte = 8 // nbits
    mask = (1 << nbits) - 1
    rows = []
    for byte_val in range(256):
        row = []
        for k in reversed(range(keys_per_byte)):
            row.append((byte_val >> (k * nbits)) & mask)
        rows.append(row)
    return torch.tensor(rows, dtype=torch.int64)


# --------------------------------------------------------------------------- #
# decompress_residuals  (rust/search/search.rs:53-107)
# --------------------------------------------------------------------------- #


def decompress_residuals(
    packed_residuals: torch.Tensor,  # [n, D*nbits/8] uint8
    bucket_weights: torch.Tensor,  # [2^nbits] f16
    rev_map: torch.Tensor,  # [256] u8
    idx_lookup: torch.Tensor,  # [256, 8/nbits] i64
    codes: torch.Tensor,  # [n] i64
    centroids: torch.Tensor,  # [C, D] f16
    dim: int,
    nbits: int,
) -> torch.Tensor:
    n = codes.shape[0]
    packed_dim = (dim * nbits) // 8
    per_byte = 8 // nbits
    retrieved = centroids.index_select(0, codes)  # :70
    reshaped_centroids = retrieved.view(n, packed_dim, per_byte)  # :71-72
    flat_idx = packed_residuals.flatten().to(torch.int32)  # :75
    flat_rev = rev_map.index_select(0, flat_idx).to(torch.uint8)  # :76-78
    rev = flat_rev.view(n, packed_dim)  # :79
    sel = idx_lookup.index_select(0, rev.flatten().to(torch.int32)).to(torch.uint8)  # :83-85
    sel = sel.view(n, packed_dim, per_byte)  # :86-87
    gathered = bucket_weights.index_select(0, sel.flatten().to(torch.int32))  # :91-92
    gathered = gathered.view(n, packed_dim, per_byte)  # :93-94
    out = (gathered + reshaped_centroids).view(n, dim)  # :97-99
    norms = out.norm(p=2.0, dim=-1, keepdim=True).clamp_min(1e-12)  # :101-103
    return out / norms  # :105


# --------------------------------------------------------------------------- #
# StridedTensor  (rust/search/tensor.rs:203-355) -- semantics only: a CSR store
# whose lookup(ids) returns the rows of the requested elements concatenated in
# the order of `ids`, plus their lengths.  The as_strided window/mask machinery
# (:64-123, :154-191, :332-349) is a gather implementation detail with no
# numerical effect; trailing padding rows of `data` are never returned.
# --------------------------------------------------------------------------- #


class StridedTensor:
    def __init__(self, data: torch.Tensor, lengths: torch.Tensor):
        self.data = data
        self.lengths = lengths.to(torch.int64)  # :213
        zero = torch.zeros(1, dtype=torch.int64)
        self.offsets = torch.cat([zero, self.lengths.cumsum(0)])  # :221-224

    def lookup(self, ids: torch.Tensor):
        ids = ids.to(torch.int64)
        if ids.numel() == 0:  # :305-312
            shape = (0,) + tuple(self.data.shape[1:])
            return self.data.new_empty(shape), self.lengths.new_empty((0,))
        sel_len = self.lengths.index_select(0, ids)  # :314
        sel_off = self.offsets.index_select(0, ids)  # :315
        total = int(sel_len.sum())
        if total == 0:
            shape = (0,) + tuple(self.data.shape[1:])
            return self.data.new_empty(shape), sel_len
        # row r of the output belongs to element e(r); position inside = r - start(e)
        elem = torch.repeat_interleave(torch.arange(ids.numel()), sel_len)
        starts = torch.cat([torch.zeros(1, dtype=torch.int64), sel_len.cumsum(0)[:-1]])
        pos = torch.arange(total) - starts.index_select(0, elem)
        src = sel_off.index_select(0, elem) + pos
        return self.data.index_select(0, src), sel_len  # :346-355


# --------------------------------------------------------------------------- #
# direct_pad_sequences  (rust/search/padding.rs:61-109)
# --------------------------------------------------------------------------- #


def direct_pad_sequences(sequences: torch.Tensor, lengths: torch.Tensor, pad_value: float):
    if lengths.numel() == 0:  # :67-72
        return (
            sequences.new_empty((0, 0, sequences.shape[1])),
            torch.empty((0, 0), dtype=torch.bool),
        )
    batch = lengths.shape[0]
    feat = sequences.shape[1]
    max_len = int(lengths.max())  # :77-78
    padded = torch.full((batch, max_len, feat), pad_value, dtype=sequences.dtype)  # :80-87
    mask = torch.arange(max_len).unsqueeze(0) < lengths.unsqueeze(-1)  # :90-93
    nz = mask.nonzero()  # :96
    padded.index_put_((nz[:, 0], nz[:, 1]), sequences, accumulate=False)  # :102-106
    return padded, mask


# --------------------------------------------------------------------------- #
# colbert_score_reduce  (rust/search/search.rs:385-402)
# --------------------------------------------------------------------------- #


def colbert_score_reduce(token_scores: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
    expanded = attention_mask.unsqueeze(-1).expand(token_scores.shape)  # :389
    padding_mask = expanded.logical_not()  # :392
    masked = token_scores.masked_fill(padding_mask, -9999.0)  # :395
    max_per_tok, _ = masked.max(dim=1)  # :398
    return max_per_tok.sum(dim=-1, dtype=torch.float32)  # :401


# --------------------------------------------------------------------------- #
# subset helpers  (rust/search/search.rs:407-439)
# --------------------------------------------------------------------------- #


def intersect_sorted_unique(t1: torch.Tensor, t2: torch.Tensor) -> torch.Tensor:
    if t1.numel() == 0 or t2.numel() == 0:  # :408-410
        return t1.new_empty((0,))
    cat = torch.cat([t1, t2])  # :412
    srt, _ = cat.sort()  # :413
    size = srt.shape[0]
    if size < 2:
        return t1.new_empty((0,))
    dup = srt[: size - 1] == srt[1:]  # :420-422
    return srt[1:].masked_select(dup)  # :424-426


def filter_passage_ids_with_subset(pids: torch.Tensor, subset: torch.Tensor) -> torch.Tensor:
    if subset.numel() == 0 or pids.numel() == 0:  # :431-433
        return torch.empty((0,), dtype=torch.int64)
    s, _ = subset.sort()  # :435
    u = torch.unique_consecutive(s)  # :436
    return intersect_sorted_unique(pids, u)  # :438


# --------------------------------------------------------------------------- #
# LoadedIndex / construct_index  (rust/search/load.rs:124-186)
# --------------------------------------------------------------------------- #


@dataclass
class LoadedIndex:
    nbits: int
    centroids: torch.Tensor  # f16 [C, D]
    bucket_weights: torch.Tensor  # f16 [2^nbits]
    rev_map: torch.Tensor
    idx_lookup: torch.Tensor
    ivf: StridedTensor | None
    doc_codes: StridedTensor
    doc_residuals: StridedTensor


def construct_index(
    nbits, centroids, avg_residual, bucket_cutoffs, bucket_weights, ivf, ivf_lengths,
    doc_codes, doc_residuals, doc_lengths,
) -> LoadedIndex:
    """load.rs:124-186 (dtype normalisation :146-172); avg_residual / bucket_cutoffs
    are carried by the reference codec but unused by search."""
    del avg_residual, bucket_cutoffs
    ivf_st = None
    if ivf is not None and ivf_lengths is not None:
        ivf_st = StridedTensor(ivf.to(torch.int64), ivf_lengths.to(torch.int32))
    lens = doc_lengths.to(torch.int64)
    return LoadedIndex(
        nbits=int(nbits),
        centroids=centroids.to(torch.float16),
        bucket_weights=bucket_weights.to(torch.float16),
        rev_map=byte_reversed_bits_map(int(nbits)),
        idx_lookup=bucket_weight_indices_lookup(int(nbits)),
        ivf=ivf_st,
        doc_codes=StridedTensor(doc_codes.to(torch.int64), lens),
        doc_residuals=StridedTensor(doc_residuals.to(torch.uint8), lens),
    )


# --------------------------------------------------------------------------- #
# search  (rust/search/search.rs:471-696)
# --------------------------------------------------------------------------- #


@dataclass
class SearchTrace:
    """Stage outputs captured for stage-level parity tests."""
    centroid_scores: torch.Tensor | None = None  # [C, Q] f16        (S1)
    cells: torch.Tensor | None = None  # sorted unique probed cells  (S2)
    candidates: torch.Tensor | None = None  # ascending unique pids  (S3)
    approx_scores: torch.Tensor | None = None  # [n_cand] f32        (S4)
    rerank_pids: torch.Tensor | None = None  # pids exact-scored     (S5)
    exact_scores: torch.Tensor | None = None  # [n_rerank] f32       (S7)
    pids: list = field(default_factory=list)  # final                (S8)
    scores: list = field(default_factory=list)
    token_matrices: list | None = None


@torch.no_grad()
def search(
    query: torch.Tensor,  # [Q, D] f16
    index: LoadedIndex,
    n_ivf_probe: int,
    batch_size: int,
    n_full_scores: int,
    top_k: int,
    subset: torch.Tensor | None = None,
    return_token_scores: bool = False,
) -> SearchTrace:
    tr = SearchTrace()
    dim = query.shape[1]
    q_unsq = query.unsqueeze(0)  # :488
    scores_cq = index.centroids.matmul(query.transpose(0, 1))  # :491
    tr.centroid_scores = scores_cq

    if subset is not None:  # :494-517
        sub_codes, _ = index.doc_codes.lookup(subset)
        if sub_codes.numel() == 0:
            flat_cells = torch.empty((0,), dtype=torch.int64)
        else:
            uniq_cent = torch.unique(sub_codes.flatten(), sorted=True)  # :501-503
            sub_scores = scores_cq.index_select(0, uniq_cent)  # :505
            actual_k = min(n_ivf_probe, uniq_cent.shape[0])  # :506-507
            if actual_k == 1:
                top_local = sub_scores.argmax(dim=0, keepdim=True)  # :510
            else:
                top_local = sub_scores.topk(actual_k, dim=0, largest=True, sorted=False).indices
            flat_cells = uniq_cent.index_select(0, top_local.flatten())  # :515-516
    else:  # :519-529
        if n_ivf_probe == 1:
            sel = scores_cq.argmax(dim=0, keepdim=True).permute(1, 0)
        else:
            sel = scores_cq.topk(n_ivf_probe, dim=0, largest=True, sorted=False).indices.permute(1, 0)
        flat_cells = sel.flatten().contiguous()

    cells = torch.unique(flat_cells, sorted=True)  # :531-532
    tr.cells = cells

    pids_ivf, _ = index.ivf.lookup(cells)  # :535-536
    srt, _ = pids_ivf.sort()  # :538
    uniq_pids = torch.unique_consecutive(srt)  # :540-541
    if subset is not None:  # :544-547
        uniq_pids = filter_passage_ids_with_subset(uniq_pids, subset)
    tr.candidates = uniq_pids
    if uniq_pids.numel() == 0:  # :549-551
        return tr

    chunks = []
    total = uniq_pids.shape[0]
    n_batches = (total + batch_size - 1) // batch_size  # :556
    for step in range(n_batches):  # :558-586
        b0 = step * batch_size
        b1 = min((step + 1) * batch_size, total)
        if b0 >= b1:
            continue
        bp = uniq_pids.narrow(0, b0, b1 - b0)
        bcodes, blens = index.doc_codes.lookup(bp)
        if bcodes.numel() == 0:  # :570-576
            chunks.append(torch.zeros(bp.shape[0], dtype=torch.float32))
            continue
        bscores = scores_cq.index_select(0, bcodes)  # :578
        padded, mask = direct_pad_sequences(bscores, blens, 0.0)  # :580-581
        chunks.append(colbert_score_reduce(padded, mask))  # :583
    approx = torch.cat(chunks) if chunks else torch.empty((0,), dtype=torch.float32)  # :588-592
    tr.approx_scores = approx

    rerank = uniq_pids  # :602
    if n_full_scores < approx.shape[0] and approx.numel() > 0:  # :605-611
        top_s, top_i = approx.topk(n_full_scores, dim=0, largest=True, sorted=True)
        rerank = rerank.index_select(0, top_i)
        approx = top_s
    n_dec = max(n_full_scores // 4, 1)  # :614
    if n_dec < approx.shape[0] and approx.numel() > 0:  # :615-619
        _, top_i = approx.topk(n_dec, dim=0, largest=True, sorted=True)
        rerank = rerank.index_select(0, top_i)
    tr.rerank_pids = rerank
    if rerank.numel() == 0:  # :621-623
        return tr

    fcodes, flens = index.doc_codes.lookup(rerank)  # :626-627
    fres, _ = index.doc_residuals.lookup(rerank)  # :629
    emb = decompress_residuals(  # :640-649
        fres, index.bucket_weights, index.rev_map, index.idx_lookup, fcodes,
        index.centroids, dim, index.nbits,
    )
    padded, mask = direct_pad_sequences(emb, flens, 0.0)  # :651-652
    tok3d = padded.matmul(q_unsq.transpose(-2, -1))  # :654-655
    reduced = colbert_score_reduce(tok3d, mask)  # :656
    tr.exact_scores = reduced
    srt_scores, srt_idx = reduced.sort(dim=0, descending=True)  # :659
    srt_pids = rerank.index_select(0, srt_idx)  # :661
    count = min(top_k, srt_pids.shape[0])  # :666
    if return_token_scores:  # :668-686
        st = tok3d.index_select(0, srt_idx)
        sl = flens.index_select(0, srt_idx).tolist()
        tr.token_matrices = [st[i].narrow(0, 0, sl[i]).transpose(0, 1) for i in range(count)]
    tr.pids = srt_pids[:count].tolist()  # :688-692
    tr.scores = srt_scores[:count].tolist()
    return tr


def search_many(queries, index, n_ivf_probe, batch_size, n_full_scores, top_k, subset=None):
    """search.rs:219-288 -- sequential loop, per-query failures become empty results."""
    if index.ivf is None:  # :227-232
        raise ValueError(
            "This index was built with compress_only=True and does not support search. "
            "Rebuild with compress_only=False to enable search."
        )
    if queries.dim() != 3:  # :234-239
        raise ValueError(f"Expected a 3D tensor for queries, but got shape {list(queries.shape)}")
    out = []
    for qi in range(queries.shape[0]):
        sub = None
        if subset is not None and qi < len(subset):
            sub = torch.tensor(subset[qi], dtype=torch.int64)
        try:
            tr = search(queries[qi], index, n_ivf_probe, batch_size, n_full_scores, top_k, sub)
            out.append((tr.pids, tr.scores))
        except Exception:  # .unwrap_or_default()  :268
            out.append(([], []))
    return out


# --------------------------------------------------------------------------- #
# Index construction following rust/index/create.rs (format definition only:
# used to make *meaningful* compressed fixtures; k-means is third-party
# (fastkmeans 0.5.0) and out of scope -- centroids are an input here).
# --------------------------------------------------------------------------- #


def scalar_quantile_kthvalue(t: torch.Tensor, q: float) -> torch.Tensor:
    """rust/search/tensor.rs:18-34."""
    n = t.shape[0]
    idx_float = q * (n - 1)
    lo = math.floor(idx_float)
    hi = math.ceil(idx_float)
    if lo == hi:
        return t.kthvalue(lo + 1, 0, True).values
    lv = t.kthvalue(lo + 1, 0, True).values
    hv = t.kthvalue(hi + 1, 0, True).values
    return lv.lerp(hv, idx_float - lo)


def compress_into_codes(emb: torch.Tensor, centroids: torch.Tensor) -> torch.Tensor:
    """create.rs:148-170 (chunks of 2048, matmul + argmax)."""
    ct = centroids.transpose(0, 1)
    out = []
    for s in range(0, emb.shape[0], 2048):
        out.append(emb[s : s + 2048].matmul(ct).argmax(1))
    return torch.cat(out) if out else torch.empty((0,), dtype=torch.int64)


def packbits(bits: torch.Tensor) -> torch.Tensor:
    """create.rs:176-184 (big-endian within each byte)."""
    m = bits.reshape(-1, 8).to(torch.float16)
    w = torch.tensor([128, 64, 32, 16, 8, 4, 2, 1], dtype=torch.float16)
    return m.matmul(w).to(torch.uint8)


def train_codec(heldout: torch.Tensor, centroids: torch.Tensor, nbits: int):
    """create.rs:317-364: bucket cutoffs / weights = quantiles of held-out residuals."""
    codes = compress_into_codes(heldout, centroids)
    res = (heldout - centroids.index_select(0, codes)).to(torch.float32)
    flat = res.flatten()
    n_opt = 2 ** nbits
    cut = torch.cat([scalar_quantile_kthvalue(flat, i / n_opt) for i in range(1, n_opt)])
    wts = torch.cat([scalar_quantile_kthvalue(flat, (i + 0.5) / n_opt) for i in range(n_opt)])
    avg = res.abs().mean(0)
    return cut, wts, avg


def compress_documents(docs: list[torch.Tensor], centroids: torch.Tensor, cutoffs: torch.Tensor, nbits: int, cast_cutoffs: bool = True):
    """create.rs:404-428 process_batch: codes + packed residual bytes.  cast_cutoffs=True buckets against the cutoffs cast to Half
    (what update.rs does with the loaded codec, and what the committed search fixtures were generated with); False hands
    torch.bucketize the fp32 cutoffs as create.rs:413 does (ATen promotes the Half residuals: an fp32 comparison)."""
    dim = centroids.shape[1]
    emb = torch.cat([d.to(torch.float16) for d in docs])
    codes = compress_into_codes(emb, centroids)
    res = emb - centroids.index_select(0, codes)
    b = torch.bucketize(res, cutoffs.to(res.dtype) if cast_cutoffs else cutoffs, out_int32=True, right=False)  # :413
    b = b.unsqueeze(-1).expand(*b.shape, nbits)
    b = b.bitwise_right_shift(torch.arange(nbits, dtype=torch.int8))  # :416-418 bit_helper
    b = b.bitwise_and(torch.ones_like(b))
    packed = packbits(b.flatten()).reshape(emb.shape[0], dim // 8 * nbits)
    return codes, packed


def build_ivf(codes: torch.Tensor, doclens: torch.Tensor, num_partitions: int):
    """create.rs:528-559 + optimize_ivf :55-132 -> (ivf pids int64, ivf_lengths int32)."""
    srt_codes, srt_idx = codes.sort()
    counts = torch.bincount(srt_codes, minlength=num_partitions)
    emb2pid = torch.repeat_interleave(torch.arange(doclens.shape[0]), doclens)
    pids = emb2pid.index_select(0, srt_idx)
    out, lens, off = [], [], 0
    for ln in counts.tolist():
        u = torch.unique(pids.narrow(0, off, ln), sorted=True)
        out.append(u)
        lens.append(u.shape[0])
        off += ln
    ivf = torch.cat(out) if out else torch.empty((0,), dtype=torch.int64)
    return ivf.to(torch.int64), torch.tensor(lens, dtype=torch.int32)


def num_partitions_for(n_tokens_est: float) -> int:
    """fast_plaid.py:150-154 / create.rs:292-294."""
    return int(2 ** math.floor(math.log2(16 * math.sqrt(n_tokens_est))))


def build_index_arrays(docs: list[torch.Tensor], centroids: torch.Tensor, nbits: int, num_partitions: int | None = None,
                       cast_cutoffs: bool = True):
    """End-to-end array set that ``construct_index`` consumes (load.py:220-322 layout:
    codes/residuals carry ``max_len - last_len`` trailing padding rows)."""
    centroids = centroids.to(torch.float16)
    doclens = torch.tensor([d.shape[0] for d in docs], dtype=torch.int64)
    allemb = torch.cat([d.to(torch.float16) for d in docs])
    cut, wts, avg = train_codec(allemb, centroids, nbits)
    codes, packed = compress_documents(docs, centroids, cut, nbits, cast_cutoffs)
    if num_partitions is None:
        num_partitions = max(centroids.shape[0], 1)
    ivf, ivf_lengths = build_ivf(codes, doclens, num_partitions)
    pad = max(0, int(doclens.max()) - int(doclens[-1])) if len(docs) else 0
    codes_p = torch.cat([codes, torch.zeros(pad, dtype=torch.int64)])
    res_p = torch.cat([packed, torch.zeros((pad, packed.shape[1]), dtype=torch.uint8)])
    return dict(
        nbits=nbits,
        centroids=centroids,
        avg_residual=avg.to(torch.float16),
        bucket_cutoffs=cut.to(torch.float16),
        bucket_weights=wts.to(torch.float16),
        ivf=ivf,
        ivf_lengths=ivf_lengths,
        doc_codes=codes_p,
        doc_residuals=res_p,
        doc_lengths=doclens,
    )
