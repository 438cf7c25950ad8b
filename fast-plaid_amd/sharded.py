"""Document-sharded PLAID search: one process per GPU, collectives over torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU protocol tests).

New relative to the reference, whose multi-GPU mode is full replicas + query split
(python/fast_plaid/search/fast_plaid.py:893-928).  Every rank holds the full centroid table
and a contiguous, token-balanced range of documents.  Per batch of queries:

  stage1  (local)   S1..S4 on the shard, local top-R candidates by approximate score
  all-gather        [B,R] records {i64 id, f32 approx} (16 B)                                  (R = max(n_full/4, 1))
  stage2  (local)   global top-R cut == search.rs:605-619 applied to the union, exact
                    MaxSim of the survivors that live here
  all-gather        [B,R] records {i64 id, f32 score, f32 budget_down, f32 budget} (24 B)
  stage3  (local)   union in id order, near-tie marking as in the unsharded search (identical on every rank); exact-order
                    repair of the marked documents that live here
  all-gather        [B,R] f32: the repaired scores, by union position
  stage4  (local)   marked documents take their repaired score, global (score desc, id asc) sort, top_k

The messages are fixed-size buffers (B*R*16, B*R*24 and B*R*4 bytes per rank; 11 MB at B=256, R=1024): latency-bound on
xGMI, so each is ONE all_gather_into_tensor call on a device buffer -- no host staging, no variable-length exchange, no
repacking (the library writes and reads the record layout directly).  The result is identical to the unsharded search on
the concatenated corpus, bit for bit (tests/test_hip_parity.py::test_sharded_equals_unsharded).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from . import fast_plaid_rust as native


def plan_shards(doc_lengths, world_size: int) -> list[tuple[int, int]]:
    """Contiguous doc ranges with (nearly) equal token mass. `doc_lengths` may be an int
    (uniform lengths: n_docs) or an array."""
    if isinstance(doc_lengths, (int, np.integer)):
        n = int(doc_lengths)
        cuts = [round(i * n / world_size) for i in range(world_size + 1)]
    else:
        lens = np.asarray(doc_lengths, dtype=np.int64)
        n = int(lens.shape[0])
        cum = np.concatenate([[0], np.cumsum(lens)])
        total = int(cum[-1])
        cuts = [0]
        for i in range(1, world_size):
            cuts.append(int(np.searchsorted(cum, i * total / world_size, side="left")))
        cuts.append(n)
        for i in range(1, len(cuts)):
            cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[i], cuts[i + 1]) for i in range(world_size)]


def shard_arrays(arrays: dict, begin: int, end: int) -> dict:
    """Slice a construct_index argument set to documents [begin, end): codes/residuals rows
    of those docs, IVF restricted (and re-based) to them."""
    lens = np.asarray(arrays["doc_lengths"], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)])
    t0, t1 = int(offs[begin]), int(offs[end])
    out = dict(arrays)
    out["doc_lengths"] = lens[begin:end].copy()
    out["doc_codes"] = np.asarray(arrays["doc_codes"])[t0:t1].copy()
    out["doc_residuals"] = np.asarray(arrays["doc_residuals"])[t0:t1].copy()
    if arrays.get("ivf") is not None:
        ivf = np.asarray(arrays["ivf"], dtype=np.int64)
        il = np.asarray(arrays["ivf_lengths"], dtype=np.int64)
        cell = np.repeat(np.arange(il.shape[0]), il)
        keep = (ivf >= begin) & (ivf < end)
        out["ivf"] = (ivf[keep] - begin).astype(np.int64)
        out["ivf_lengths"] = np.bincount(cell[keep], minlength=il.shape[0]).astype(np.int32)
    return out


REC1_BYTES = 16   # fp_shard_rec1 {i64 pid; f32 approx; i32 pad}
REC2_BYTES = 24   # fp_shard_rec2 {i64 pid; f32 score; f32 unc_down; f32 unc; i32 pad}
REC1_DTYPE = np.dtype([("pid", "<i8"), ("approx", "<f4"), ("pad", "<i4")])
REC2_DTYPE = np.dtype([("pid", "<i8"), ("score", "<f4"), ("unc_down", "<f4"), ("unc", "<f4"), ("pad", "<i4")])


class HipShardEngine:
    """Per-rank engine over the C ABI's fp_shard_* entry points.  Device buffers are torch
    tensors so that torch.distributed (RCCL) can move them without host staging."""

    def __init__(self, index: native.PyLoadedIndex, device):
        import torch
        if not N.TORCH_LOADED_FIRST:
            raise RuntimeError(
                "import torch before the first fast_plaid_amd call in a process that uses the sharded path: torch wheels "
                "bundle their own libamdhip64.so.7 and cannot see the GPU once another HIP runtime has been loaded")
        self.torch = torch
        self.index = index
        self.device = torch.device(device)

    def begin(self, queries_f16: np.ndarray, params: native.SearchParameters):
        q = np.ascontiguousarray(queries_f16, dtype=np.float16)
        B, Q, D = q.shape
        ctx = C.c_void_p()
        p = params._c()
        N.check(N.lib().fp_shard_begin(self.index._h, q.ctypes.data_as(C.c_void_p), B, Q, D, C.byref(p), C.byref(ctx)))
        R = int(N.lib().fp_shard_R(ctx))
        return dict(ctx=ctx, B=B, R=R, top_k=params.top_k)

    def stage1(self, st):
        t = self.torch
        rec = t.empty((st["B"], st["R"] * REC1_BYTES), dtype=t.uint8, device=self.device)
        N.check(N.lib().fp_shard_stage1(st["ctx"], rec.data_ptr()))
        return rec

    def stage2(self, st, all_rec1, world):
        t = self.torch
        rec = t.empty((st["B"], st["R"] * REC2_BYTES), dtype=t.uint8, device=self.device)
        N.check(N.lib().fp_shard_stage2(st["ctx"], all_rec1.data_ptr(), world, rec.data_ptr()))
        return rec

    def stage3(self, st, all_rec2, world, rank):
        t = self.torch
        x = t.zeros((st["B"], st["R"]), dtype=t.float32, device=self.device)
        N.check(N.lib().fp_shard_stage3(st["ctx"], all_rec2.data_ptr(), world, rank, x.data_ptr()))
        return x

    def stage4(self, st, all_x, world):
        B, k = st["B"], max(st["top_k"], 1)
        pids = np.full((B, k), -1, np.int64)
        scores = np.zeros((B, k), np.float32)
        counts = np.zeros(B, np.int32)
        N.check(N.lib().fp_shard_stage4(st["ctx"], all_x.data_ptr(), world, pids.ctypes.data_as(C.c_void_p),
                                        scores.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p)))
        return pids[:, : st["top_k"]], scores[:, : st["top_k"]], counts

    def end(self, st):
        N.lib().fp_shard_end(st["ctx"])


def group_broadcast_bytes(dist, raw, group=None, nbytes=128):
    """`raw` (bytes on the group's first member, None elsewhere) -> the same bytes on every member of `group`.
    `src` of dist.broadcast is a GLOBAL rank: a sub-group that does not contain global rank 0 (every query group but the first
    of the 2-D grid) has to name its own first member."""
    import torch
    t = torch.zeros(nbytes, dtype=torch.uint8) if raw is None else torch.frombuffer(bytearray(raw), dtype=torch.uint8).clone()
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast(t, src=src, group=group)
    return bytes(t.cpu().numpy().tobytes())


class NativeComm:
    """RCCL communicator owned by the library (include/fastplaid.h fp_comm_*): rank 0 draws the 128-byte unique id, the caller's
    `broadcast(bytes_or_None) -> bytes` ships it (torch.distributed / MPI / a file -- any out-of-band channel), every rank joins."""

    def __init__(self, device_id: int, world: int, rank: int, broadcast):
        uid = (C.c_ubyte * 128)()
        if rank == 0:
            N.check(N.lib().fp_comm_unique_id(C.cast(uid, C.c_void_p)))
        raw = broadcast(bytes(uid) if rank == 0 else None)
        buf = (C.c_ubyte * 128).from_buffer_copy(raw)
        h = C.c_void_p()
        N.check(N.lib().fp_comm_create(int(device_id), int(world), int(rank), C.cast(buf, C.c_void_p), C.byref(h)))
        self._h, self.world, self.rank = h, world, rank

    @classmethod
    def from_torch_dist(cls, device_id: int, dist, group=None):
        """bootstrap through an initialised torch.distributed process group (the id travels as a CPU byte tensor)."""
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        return cls(device_id, world, rank, lambda raw: group_broadcast_bytes(dist, raw, group))

    def close(self):
        if getattr(self, "_h", None):
            N.lib().fp_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def native_sharded_search(index: native.PyLoadedIndex, comm: NativeComm, queries_f16, params):
    """fp_shard_search: the three stages and the two RCCL all-gathers inside ONE library call (collective over `comm`)."""
    q = np.ascontiguousarray(queries_f16, dtype=np.float16)
    B, Q, D = q.shape
    k = max(params.top_k, 1)
    pids = np.full((B, k), -1, np.int64)
    scores = np.zeros((B, k), np.float32)
    counts = np.zeros(B, np.int32)
    p = params._c()
    N.check(N.lib().fp_shard_search(index._h, comm._h, q.ctypes.data_as(C.c_void_p), B, Q, D, C.byref(p), pids.ctypes.data_as(C.c_void_p),
                                    scores.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p)))
    return pids[:, : params.top_k], scores[:, : params.top_k], counts


def _all_gather(torch, dist, x, world, group, force=False):
    """[B,R] per rank -> [G,B,R] on every rank."""
    if dist is None or (world == 1 and not force):
        return x.unsqueeze(0).contiguous()
    src = x.contiguous()
    if src.is_cuda and dist.get_backend(group) == "gloo":   # test mode (several ranks on one GPU): stage through the host
        src = src.cpu()
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src, group=group)  # concatenation along dim 0 == [G][B][R]
    return out.view((world,) + tuple(x.shape)).to(x.device)


def sharded_search(engine, queries_f16, params, dist=None, group=None, force_collectives=False):
    """Runs one batch through the four stages with the three all-gathers in between.
    `engine` implements begin/stage1/stage2/stage3/stage4/end (HipShardEngine in production).
    Returns (pids [B,top_k], scores, counts) -- identical on every rank."""
    import torch
    world = dist.get_world_size(group) if dist is not None and dist.is_initialized() else 1
    st = engine.begin(queries_f16, params)
    try:
        rec1 = engine.stage1(st)                                     # [B, R*16] bytes (the library's stream is idle on return)
        all1 = _all_gather(torch, dist, rec1, world, group, force_collectives)
        if all1.is_cuda:
            torch.cuda.current_stream().synchronize()  # library kernels run on their own stream
        rec2 = engine.stage2(st, all1, world)                        # [B, R*24] bytes
        all2 = _all_gather(torch, dist, rec2, world, group, force_collectives)
        if all2.is_cuda:
            torch.cuda.current_stream().synchronize()
        rank = dist.get_rank(group) if dist is not None and dist.is_initialized() else 0
        x = engine.stage3(st, all2, world, rank)                     # [B, R] f32: repaired scores of this rank's near-tied documents
        allx = _all_gather(torch, dist, x, world, group, force_collectives)
        if allx.is_cuda:
            torch.cuda.current_stream().synchronize()
        return engine.stage4(st, allx, world)
    finally:
        engine.end(st)


# ---- replicated index, batch split across ranks -------------------------------------------------
# The reference's own multi-GPU mode (python/fast_plaid/search/fast_plaid.py:893-928: one full
# index per device, the query list cut into one contiguous slice per device).  Every stage of the
# path is per query, so this mode replicates NO work (the document-sharded mode above replicates
# S1/S2 and the per-launch latencies on every rank); it is the throughput mode whenever the index
# fits one GPU (a 288 GB MI355X holds ~30 M documents x 128 tokens at nbits=4).  The only exchange
# is the result gather: one fixed-size all_gather_into_tensor call.
def split_batch(n_queries: int, world_size: int) -> list[tuple[int, int]]:
    """contiguous, near-equal query ranges, one per rank (the first ranks take the remainder)."""
    base, rem = divmod(int(n_queries), int(world_size))
    out, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def plan_grid(world_size: int, doc_shards: int) -> list[tuple[int, int]]:
    """2-D layout of `world_size` ranks: (document shard d, query group g) of every rank, rank = g * doc_shards + d.  The
    ranks of one query group are consecutive (on one node: neighbouring GPUs, whose xGMI links carry the group's exchanges)."""
    if doc_shards < 1 or world_size % doc_shards:
        raise ValueError(f"{world_size} ranks do not divide into groups of {doc_shards} document shards")
    return [(r % doc_shards, r // doc_shards) for r in range(world_size)]


def replicated_search(search_local, queries_f16, top_k: int, dist=None, group=None, device="cpu", force_collectives=False,
                      group_size: int = 1):
    """`search_local(q [b,Q,D]) -> (pids [b,top_k] i64, scores [b,top_k] f32, counts [b] i32)` on this
    rank's full replica.  Returns the whole batch's (pids, scores, counts), identical on every rank.

    group_size > 1 = the 2-D layout (plan_grid): `group_size` consecutive ranks form a query group that holds the corpus
    split into `group_size` document shards; the batch is cut into one slice per GROUP and `search_local` is the group's
    document-sharded search of its slice (a collective over the group: fp_shard_search on the group's communicator), whose
    result is the same on the group's ranks.  S1 -- the stage a document-sharded search replicates on every rank -- then
    runs on 1/n_groups of the batch per rank."""
    import torch
    q = np.ascontiguousarray(queries_f16)
    B = q.shape[0]
    live = dist is not None and dist.is_initialized()
    world = dist.get_world_size(group) if live else 1
    rank = dist.get_rank(group) if live else 0
    if group_size < 1 or world % group_size:
        raise ValueError(f"{world} ranks do not divide into groups of {group_size}")
    gsz = group_size
    ranges_g = split_batch(B, world // gsz)
    ranges = [ranges_g[r // gsz] for r in range(world)]   # (per rank; the ranks of a group share a slice)
    lo, hi = ranges[rank]
    mb = max(h - l for l, h in ranges)
    pids = np.full((mb, top_k), -1, np.int64)
    scores = np.zeros((mb, top_k), np.float32)
    counts = np.zeros((mb,), np.int32)
    if hi > lo:
        p, s, c = search_local(q[lo:hi])
        pids[: hi - lo], scores[: hi - lo], counts[: hi - lo] = p, s, c
    if not live or (world == gsz and not force_collectives):   # one group: its result is the batch's
        return pids[:B], scores[:B], counts[:B]
    # ONE exchange per batch: ids | scores | counts packed row-wise into a byte tensor (three separate small all-gathers cost
    # three collective latencies, which is what a rank's 8-query slice of a 64-query batch is measured in)
    row = top_k * 12 + 4
    packed = np.empty((mb, row), np.uint8)
    packed[:, : top_k * 8] = pids.view(np.uint8).reshape(mb, top_k * 8)
    packed[:, top_k * 8: top_k * 12] = scores.view(np.uint8).reshape(mb, top_k * 4)
    packed[:, top_k * 12:] = counts.view(np.uint8).reshape(mb, 4)
    t = torch.from_numpy(packed).to(device)
    g = torch.empty((world * mb, row), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(g, t, group=group)
    allp = g.cpu().numpy().reshape(world, mb, row)
    lead = [(r, lh) for r, lh in enumerate(ranges) if r % gsz == 0]   # one contribution per group: its first rank's
    P = np.concatenate([np.ascontiguousarray(allp[r, : h - l, : top_k * 8]).view(np.int64).reshape(h - l, top_k) for r, (l, h) in lead], axis=0)
    S = np.concatenate([np.ascontiguousarray(allp[r, : h - l, top_k * 8: top_k * 12]).view(np.float32).reshape(h - l, top_k)
                        for r, (l, h) in lead], axis=0)
    Cn = np.concatenate([np.ascontiguousarray(allp[r, : h - l, top_k * 12:]).view(np.int32).reshape(h - l) for r, (l, h) in lead], axis=0)
    return P, S, Cn
