#!/usr/bin/env python
"""Benchmark of the MI355X PLAID search hot path (contract: see the task's bench.py section).

A "step" = one fp_search call over one batch of synthetic queries against a synthetic,
HBM-resident compressed corpus.  Default workload = BASELINE.json configs[1]:
1M docs x 128 tok x dim128 (nbits 4, 2^17 centroids), batch 64 queries x 32 tok, top_k=1000,
n_full_scores=4096 (R=1024 exact-scored docs / query), n_ivf_probe=8.

    python bench.py                               # N=1, finishes in a few minutes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N         # N GPUs, one process each (RCCL)

N>1 defaults to BASELINE.json configs[2] (`--config cfg3`): 10M docs x 128 tok, 2^19 centroids, batch 256, top_k 1000, the corpus
split into N contiguous token-balanced DOCUMENT SHARDS, the SAME batch on every rank, three fixed-size RCCL all-gathers per batch
issued by the library itself on its search stream (fp_shard_search) -- "scaling": "strong"; `python bench.py --gpus 1 --config cfg3`
is the one-GPU anchor of that series (the newest committed measurement of it is echoed as `n1_anchor`).
Other `--dist-mode`s (fast-plaid_amd/sharded.py), one flag away: "replica" = the reference's multi-GPU layout -- a full index per
GPU, one process per GPU -- with every rank searching its OWN batch per step through fp_search and no data-path collective
("weak"); "split" = the same replicas with ONE batch split across the ranks and a result all-gather ("strong").
With --alt-mode a second mode is timed too and reported as `alt_mode`.

One JSON line on rank 0.  `value` is timed on the boundary call fp_search (host query buffer in, host
results out -- what the reference's pysearch hands over, rust/lib.rs:195-223); the same steps with queries and
results resident in HBM (fp_search_device) are reported as `value_device_io`.  Every step uses its own query
batch.  `roofline` describes the DOMINANT kernel of the step: whichever of the three single-kernel stages -- S1's centroid
GEMM (MFMA bound), S4's per-candidate scan (HBM), the fused decompress + MaxSim kernel (HBM; the kernel BASELINE.json's north star
puts the roofline target on) -- has the longest measured launch; all three are always under `roofline_by_kernel` (the MaxSim entry
also with its exact-order repair included: `frac_with_repair`); `stages_ms` gives every stage.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_F16_PEAK_TFLOPS = 2500.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=["auto", "cfg2", "cfg3"], default="auto",
                    help="BASELINE.json configs[1] (cfg2: 1M docs, batch 64; the single-GPU metric) or configs[2] (cfg3: 10M docs, batch 256, "
                         "document-sharded over the ranks: the multi-GPU configuration); auto = cfg2 on one GPU, cfg3 on several.  "
                         "`--gpus 1 --config cfg3` is the one-GPU anchor of the strong-scaling series")
    ap.add_argument("--docs", type=int, default=0, help="0 = the config's (1M / 10M)")
    ap.add_argument("--doc-len", type=int, default=128)
    ap.add_argument("--centroids", type=int, default=0, help="0 = 2^floor(log2(16*sqrt(tokens)))")
    ap.add_argument("--batch", type=int, default=0, help="0 = the config's (64 / 256)")
    ap.add_argument("--qlen", type=int, default=32)
    ap.add_argument("--topk", type=int, default=1000)
    ap.add_argument("--nfull", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=8)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--nbits", type=int, default=4)
    ap.add_argument("--cpu-queries", type=int, default=-1, help="queries for the CPU baseline leg (0 = skip, -1 = auto)")
    ap.add_argument("--workload", type=str, default="", help="label override")
    ap.add_argument("--dup-centroids", type=int, default=0, help="make centroid rows 1000 .. 1000 + N identical (exact ties of the centroid scores: a query "
                    "token near them overflows the threshold probe's tie room and takes its fallback); 0 = the config's table")
    ap.add_argument("--zero-rows", type=int, default=0, help="zero the last N token rows of every query (what list inputs of unequal lengths "
                    "are padded with, fast_plaid.py:772-780); 0 = the config's full-length queries")
    ap.add_argument("--force-dist", action="store_true", help="run the sharded/RCCL code path even with one rank (testing)")
    ap.add_argument("--doc-shards", type=int, default=0, help="grid mode: document shards per query group (0 = 2; the other factor of "
                    "--gpus is the number of query groups)")
    ap.add_argument("--dist-mode", choices=["auto", "replica", "split", "shard", "grid"], default="auto",
                    help="N>1: replica = full index per GPU, every rank searches its OWN batch of --batch queries per step, no "
                         "data-path collective (weak scaling: the reference's replicas, one client stream per GPU); split = "
                         "full index per GPU, ONE batch split across the ranks + a result all-gather (strong); shard = document "
                         "shards + 3 RCCL all-gathers per batch (strong); auto = replica when the index fits one GPU, else shard")
    ap.add_argument("--alt-mode", action="store_true", help="N>1: also time the other distribution mode and report it as alt_mode")
    ap.add_argument("--no-alt-mode", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--dist-impl", choices=["native", "torch"], default="native",
                    help="shard mode: native = fp_shard_search (the library issues the RCCL all-gathers itself); torch = the three stage calls "
                         "with torch.distributed all-gathers in between (forced with --dist-backend gloo)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo = result gather through CPU tensors, ranks wrapped onto the visible GPUs (testing the multi-process "
                         "path on a box with fewer GPUs than ranks)")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    a.cfg = a.config if a.config != "auto" else ("cfg3" if max(world, a.gpus) > 1 else "cfg2")
    if a.docs == 0:
        a.docs = 10_000_000 if a.cfg == "cfg3" else 1_000_000
    if a.batch == 0:
        a.batch = 256 if a.cfg == "cfg3" else 64
    return a


def _pmc_traffic(kernel_prefix, applicable):
    """HBM-side read bytes per launch of one kernel, from the newest committed PMC summary
    (profiles/*_pmc_traffic.json, written by tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE in its own
    pass, doubled per MI355X_MICROARCH.md's gfx950 correction).  PMC counters cannot be collected from
    inside this process, so the number is the one measured on the default workload; any other workload
    reports null -- and so does a summary that was measured on a DIFFERENT build of the library than the one
    loaded now (the summary carries the sha256 of the .so it profiled)."""
    import glob
    import hashlib
    if not applicable:
        return None, None
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "profiles", "*_pmc_traffic.json")))   # (tags sort by round: r02_f < r03_a)
    if not files:
        return None, None
    try:
        doc = json.load(open(files[-1]))
        ks = doc["kernels"]
        lib = os.path.join(here, "fast-plaid_amd", "libfastplaid_hip.so")
        sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
    except Exception:
        return None, None
    if doc.get("library_sha16") != sha:
        return None, "profiles/" + os.path.basename(files[-1]) + " is from another build of the library (ignored)"
    for name, e in ks.items():
        if name.startswith(kernel_prefix) and "hbm_read_bytes_corrected" in e:
            return int(e["hbm_read_bytes_corrected"]), "profiles/" + os.path.basename(files[-1])
    return None, None


def main():
    a = parse()
    # stdout carries exactly ONE line (rank 0's JSON): RCCL prints a version banner to stdout when a communicator is created, and
    # any library may do likewise, so fd 1 points at stderr until the result is printed
    sys.stdout.flush()
    _real_stdout = os.dup(1)
    os.dup2(2, 1)
    default_cfg = (a.cfg == "cfg2" and a.docs == 1_000_000 and a.doc_len == 128 and a.centroids == 0 and a.batch == 64 and a.qlen == 32 and
                   a.topk == 1000 and a.nfull == 4096 and a.nprobe == 8 and a.dim == 128 and a.nbits == 4)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and world > 1:
        a.gpus = world
    dist = None
    torch = None
    use_dist = world > 1 or a.force_dist
    if use_dist:
        import torch  # BEFORE the HIP library: see fast-plaid_amd/sharded.py
        import torch.distributed as dist
        if a.dist_backend == "gloo":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(a.dist_backend, rank=rank, world_size=world)

    import fast_plaid_amd as fp
    from fast_plaid_amd import sharded
    R = fp.fast_plaid_rust

    n_tokens = a.docs * a.doc_len
    C = a.centroids or fp.synth.default_num_centroids(n_tokens)
    spec = fp.synth.SynthSpec(n_docs=a.docs, doc_len=a.doc_len, n_centroids=C, dim=a.dim, nbits=a.nbits, seed=42)
    cent = fp.synth.centroids(spec)
    if a.dup_centroids > 0:
        cent = cent.copy()
        cent[1000:1000 + a.dup_centroids] = cent[1000]
    bw = fp.synth.bucket_weights(spec)
    dev = f"cuda:{local_rank}"
    params = R.SearchParameters(2000, a.nfull, a.topk, a.nprobe)
    n_batches = a.steps + a.warmup   # a distinct query batch for every step (no cache-warm repeats)
    batches = [fp.synth.make_queries(spec, cent, a.batch, a.qlen, seed=1000 + i) for i in range(n_batches)]
    if a.zero_rows > 0:
        for q in batches:
            q[:, max(a.qlen - a.zero_rows, 0):, :] = 0
    # "replica" mode: every rank has its own stream of batches (rank 0's is the single-GPU stream)
    own_batches = batches if rank == 0 else [fp.synth.make_queries(spec, cent, a.batch, a.qlen, seed=1000 + i + 7919 * rank)
                                             for i in range(n_batches)]

    def sync():
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def run_mode(mode):
        """builds the index for `mode`, W warm-up steps, then EXACTLY K timed steps bracketed by
        barrier + synchronize; returns the max-over-ranks elapsed time and per-step records."""
        shard_group, gsz = None, 1
        if mode == "shard":
            lo, hi = sharded.plan_shards(a.docs, world)[rank]
        elif mode == "grid":
            # 2-D layout: `gsz` consecutive ranks hold the corpus as gsz document shards and search one slice of the batch
            gsz = a.doc_shards if a.doc_shards > 0 else 2
            if world % gsz:
                gsz = 1
            d_idx, g_idx = sharded.plan_grid(world, gsz)[rank]
            lo, hi = sharded.plan_shards(a.docs, gsz)[d_idx]
            groups = [dist.new_group([g * gsz + d for d in range(gsz)]) for g in range(world // gsz)]   # (every rank creates every group)
            shard_group = groups[g_idx]
        else:
            lo, hi = 0, a.docs
        t0 = time.time()
        index = R.construct_synthetic_index(spec, dev, doc_begin=lo, doc_end=hi, centroids=cent, bucket_weights=bw)
        t_build = time.time() - t0
        native_comm = None
        engine = None
        if mode in ("shard", "grid"):
            if a.dist_impl == "native" and a.dist_backend == "nccl":
                native_comm = sharded.NativeComm.from_torch_dist(index.device_id, dist, group=shard_group)
            else:
                engine = sharded.HipShardEngine(index, dev)

        # single-GPU mode: the timed path is the boundary call fp_search -- host query buffer in, host results out (the index is
        # resident in HBM; + 512 KB up, 768 KB down per batch over PCIe); fp_search_device (queries / results resident in HBM) is
        # timed separately below and reported beside it.

        def step(i):
            q = batches[i % n_batches]
            if mode == "shard":
                if native_comm is not None:
                    return sharded.native_sharded_search(index, native_comm, q, params)
                return sharded.sharded_search(engine, q, params, dist=dist, force_collectives=True)
            if mode == "grid":
                if native_comm is not None:
                    in_group = lambda qs: sharded.native_sharded_search(index, native_comm, qs, params)   # noqa: E731
                else:
                    in_group = lambda qs: sharded.sharded_search(engine, qs, params, dist=dist, group=shard_group, force_collectives=True)   # noqa: E731
                return sharded.replicated_search(in_group, q, a.topk, dist=dist, device=("cpu" if a.dist_backend == "gloo" else dev),
                                                 force_collectives=True, group_size=gsz)
            if mode == "split":
                return sharded.replicated_search(lambda qs: R.search_arrays(index, qs, params), q, a.topk, dist=dist,
                                                 device=("cpu" if a.dist_backend == "gloo" else dev), force_collectives=True)
            if mode == "replica":   # this rank's own batch on its own replica: the fp_search boundary, nothing else
                return R.search_arrays(index, own_batches[i % n_batches], params)
            return R.search_arrays(index, q, params)

        for i in range(a.warmup):
            step(i)
        stage_acc: dict[str, float] = {}
        s4_form = None
        s1_form = None
        lat = []
        cand_total = 0
        exact_total = 0
        repaired_total = 0
        sync()
        t_start = time.perf_counter()
        for i in range(a.steps):
            ts = time.perf_counter()
            step(a.warmup + i)
            lat.append(time.perf_counter() - ts)
        sync()
        elapsed = time.perf_counter() - t_start
        # `value` is the K steps above and nothing else.  K = 20 steps are ~37 ms: for the reader, the same K steps five more times
        # (reported as `repeat_ms_per_step`: how far one timed region of this length scatters on this box)
        repeats = []
        if mode == "single":
            for _ in range(5):
                sync()
                t_r = time.perf_counter()
                for i in range(a.steps):
                    step(a.warmup + i)
                sync()
                repeats.append((time.perf_counter() - t_r) / a.steps * 1e3)
        if mode in ("single", "replica"):
            # per-stage HIP-event times and work counters: the SAME K steps once more with graph replay off -- in the timed loop
            # above every step is one hipGraphLaunch, which records no per-stage events (same kernels, same launch parameters)
            was = R.set_graph_replay(False)
            step(a.warmup)   # (the first call after the switch re-learns nothing, but keep it out of the averages)
            for i in range(a.steps):
                step(a.warmup + i)
                for k, v in R.last_search_timings().items():
                    stage_acc[k] = stage_acc.get(k, 0.0) + v
                cnts = R.last_search_counts()
                s4_form = cnts.get("s4_form")
                s1_form = {1: "lazy (upper candidates; the probe and the selection settle what they use)", 0: "eager (certified + repaired in S1)"}.get(
                    R.last_s1_counts().get("lazy"), "?")
                cand_total += cnts["candidates"]
                exact_total += cnts["approx_exact"]
                repaired_total += cnts.get("repaired", 0)
            R.set_graph_replay(was)
            sync()
        if use_dist:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=("cpu" if a.dist_backend == "gloo" else dev))
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        dev_elapsed = None
        if mode == "single":   # the same K steps with queries uploaded beforehand and results left in HBM
            devid = index.device_id
            dq = [R.DeviceBuffer(devid, q.nbytes).upload(q) for q in batches]
            dp = R.DeviceBuffer(devid, a.batch * max(a.topk, 1) * 8)
            dsc = R.DeviceBuffer(devid, a.batch * max(a.topk, 1) * 4)
            dc = R.DeviceBuffer(devid, a.batch * 4)
            for i in range(min(a.warmup, 2)):
                R.search_device(index, dq[i], a.batch, a.qlen, params, dp, dsc, dc)
            t1 = time.perf_counter()
            for i in range(a.steps):
                R.search_device(index, dq[a.warmup + i], a.batch, a.qlen, params, dp, dsc, dc)
            dev_elapsed = time.perf_counter() - t1
            del dq, dp, dsc, dc
        return dict(index=index, elapsed=elapsed, lat=lat, repeats=repeats, stage_acc=stage_acc, cand_total=cand_total, exact_total=exact_total,
                    repaired_total=repaired_total, s4_form=s4_form, s1_form=s1_form,
                    t_build=t_build, dev_elapsed=dev_elapsed)

    if use_dist:
        est_bytes = a.docs * a.doc_len * (a.dim * a.nbits // 8 + 4 + 2 + 4) + C * a.dim * 2
        hbm = torch.cuda.get_device_properties(local_rank).total_memory
        fits = est_bytes * 2.2 + a.batch * C * 64 < 0.8 * hbm   # index + build scratch + S
        # auto: BASELINE configs[2] is the SHARDED index (documents split over the ranks, RCCL all-gathers in the data path, strong
        # scaling against `--gpus 1 --config cfg3`); replicas only for an explicitly requested single-GPU-sized workload
        # (round 4: the sharded default is the 2-D layout -- 2 document shards x N/2 query groups -- because a pure document split
        # repeats S1, the centroid GEMM of the WHOLE batch, on every rank: 6 of 68 ms at cfg3, the term that capped 8 ranks at
        # 5.6 x.  With two ranks it is the pure document split.)
        primary = a.dist_mode if a.dist_mode != "auto" else (("grid" if world % 2 == 0 else "shard") if (a.cfg == "cfg3" or not fits) else "replica")
        other = "shard" if primary in ("replica", "split") else "split"
        run_other = a.alt_mode and (not a.no_alt_mode) and (other == "shard" or fits)
    else:
        primary, run_other = "single", False
    res = run_mode(primary)
    index, elapsed, lat, stage_acc, cand_total, t_build = (res[k] for k in ("index", "elapsed", "lat", "stage_acc", "cand_total", "t_build"))
    exact_total = res["exact_total"]
    dev_elapsed = res["dev_elapsed"]
    ms_per_step = elapsed / a.steps * 1e3
    weak = primary == "replica"
    global_batch = a.batch * (world if weak else 1)   # replica: every rank searched its own batch in each step
    qps = global_batch * a.steps / elapsed
    index_bytes = index.device_bytes
    alt = None
    if run_other:
        del res
        index = None   # free the primary mode's index before building the other one
        r2 = run_mode(other)
        alt = {"mode": other, "value": a.batch * a.steps / r2["elapsed"], "ms_per_step": r2["elapsed"] / a.steps * 1e3}
        del r2
    par = {"single": "1 GPU",
           "replica": "%d full-index replicas, one process per GPU, each searching its own batch of %d queries per step through "
                      "fp_search; no data-path collective (barrier + max-over-ranks timing only)" % (world, a.batch),
           "split": "full index replica per GPU, batch split %d-way, result all-gather over %s" % (
               world, "RCCL" if a.dist_backend == "nccl" else "gloo (test mode: ranks share the visible GPUs)"),
           "grid": "%d document shards x %d query groups: fp_shard_search (3 RCCL all-gathers/batch) inside a group of %d neighbouring "
                   "ranks on its slice of the batch, one result all-gather over all ranks" % (
                       (a.doc_shards or 2) if world % (a.doc_shards or 2) == 0 else 1, world // ((a.doc_shards or 2) if world % (a.doc_shards or 2) == 0 else 1),
                       (a.doc_shards or 2) if world % (a.doc_shards or 2) == 0 else 1),
           "shard": "document-sharded x%d, 3 RCCL all-gathers/batch (%s)" % (
               world, "issued by the library on the search stream" if (a.dist_impl == "native" and a.dist_backend == "nccl") else "torch.distributed between four stage calls")}[primary]

    out = {
        "metric": "queries/sec @ top_k=%d (batch=%d, dim=%d); p50 search latency" % (a.topk, a.batch, a.dim),
        "value": qps, "unit": "queries/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": ("strong" if primary in ("shard", "split", "grid") else "weak"),
        "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {
            "workload": a.workload or ("BASELINE %s: %d docs x %d tok x dim%d, nbits=%d, %d centroids, batch=%d x %d tok, "
                                       "top_k=%d, n_full_scores=%d, n_ivf_probe=%d" % (
                                           a.cfg, a.docs, a.doc_len, a.dim, a.nbits, C, a.batch, a.qlen, a.topk, a.nfull, a.nprobe)),
            "parallelism": par, "global_batch": global_batch,
            "index_bytes_per_gpu": index_bytes, "index_build_s": round(t_build, 2),
        },
        "p50_ms": float(np.percentile(np.array(lat) * 1e3, 50)), "p90_ms": float(np.percentile(np.array(lat) * 1e3, 90)),
        "repeat_ms_per_step": [round(x, 4) for x in res.get("repeats", [])],
    }
    if primary in ("shard", "split", "grid") or (a.cfg == "cfg3" and world == 1):
        # the one-GPU anchor of the strong-scaling series: `python bench.py --gpus 1 --config cfg3` (the newest committed measurement of it)
        import glob
        anchors = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_cfg3_1gpu.json")))
        if anchors and world > 1:
            try:
                an = json.loads(open(anchors[-1]).read().strip().splitlines()[-1])
                out["n1_anchor"] = {"value": an["value"], "ms_per_step": an["ms_per_step"], "source": "profiles/" + os.path.basename(anchors[-1]),
                                    "command": "python bench.py --gpus 1 --config cfg3", "speedup_vs_n1": qps / an["value"] if an["value"] else None}
            except Exception:
                pass
    if alt is not None:
        out["alt_mode"] = alt
    if dev_elapsed:
        out["io"] = "fp_search: host query buffer in, host results out (index resident in HBM); %d KB up and %d KB down per batch" % (
            a.batch * a.qlen * a.dim * 2 // 1024, a.batch * a.topk * 12 // 1024)
        out["value_device_io"] = {"value": a.batch * a.steps / dev_elapsed, "ms_per_step": dev_elapsed / a.steps * 1e3,
                                  "note": "fp_search_device: queries uploaded before the timed region, results left in HBM"}
    if primary in ("single", "replica") and rank == 0:   # (replica: rank 0's own kernels; every rank runs the same pipeline)
        stages = {k: v / a.steps for k, v in stage_acc.items()}
        out["stages_ms"] = {k: round(v, 4) for k, v in stages.items()}
        out["stages_note"] = ("HIP-event stage times of the same K steps repeated with graph replay off (in the timed loop a step is "
                              "one hipGraphLaunch, which records no per-stage events)")
        out["docs_repaired_per_batch"] = res["repaired_total"] / a.steps
        Rr = max(a.nfull // 4, 1)
        rer_docs = a.batch * min(Rr, a.docs)
        pr = a.dim * a.nbits // 8
        # per exact-scored token D*nbits/8 + 4 B, per (query,doc) 20 B, centroid table + query tile once per batch
        bytes_maxsim = rer_docs * (a.doc_len * (pr + 4) + 20) + C * a.dim * 2 + a.batch * a.qlen * a.dim * 2
        t_ms = stages.get("S6+S7 maxsim", 0.0)
        ach = bytes_maxsim / (t_ms * 1e-3) / 1e9 if t_ms > 0 else 0.0
        # The `roofline` object describes the DOMINANT kernel of the step.  Both candidates use SURVEY 8d's algorithmic bytes:
        #   S4's per-candidate kernel (k_approx_q8 bounds, or k_approx when the bound stage is off): 4 B per candidate token;
        #   the fused decompress + MaxSim kernel (the north star's roofline target): 68 B per exact-scored token + 20 B per
        #   (query, doc) + the centroid table and the query tile once.
        # Durations are HIP-event times of exactly those single kernels on the search stream (fp_last_search_timings).
        cand_per_step = cand_total / a.steps
        t_ap = stages.get("S4 approx", 0.0)
        b_ap = cand_per_step * a.doc_len * 4
        used_bounds = exact_total < cand_total
        l0 = res.get("s4_form") == "l0"   # (the library reports which form of S4 the stage passes ran)
        l0h = res.get("s4_form") == "l0h"
        ap_name = "k_l0_scan" if l0 else ("k_l0h_scan" if l0h else ("k_approx_q8" if used_bounds else "k_approx"))
        ap_what = {"k_l0_scan": " (S4 level 0: upper bound of every candidate from its code list and a per-centroid byte table in LDS)",
                   "k_l0h_scan": " (S4 level 0 over the hot codes: per-column maxima of a candidate's hot codes, byte table in LDS, rows of S gathered for the hot codes only)",
                   "k_approx_q8": " (S4: 8-bit bounds of every candidate)", "k_approx": " (S4: exact approximate score of every candidate)"}[ap_name]
        ap_traffic, ap_src = _pmc_traffic(ap_name, default_cfg)
        ucodes_per_doc = index.n_unique_codes / max(index.n_docs, 1)
        ach_ap = b_ap / (t_ap * 1e-3) / 1e9 if t_ap > 0 else 0.0
        r_ap = {"kernel": ap_name + ap_what, "bound": "hbm", "achieved": ach_ap, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach_ap / HBM_PEAK_GBS, "traffic": ap_traffic, "traffic_source": ap_src,
                "algorithmic_bytes_per_launch": b_ap,
                "algorithmic_bytes_note": "SURVEY 8d accounting: 4 B (int32 code) per candidate document token",
                "avg_launch_ms": t_ap, "share_of_step": t_ap / ms_per_step if ms_per_step > 0 else 0.0,
                "candidate_docs_per_batch": cand_per_step, "docs_rescored_exactly_per_batch": exact_total / a.steps,
                "prepare_ms": stages.get("S4 prepare", 0.0), "refine_ms": stages.get("S4 refine", 0.0)}
        if l0:
            # Level 0 does not read the token codes SURVEY 8d counts: a candidate costs its document's UNIQUE codes packed at 20
            # bits into whole 128-byte lines (48 codes per line; a document's first line is addressed by its id), 4 B of id and 2 B of bound written.  Those
            # are the algorithmic bytes of the stage as built and what `achieved` / `frac` are quoted on (PMC `traffic` agrees
            # with them); the reference-algorithm accounting is kept beside it -- its "rate" exceeds the HBM peak because the
            # bytes are simply not moved any more.
            lines_per_doc = index.n_code_lines / max(index.n_docs, 1)   # corpus average (40 codes of 24 bits per line)
            b_l0 = cand_per_step * (128.0 * lines_per_doc + 6.0)
            ach_l0 = b_l0 / (t_ap * 1e-3) / 1e9 if t_ap > 0 else 0.0
            r_ap.update({"achieved": ach_l0, "frac": ach_l0 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": b_l0,
                         "algorithmic_bytes_note": "per candidate: %.2f packed-code lines of 128 B (%.1f unique codes at 20 bits, corpus average) + 4 B id "
                                                   "+ 2 B bound" % (lines_per_doc, ucodes_per_doc),
                         "survey_8d_accounting": {"bytes_per_launch": b_ap, "equivalent_GBps": ach_ap, "equivalent_frac": ach_ap / HBM_PEAK_GBS,
                                                  "note": "4 B (int32 code) per candidate document token, the reference algorithm's traffic"}})
        if l0h:
            # the stage as built: a candidate's unique-code list (4 B per code) + id + bound; the gathered rows of S depend on the
            # query (hot codes only) and are not counted
            b_h = cand_per_step * (4.0 * ucodes_per_doc + 6.0)
            ach_h = b_h / (t_ap * 1e-3) / 1e9 if t_ap > 0 else 0.0
            r_ap.update({"achieved": ach_h, "frac": ach_h / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": b_h,
                         "algorithmic_bytes_note": "per candidate: %.1f unique codes x 4 B + 4 B id + 2 B bound (the 64-byte rows of S gathered for the "
                                                   "hot codes are not counted)" % ucodes_per_doc,
                         "survey_8d_accounting": {"bytes_per_launch": b_ap, "equivalent_GBps": ach_ap, "equivalent_frac": ach_ap / HBM_PEAK_GBS,
                                                  "note": "4 B (int32 code) per candidate document token, the reference algorithm's traffic"}})
        traffic, traffic_src = _pmc_traffic("k_maxsim", default_cfg)
        t_ms = stages.get("S6+S7 maxsim", 0.0)   # the MaxSim kernel alone; the exact-order repair behind it is its own stage
        ach = bytes_maxsim / (t_ms * 1e-3) / 1e9 if t_ms > 0 else 0.0
        r_ms = {"kernel": "k_maxsim6 (S6+S7: fused decompress + exact MaxSim; the north star's roofline target)", "bound": "hbm",
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": traffic_src, "algorithmic_bytes_per_launch": bytes_maxsim, "avg_launch_ms": t_ms,
                "share_of_step": t_ms / ms_per_step if ms_per_step > 0 else 0.0,
                "order_repair_ms": stages.get("S7 order repair", 0.0),
                "mfma_tflops": 2.0 * a.dim * a.qlen * rer_docs * a.doc_len / (t_ms * 1e-3) / 1e12 if t_ms > 0 else 0.0}
        # the MaxSim stage is not finished by its MFMA pass: the near-tie marking and the exact-order repair follow ("S7 order
        # repair"); the same bytes over both stages is what the stage as a whole achieves
        t_rep = stages.get("S7 order repair", 0.0)
        ach_wr = bytes_maxsim / ((t_ms + t_rep) * 1e-3) / 1e9 if t_ms + t_rep > 0 else 0.0
        r_ms.update({"achieved_with_repair": ach_wr, "frac_with_repair": ach_wr / HBM_PEAK_GBS, "stage_with_repair_ms": t_ms + t_rep})
        # S1, the centroid GEMM: "S1 centroid_gemm" is the main kernel alone (the sampled pre-pass and the floors are their own stage)
        t_s1 = stages.get("S1 centroid_gemm", 0.0)
        fl_s1 = 2.0 * C * a.dim * a.batch * a.qlen
        ach_s1 = fl_s1 / (t_s1 * 1e-3) / 1e12 if t_s1 > 0 else 0.0
        s1_traffic, s1_src = _pmc_traffic("k_centroid_scores_stream", default_cfg)
        b_s1 = C * a.dim * 2 + a.batch * a.qlen * a.dim * 2 + C * a.batch * (((a.qlen + 31) // 32 * 32) if not (64 < a.qlen <= 128) else 128) * 2   # table + queries + S written
        r_s1 = {"kernel": "k_centroid_scores_stream (S1: fp16 MFMA GEMM centroids x queries; stores the upper candidates h(x + u), "
                          "level 0's excess bytes and the probe's column maxima leave with the tile)",
                "bound": "mfma", "achieved": ach_s1, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_s1 / MFMA_F16_PEAK_TFLOPS,
                "traffic": s1_traffic, "traffic_source": s1_src, "flops_per_launch": fl_s1, "avg_launch_ms": t_s1,
                "share_of_step": t_s1 / ms_per_step if ms_per_step > 0 else 0.0,
                "algorithmic_bytes_per_launch": b_s1, "hbm_frac_on_algorithmic_bytes": (b_s1 / (t_s1 * 1e-3) / 1e9 / HBM_PEAK_GBS) if t_s1 > 0 else 0.0,
                "prepass_floors_ms": stages.get("S1 prepass+floors", 0.0), "s1_form": res.get("s1_form")}
        # `roofline` = the kernel with the longest average launch among the three single-kernel stages (chosen by measured time,
        # whichever it is); all three are always under `roofline_by_kernel`
        cands = [(t_s1, "k_centroid_scores", r_s1), (t_ap, ap_name, r_ap), (t_ms, "k_maxsim", r_ms)]
        cands.sort(key=lambda x: -x[0])
        out["roofline"] = cands[0][2]
        out["roofline_by_kernel"] = {"k_centroid_scores": r_s1, ap_name: r_ap, "k_maxsim": r_ms}
        # ---- CPU baseline: the plain-C oracle ("port"), all host cores, same corpus, bounded sample
        ncpu = a.cpu_queries if not use_dist else 0   # rank 0 at N=1 only
        if ncpu < 0 and a.docs * a.doc_len > 400_000_000:
            ncpu = 0   # the C oracle runs on a host copy of the corpus: not for the 10M-document anchor run (97 GB)
        if ncpu != 0:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import plaid_oracle as OC
            cores = OC.num_procs()
            if ncpu < 0:   # one query per host core (OpenMP over queries, dynamic schedule), at least one batch
                ncpu = min(max(a.batch, cores), a.batch * n_batches, 512)
            arr = R.export_index_arrays(index, centroids=cent, bucket_weights=bw)
            orc = OC.OracleIndex(nbits=spec.nbits, centroids=cent, bucket_weights=bw, ivf=arr["ivf"], ivf_lengths=arr["ivf_lengths"],
                                 doc_codes=arr["doc_codes"], doc_residuals=arr["doc_residuals"], doc_lengths=arr["doc_lengths"])
            qs_all = np.concatenate(batches[: (ncpu + a.batch - 1) // a.batch], axis=0)[:ncpu]
            threads = min(cores, ncpu)
            tc = time.perf_counter()
            ref = orc.search(qs_all, a.topk, a.nfull, a.nprobe, nthreads=threads)
            tcpu = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": ncpu / tcpu, "unit": "queries/s", "cores": threads, "host_cores": cores, "kind": "port",
                                   "sample": "%d queries (%d batches of the timed workload) on the same corpus, plain-C restatement of the reference "
                                             "(oracle/plaid_oracle.c), one query per OpenMP thread, %d threads busy on %d host cores; %.1f s wall" % (
                                                 ncpu, (ncpu + a.batch - 1) // a.batch, threads, cores, tcpu)}
            ncpu = min(ncpu, a.batch)
            qs = qs_all[:ncpu]
            # parity spot check on the benchmark corpus itself (ids modulo near-ties, scores within 1e-3)
            gp, gs, gc = R.search_arrays(index, qs, params)
            ident = sum(int(np.array_equal(gp[b, : gc[b]], ref[b][0])) for b in range(ncpu))
            overlap = float(np.mean([len(set(gp[b, : gc[b]].tolist()) & set(ref[b][0].tolist())) / max(len(ref[b][0]), 1) for b in range(ncpu)]))
            md = 0.0
            for b in range(ncpu):
                rm = dict(zip(ref[b][0].tolist(), ref[b][1].tolist()))
                for p, s in zip(gp[b, : gc[b]].tolist(), gs[b, : gc[b]].tolist()):
                    if p in rm:
                        md = max(md, abs(rm[p] - s))
            out["parity_vs_cpu"] = {"queries": ncpu, "identical_id_lists": ident, "mean_id_overlap": overlap, "max_abs_score_diff": md}
    if rank == 0:
        sys.stdout.flush()
        os.write(_real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
