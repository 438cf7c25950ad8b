#!/bin/bash
TAG=${1:-r02_e}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 300 python tools/debug_parity.py > $OUT/${TAG}_debug_parity.txt 2>&1
tail -60 $OUT/${TAG}_debug_parity.txt | cut -c1-700
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $OUT/${TAG}_gpu_tests.log
cat $OUT/${TAG}_gpu_tests.log
timeout 400 python bench.py --cpu-queries 64 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -3 $OUT/${TAG}_bench.err
# native RCCL path with one rank vs the single-GPU line
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --force-dist --dist-mode shard --no-alt-mode --steps 10 --warmup 3 --cpu-queries 0 > $OUT/${TAG}_bench_dist1_native.json 2>> $OUT/${TAG}_bench.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 1 --force-dist --dist-mode shard --dist-impl torch --no-alt-mode --steps 10 --warmup 3 --cpu-queries 0 > $OUT/${TAG}_bench_dist1_torch.json 2>> $OUT/${TAG}_bench.err
tail -5 $OUT/${TAG}_bench.err
python - <<PY
import json, glob, os
for p in sorted(glob.glob("$OUT/${TAG}_bench*.json")):
    try:
        d = json.load(open(p))
    except Exception as e:
        print(os.path.basename(p), "unreadable", e); continue
    st = d.get("stages_ms", {})
    print(os.path.basename(p), "qps %.0f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in st.items()},
          "exact-rescored", d.get("roofline", {}).get("docs_rescored_exactly_per_batch"), "repaired", d.get("docs_repaired_per_batch"), d.get("parity_vs_cpu"), d.get("config", {}).get("parallelism"))
PY
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 5 --cpu-queries 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o run -- $CMD > $OUT/${TAG}_prof.log 2>&1
python $R/tools/summarize_prof.py $(ls $OUT/${TAG}_prof/*/run_kernel_stats.csv $OUT/${TAG}_prof/run_kernel_stats.csv 2>/dev/null | head -1) \
    $OUT/${TAG}_kernel_stats.csv "bench.py --steps 20 --warmup 5 (cfg2: 1M docs, 64 queries/batch), MI355X"
head -24 $OUT/${TAG}_kernel_stats.csv
