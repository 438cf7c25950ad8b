#!/bin/bash
TAG=${1:-r02_ae}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -k "shard or native or full_size or golden" 2>&1 | tail -40 > $OUT/${TAG}_tests.log
grep -E "^E   |passed|failed|^FAILED" $OUT/${TAG}_tests.log | cut -c1-300 | head -30
timeout 400 python bench.py --cpu-queries 0 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --force-dist --dist-mode shard --no-alt-mode --steps 20 --warmup 5 --cpu-queries 0 > $OUT/${TAG}_bench_dist1_native.json 2> $OUT/${TAG}_dist_native.err
grep -E "rank0\]:|Error" $OUT/${TAG}_dist_native.err | head -10 | cut -c1-300
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 1 --force-dist --dist-mode shard --dist-impl torch --no-alt-mode --steps 20 --warmup 5 --cpu-queries 0 > $OUT/${TAG}_bench_dist1_torch.json 2> $OUT/${TAG}_dist_torch.err
grep -E "rank0\]:|Error" $OUT/${TAG}_dist_torch.err | head -10 | cut -c1-300
python - <<PY
import json, glob, os
for p in sorted(glob.glob("$OUT/${TAG}_bench*.json")):
    try:
        d = json.load(open(p))
    except Exception as e:
        print(os.path.basename(p), "unreadable", e); continue
    print(os.path.basename(p), "qps %.0f ms %.3f" % (d["value"], d["ms_per_step"]), d.get("config", {}).get("parallelism"))
PY
