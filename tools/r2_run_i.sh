#!/bin/bash
TAG=${1:-r02_i}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 120 ./tools/probe/chain_probe 2>&1 | tee $OUT/${TAG}_chain_probe.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -150 > $OUT/${TAG}_gpu_tests.log
grep -E "^E   |passed|failed|^FAILED" $OUT/${TAG}_gpu_tests.log | cut -c1-300 | head -60
timeout 600 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
FP_L0_PILOT=2 timeout 400 python bench.py --cpu-queries 0 > $OUT/${TAG}_bench_pilot2.json 2> /dev/null
FP_L0_PILOT=3 timeout 400 python bench.py --cpu-queries 0 > $OUT/${TAG}_bench_pilot3.json 2> /dev/null
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --force-dist --dist-mode shard --no-alt-mode --steps 20 --warmup 5 --cpu-queries 0 > $OUT/${TAG}_bench_dist1_native.json 2> $OUT/${TAG}_dist_native.err
grep -E "rank0\]:|Error" $OUT/${TAG}_dist_native.err | head -10 | cut -c1-300
python - <<PY
import json, glob, os
for p in sorted(glob.glob("$OUT/${TAG}_bench*.json")):
    try:
        d = json.load(open(p))
    except Exception as e:
        print(os.path.basename(p), "unreadable", e); continue
    st = d.get("stages_ms", {})
    print(os.path.basename(p), "qps %.0f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in st.items()},
          "repaired", d.get("docs_repaired_per_batch"), d.get("config", {}).get("parallelism"), d.get("parity_vs_cpu"), d.get("cpu_baseline"))
PY
