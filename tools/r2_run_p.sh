#!/bin/bash
# checkpoint: full GPU suite, bench (+cpu baseline), variants, kernel stats
TAG=${1:-r02_p}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $OUT/${TAG}_gpu_tests.log
grep -E "^E   |passed|failed|^FAILED" $OUT/${TAG}_gpu_tests.log | cut -c1-300 | head -30
timeout 600 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
FP_L0_PILOT=6 timeout 300 python bench.py --cpu-queries 0 > $OUT/${TAG}_bench_pilot6.json 2> /dev/null
FP_L0_CPW=2048 timeout 300 python bench.py --cpu-queries 0 > $OUT/${TAG}_bench_cpw2048.json 2> /dev/null
FP_L0_CPW=8192 timeout 300 python bench.py --cpu-queries 0 > $OUT/${TAG}_bench_cpw8192.json 2> /dev/null
for abl in 1 2 3; do
  FP_L0_ABL=$abl timeout 300 python bench.py --cpu-queries 0 --steps 10 --warmup 3 > $OUT/${TAG}_bench_abl$abl.json 2> /dev/null
done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --force-dist --dist-mode shard --no-alt-mode --steps 20 --warmup 5 --cpu-queries 0 > $OUT/${TAG}_bench_dist1_native.json 2> $OUT/${TAG}_dist_native.err
python - <<PY
import json, glob, os
for p in sorted(glob.glob("$OUT/${TAG}_bench*.json")):
    try:
        d = json.load(open(p))
    except Exception as e:
        print(os.path.basename(p), "unreadable", e); continue
    st = d.get("stages_ms", {})
    print(os.path.basename(p), "qps %.0f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in st.items()},
          d.get("parity_vs_cpu"), d.get("cpu_baseline"))
PY
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 5 --cpu-queries 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o run -- $CMD > $OUT/${TAG}_prof.log 2>&1
python $R/tools/summarize_prof.py $(ls $OUT/${TAG}_prof/*/run_kernel_stats.csv $OUT/${TAG}_prof/run_kernel_stats.csv 2>/dev/null | head -1) \
    $OUT/${TAG}_kernel_stats.csv "bench.py --steps 20 --warmup 5 (cfg2: 1M docs, 64 queries/batch), MI355X"
head -16 $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/${TAG}_prof
