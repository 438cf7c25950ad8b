#!/bin/bash
TAG=${1:-r02_j}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
for cpw in 8192 16384 32768; do
  FP_L0_CPW=$cpw timeout 300 python bench.py --cpu-queries 0 > $OUT/${TAG}_bench_cpw$cpw.json 2> /dev/null
done
python - <<PY
import json, glob, os
for p in sorted(glob.glob("$OUT/${TAG}_bench*.json")):
    try:
        d = json.load(open(p))
    except Exception as e:
        print(os.path.basename(p), "unreadable", e); continue
    st = d.get("stages_ms", {})
    print(os.path.basename(p), "qps %.0f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in st.items()})
PY
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 5 --cpu-queries 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o run -- $CMD > $OUT/${TAG}_prof.log 2>&1
python $R/tools/summarize_prof.py $(ls $OUT/${TAG}_prof/*/run_kernel_stats.csv $OUT/${TAG}_prof/run_kernel_stats.csv 2>/dev/null | head -1) \
    $OUT/${TAG}_kernel_stats.csv "bench.py --steps 20 --warmup 5 (cfg2: 1M docs, 64 queries/batch), MI355X"
head -50 $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/${TAG}_prof
