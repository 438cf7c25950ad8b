#!/bin/bash
# round 5: lazy worker + bench + per-kernel times of the bench workload (plain launches)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
LAZY_EXPECT=1 timeout 600 python tests/lazy_worker.py > $OUT/r5d_lazy_worker.log 2>&1; echo "lazy_worker rc=$?" >> $OUT/r5d_lazy_worker.log
tail -4 $OUT/r5d_lazy_worker.log
timeout 400 python bench.py > $OUT/r5d_bench.json 2> $OUT/r5d_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5d_bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "parity", d.get("parity_vs_cpu"))
    print("stages", d.get("stages_ms"))
    for k, v in d["roofline_by_kernel"].items(): print(k, v.get("frac"), v.get("avg_launch_ms"), v.get("frac_with_repair"), v.get("s1_form"))
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/r5d_bench.err").read()[-3000:])
PY
cd /tmp && export TMPDIR=/tmp
FP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r5d_prof -o run -- python $R/bench.py --steps 8 --warmup 3 --cpu-queries 0 > $OUT/r5d_prof.log 2>&1
python $R/tools/summarize_prof.py $(ls $OUT/r5d_prof/*/run_kernel_stats.csv $OUT/r5d_prof/run_kernel_stats.csv 2>/dev/null | head -1) $OUT/r5d_kernel_stats.csv "bench.py --steps 8 --warmup 3, FP_GRAPH=0"; grep -v "^#" $OUT/r5d_kernel_stats.csv | head -45
rm -rf $OUT/r5d_prof
