#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
T=${1:-r5e}
( LAZY_EXPECT=1 timeout 600 python tests/lazy_worker.py 2>&1 | tail -2
  for v in "FP_APPROX_IMPL=l0 LAZY_EXPECT=1" "FP_APPROX_IMPL=q8 LAZY_EXPECT=1" "FP_APPROX_IMPL=exact LAZY_EXPECT=1" "FP_TEST=lz_gcap=3 LAZY_EXPECT=0" "FP_S1_EXACT=1 LAZY_EXPECT=0" "FP_S1_STREAM=0 LAZY_EXPECT=1"; do
    echo "== $v"; env $v timeout 600 python tests/lazy_worker.py 2>&1 | tail -2
  done ) > $OUT/${T}_lazy_worker.log 2>&1
grep -c LAZY_OK $OUT/${T}_lazy_worker.log; grep -v LAZY_OK $OUT/${T}_lazy_worker.log | grep -v "^==" | tail -5
timeout 400 python bench.py > $OUT/${T}_bench.json 2> $OUT/${T}_bench.err; echo "bench rc=$?"
python - $T <<'PY'
import json, sys
T = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/{T}_bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "dev", d["value_device_io"]["ms_per_step"], "parity", d.get("parity_vs_cpu"))
    print("stages", d.get("stages_ms"))
    for k, v in d["roofline_by_kernel"].items(): print(k, v.get("frac"), v.get("avg_launch_ms"), v.get("frac_with_repair"), v.get("s1_form"))
except Exception as e:
    print("bench parse failed", e); print(open(f"gpurun_out/{T}_bench.err").read()[-3000:])
PY
cd /tmp && export TMPDIR=/tmp
FP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${T}_prof -o run -- python $R/bench.py --steps 8 --warmup 3 --cpu-queries 0 > $OUT/${T}_prof.log 2>&1
python $R/tools/summarize_prof.py $(ls $OUT/${T}_prof/*/run_kernel_stats.csv $OUT/${T}_prof/run_kernel_stats.csv 2>/dev/null | head -1) $OUT/${T}_kernel_stats.csv "bench.py --steps 8 --warmup 3, FP_GRAPH=0"; grep -E "^k_(lz|sel|approx|l0_|cent|maxsim|probe)" $OUT/${T}_kernel_stats.csv | head -30
rm -rf $OUT/${T}_prof
