#!/bin/bash
# round 5, first GPU pass of the lazy S1: the lazy worker, the bench line, the lazy form's counters on cfg2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
LAZY_EXPECT=1 timeout 600 python tests/lazy_worker.py > $OUT/r5a_lazy_worker.log 2>&1; echo "lazy_worker rc=$?" >> $OUT/r5a_lazy_worker.log
tail -5 $OUT/r5a_lazy_worker.log
timeout 400 python bench.py > $OUT/r5a_bench.json 2> $OUT/r5a_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5a_bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "parity", d.get("parity_vs_cpu"))
    print("stages", d.get("stages_ms"))
    print("roofline", d["roofline"]["kernel"][:40], d["roofline"]["frac"])
    for k, v in d["roofline_by_kernel"].items(): print(k, v.get("frac"), v.get("avg_launch_ms"), v.get("frac_with_repair"), v.get("s1_form"))
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/r5a_bench.err").read()[-3000:])
PY
FP_S1_STATS=1 FP_GRAPH=0 timeout 300 python tools/s1_stats_cfg2.py 4 > $OUT/r5a_s1_stats.log 2>&1; tail -3 $OUT/r5a_s1_stats.log
