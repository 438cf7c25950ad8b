#!/bin/bash
# round-2 run B: GPU tests, bench (level-0 v2 + MaxSim v5 + exact-order repair), small sweeps, kernel trace
TAG=${1:-r02_b}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/${TAG}_gpu_tests.log
cat $OUT/${TAG}_gpu_tests.log
timeout 400 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -3 $OUT/${TAG}_bench.err
FP_MAXSIM_REPAIR=0 timeout 200 python bench.py --cpu-queries 0 --steps 10 --warmup 3 > $OUT/${TAG}_bench_norepair.json 2>> $OUT/${TAG}_bench.err
FP_MAXSIM_REPAIR=2 timeout 200 python bench.py --cpu-queries 0 --steps 10 --warmup 3 > $OUT/${TAG}_bench_repairall.json 2>> $OUT/${TAG}_bench.err
for pilot in 1 2 8; do
  FP_L0_PILOT=$pilot timeout 200 python bench.py --cpu-queries 0 --steps 10 --warmup 3 > $OUT/${TAG}_bench_pilot$pilot.json 2>> $OUT/${TAG}_bench.err
done
for cpw in 1024 2048 8192; do
  FP_L0_CPW=$cpw timeout 200 python bench.py --cpu-queries 0 --steps 10 --warmup 3 > $OUT/${TAG}_bench_cpw$cpw.json 2>> $OUT/${TAG}_bench.err
done
for tail in 0.015 0.035; do
  FP_L0_TAIL=$tail timeout 200 python bench.py --cpu-queries 0 --steps 10 --warmup 3 > $OUT/${TAG}_bench_tail$tail.json 2>> $OUT/${TAG}_bench.err
done
python - <<PY
import json, glob, os
for p in sorted(glob.glob("$OUT/${TAG}_bench*.json")):
    try:
        d = json.load(open(p))
    except Exception as e:
        print(os.path.basename(p), "unreadable", e); continue
    st = d.get("stages_ms", {})
    print(os.path.basename(p), "qps %.0f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in st.items()},
          "exact-rescored", d.get("roofline", {}).get("docs_rescored_exactly_per_batch"), d.get("parity_vs_cpu"))
PY
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 5 --cpu-queries 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o run -- $CMD > $OUT/${TAG}_prof.log 2>&1
python $R/tools/summarize_prof.py $(ls $OUT/${TAG}_prof/*/run_kernel_stats.csv $OUT/${TAG}_prof/run_kernel_stats.csv 2>/dev/null | head -1) \
    $OUT/${TAG}_kernel_stats.csv "bench.py --steps 20 --warmup 5 (cfg2: 1M docs, 64 queries/batch), MI355X"
head -42 $OUT/${TAG}_kernel_stats.csv
