#!/bin/bash
# round-2 run D: GPU tests, bench, exact-order repair sweep (flag window / repair mode), level-0 scan ablations + PMC
TAG=${1:-r02_d}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/${TAG}_gpu_tests.log
cat $OUT/${TAG}_gpu_tests.log
timeout 400 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -3 $OUT/${TAG}_bench.err
B="timeout 300 python bench.py --steps 8 --warmup 2 --cpu-queries 64"
for eps in 3.8e-6 7.6e-6 1.53e-5; do
  FP_MAXSIM_EPS=$eps $B > $OUT/${TAG}_bench_eps$eps.json 2>> $OUT/${TAG}_bench.err
done
FP_MAXSIM_REPAIR=2 $B > $OUT/${TAG}_bench_repairall.json 2>> $OUT/${TAG}_bench.err
FP_MAXSIM_REPAIR=2 FP_MAXSIM_EPS=1.53e-5 $B > $OUT/${TAG}_bench_repairall_eps1.53e-5.json 2>> $OUT/${TAG}_bench.err
B2="timeout 200 python bench.py --steps 8 --warmup 2 --cpu-queries 0"
for abl in 1 2 3; do
  FP_L0_ABL=$abl $B2 > $OUT/${TAG}_bench_l0abl$abl.json 2>> $OUT/${TAG}_bench.err
done
FP_MAXSIM_REPAIR=0 $B2 > $OUT/${TAG}_bench_norepair.json 2>> $OUT/${TAG}_bench.err
python - <<PY
import json, glob, os
for p in sorted(glob.glob("$OUT/${TAG}_bench*.json")):
    try:
        d = json.load(open(p))
    except Exception as e:
        print(os.path.basename(p), "unreadable", e); continue
    st = d.get("stages_ms", {})
    print(os.path.basename(p), "qps %.0f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in st.items()},
          "exact-rescored", d.get("roofline", {}).get("docs_rescored_exactly_per_batch"), "repaired", d.get("docs_repaired_per_batch"), d.get("parity_vs_cpu"))
PY
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 5 --cpu-queries 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o run -- $CMD > $OUT/${TAG}_prof.log 2>&1
python $R/tools/summarize_prof.py $(ls $OUT/${TAG}_prof/*/run_kernel_stats.csv $OUT/${TAG}_prof/run_kernel_stats.csv 2>/dev/null | head -1) \
    $OUT/${TAG}_kernel_stats.csv "bench.py --steps 20 --warmup 5 (cfg2: 1M docs, 64 queries/batch), MI355X"
head -30 $OUT/${TAG}_kernel_stats.csv
bash $R/tools/pmc_kernel.sh "k_l0_scan|k_maxsim5|k_approx" 2>&1 | tail -12
