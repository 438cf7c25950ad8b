#!/bin/bash
# kernel time of $1 (regex) under rocprofv3 for each library variant in tools/libs/ (ablation builds: timing only)
R=$GRAFT_REPO_ROOT
PAT=${1:-k_approx_q8}
cd /tmp && export TMPDIR=/tmp
cp $R/fast-plaid_amd/libfastplaid_hip.so /tmp/lib_orig.so
for f in /tmp/lib_orig.so $R/tools/libs/lib_*.so; do
  cp $f $R/fast-plaid_amd/libfastplaid_hip.so 2>/dev/null
  rm -rf /tmp/abl_prof
  timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_prof -o run -- python $R/bench.py --steps 5 --warmup 2 --cpu-queries 0 > /dev/null 2>&1
  python - <<PY
import csv, re
for r in csv.DictReader(open("/tmp/abl_prof/run_kernel_stats.csv")):
    if re.search(r"$PAT", r["Name"]): print("$(basename $f)", r["Name"].split("(")[0][:40], "avg_us=%.1f calls=%s" % (float(r["AverageNs"])/1e3, r["Calls"]))
PY
done
cp /tmp/lib_orig.so $R/fast-plaid_amd/libfastplaid_hip.so
