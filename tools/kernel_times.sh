#!/bin/bash
# per-kernel average durations of the bench workload (plain launches), the kernels matching $1 (regex)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; FP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o run -- python $R/bench.py --steps 10 --warmup 3 --cpu-queries 0 ${BENCH_ARGS} > /dev/null 2>&1
python - "$1" <<'PY'
import csv, glob, re, sys
f = glob.glob("/tmp/kt/**/run_kernel_stats.csv", recursive=True)[0]
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else ".")
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")[:50]
    if pat.search(n) and int(r["Calls"]) >= 10:
        print("%-52s calls %4s avg %8.1f us  min %8.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
