#!/bin/bash
# kernel timeline of the bench loop (graph replay on): raw kernel trace -> gpurun_out/${TAG}_trace.csv   usage: bash tools/r6_timeline.sh TAG [env...]
TAG=${1:-r6tl}; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trs; env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trs -o run -- python $R/bench.py --steps 6 --warmup 8 --cpu-queries 0 > /tmp/trs.log 2>&1
f=$(find /tmp/trs -name run_kernel_trace.csv | head -1)
python - "$f" > $OUT/${TAG}_trace.csv <<'PY'
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
print("queue,start_us,end_us,dur_us,name")
for r in rows:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:40]
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    print("%s,%.1f,%.1f,%.1f,%s" % (r["Queue_Id"], s, e, e - s, n.replace(",", ";")))
PY
wc -l $OUT/${TAG}_trace.csv; tail -3 /tmp/trs.log
