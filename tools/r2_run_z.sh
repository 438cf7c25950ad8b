#!/bin/bash
TAG=${1:-r02_z}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
for b in 8 1; do
timeout 300 python bench.py --batch $b --cpu-queries 0 --steps 30 --warmup 5 --workload "cfg2_b$b" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('batch $b', 'qps=%.1f ms/batch=%.3f' % (d['value'], d['ms_per_step']), d['stages_ms'])"
done
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --batch 8 --steps 6 --warmup 2 --cpu-queries 0 --workload cfg2_b8"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_prof -o run -- $CMD > $OUT/${TAG}_prof.log 2>&1
python - <<PY
import csv, glob, re
p = (glob.glob("$OUT/${TAG}_prof/*/run_kernel_trace.csv") + glob.glob("$OUT/${TAG}_prof/run_kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"(k_\w+(<[^>]*>)?)", n)
    return m.group(1) if m else n[:40]
idx = [i for i, r in enumerate(rows) if "k_pack_queries" in r["Kernel_Name"]]
start = idx[-2]; end = idx[-1]
t0 = int(rows[start]["Start_Timestamp"])
busy = 0
for r in rows[start:end]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; busy += d
    print("%9.1f us  +%7.1f  %s" % (d, (int(r["Start_Timestamp"]) - t0) / 1e3, short(r["Kernel_Name"])))
print("kernels", end - start, "busy us", busy, "span us", (int(rows[end]["Start_Timestamp"]) - t0) / 1e3)
PY
rm -rf $OUT/${TAG}_prof
