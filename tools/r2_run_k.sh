#!/bin/bash
TAG=${1:-r02_k}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 6 --warmup 2 --cpu-queries 0"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_prof -o run -- $CMD > $OUT/${TAG}_prof.log 2>&1
python - <<PY
import csv, glob, re
p = (glob.glob("$OUT/${TAG}_prof/*/run_kernel_trace.csv") + glob.glob("$OUT/${TAG}_prof/run_kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"(k_\w+(<[^>]*>)?)", n)
    return m.group(1) if m else n[:40]
# last 3 batches: find k_centroid_scores occurrences
idx = [i for i, r in enumerate(rows) if "k_centroid_scores" in r["Kernel_Name"]]
with open("$OUT/${TAG}_sequence.txt", "w") as f:
    for start in idx[-3:]:
        end = next((j for j in idx if j > start), len(rows))
        t0 = int(rows[start]["Start_Timestamp"])
        f.write("--- batch\n")
        for r in rows[max(start - 3, 0):end]:
            f.write("%9.1f us  +%7.1f  %s\n" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, (int(r["Start_Timestamp"]) - t0) / 1e3, short(r["Kernel_Name"])))
print(open("$OUT/${TAG}_sequence.txt").read()[-6000:])
PY
rm -rf $OUT/${TAG}_prof
