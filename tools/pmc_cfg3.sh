#!/bin/bash
# FETCH_SIZE / L2 hit rate of the level-0 scan at cfg3 (plain launches), FP_L0_MULTI=$1
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; export FP_GRAPH=0 FP_L0_MULTI=${1:-1}
rm -rf /tmp/pm3; timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d /tmp/pm3 -o run -- python $R/bench.py --config cfg3 --steps 2 --warmup 1 --cpu-queries 0 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for p in glob.glob("/tmp/pm3/**/run_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:40]
        if not (k.startswith("k_l0_scan") or k.startswith("k_l0_combine")): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, cs in acc.items():
    d = {c: v / max(n[(k, c)], 1) for c, v in cs.items()}
    print(k, "launches", n[(k, "FETCH_SIZE")], "fetch GB/launch (x2 corrected) %.1f" % (d["FETCH_SIZE"] * 1024 * 2 / 1e9), "L2 hit %.3f" % (d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d["TCC_MISS_sum"], 1)))
PY
