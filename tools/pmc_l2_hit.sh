#!/bin/bash
# L2 hit rate + fabric fetch of k_approx (and its duration) for the current library; env passes through
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_l2_${1:-x}
mkdir -p $OUT
CMD="python $R/bench.py --steps 2 --warmup 1 --cpu-queries 0"
timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/h -o run -- $CMD > $OUT/p.log 2>&1
timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/f -o run -- $CMD >> $OUT/p.log 2>&1
python - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/*/run_counter_collection.csv")):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"].split("(")[0][:40]
        if "k_approx" not in k and "k_maxsim" not in k: continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
    for k in acc:
        d={c: v/max(n[(k,c)],1) for c,v in acc[k].items()}
        if "TCC_REQ_sum" in d: print("${1:-x}", k, "L2 hit=%.3f req=%.1fM" % (d["TCC_HIT_sum"]/max(d["TCC_REQ_sum"],1), d["TCC_REQ_sum"]/1e6))
        if "FETCH_SIZE" in d: print("${1:-x}", k, "fetch(corrected)=%.2f GB" % (d["FETCH_SIZE"]*2048/1e9))
PY
