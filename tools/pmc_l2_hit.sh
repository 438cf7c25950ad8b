#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_ap3
mkdir -p $OUT
i=0
for bt in 1 4; do
for v in "FP_APPROX_IMPL=flat" "FP_APPROX_NR=4 FP_APPROX_DPQ=8" "FP_APPROX_NR=8 FP_APPROX_DPQ=8"; do
  i=$((i+1))
  CMD="python $R/bench.py --steps 2 --warmup 1 --cpu-queries 0 --batch $bt"
  env $v timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/b${bt}_v${i}_h -o run -- $CMD >> $OUT/p.log 2>&1
done; done
python - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/*/run_counter_collection.csv")):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"].split("(")[0][:40]
        if "k_approx" not in k: continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
    for k in acc:
        d={c: v/max(n[(k,c)],1) for c,v in acc[k].items()}
        print(p.split("/")[-2], k, "hit=%.3f req=%.1fM" % (d["TCC_HIT_sum"]/max(d["TCC_REQ_sum"],1), d["TCC_REQ_sum"]/1e6))
    # kernel durations from the trace
    t=collections.defaultdict(list)
    for r in csv.DictReader(open(p.replace("counter_collection","kernel_trace"))):
        if "k_approx" in r["Kernel_Name"]: t[r["Kernel_Name"].split("(")[0][:30]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    print("   dur_us", {k: [round(x) for x in v] for k,v in t.items()})
PY
grep -i "error code" $OUT/p.log | head -3
