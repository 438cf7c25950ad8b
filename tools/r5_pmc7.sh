#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc7
rm -rf $OUT; mkdir -p $OUT
CMD="python $R/bench.py --steps 2 --warmup 1 --cpu-queries 0"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  FP_GRAPH=0 timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/p$i -o run -- $CMD >> $OUT/p.log 2>&1
done
python - <<PY
import csv, glob, collections, re
for p in sorted(glob.glob("$OUT/p*/**/run_counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"].split("(")[0][:40]
        if not re.search(r"k_maxsim7|k_maxsim6|k_l0_scan", k): continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
    for k in acc:
        print(k, {c: round(v/max(n[(k,c)],1),1) for c,v in acc[k].items()})
PY
grep -i "error\|invalid\|not found" $OUT/p.log | sort | uniq -c | head -8
rm -rf $OUT/p[0-9]*
