#!/bin/bash
# SQ / LDS / TA counters of one kernel ($1 regex) in bench.py's default workload, one counter group per pass
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PAT=${1:-k_l0_scan}
OUT=$R/gpurun_out/pmc_s
rm -rf $OUT; mkdir -p $OUT
CMD="python $R/bench.py --steps 2 --warmup 1 --cpu-queries 0"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/p$i -o run -- $CMD >> $OUT/p.log 2>&1
done
python - <<PY
import csv, glob, collections, re
for p in sorted(glob.glob("$OUT/p*/**/run_counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"].split("(")[0][:40]
        if not re.search(r"$PAT", k): continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
    for k in acc:
        print(k, {c: round(v/max(n[(k,c)],1),1) for c,v in acc[k].items()})
PY
grep -i "error\|invalid\|not found" $OUT/p.log | sort | uniq -c | head -8
rm -rf $OUT/p[0-9]*
