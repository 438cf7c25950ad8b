#!/bin/bash
# PMC passes for kernels matching $1 (regex) in bench.py's default workload; env passes through.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PAT=${1:-k_approx}
OUT=$R/gpurun_out/pmc_k
rm -rf $OUT; mkdir -p $OUT
CMD="python $R/bench.py --steps 2 --warmup 1 --cpu-queries 0"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_FLAT_READ_WAVEFRONTS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/p$i -o run -- $CMD >> $OUT/p.log 2>&1
done
python - <<PY
import csv, glob, collections, re
for p in sorted(glob.glob("$OUT/p*/run_counter_collection.csv")):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"].split("(")[0][:40]
        if not re.search(r"$PAT", k): continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
    for k in acc:
        print(k, {c: round(v/max(n[(k,c)],1),1) for c,v in acc[k].items()})
PY
grep -i "error code" $OUT/p.log | head -3
