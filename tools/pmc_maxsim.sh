#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc2
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA_[A-Z_0-9]+|TCP_[A-Z_0-9]+|TD_[A-Z_0-9]+)\b" | sort -u | tr '\n' ' ' > $OUT/counters.txt
for v in v3 occ2; do
cp $R/tools/libs/lib_$v.so $R/fast-plaid_amd/libfastplaid_hip.so
CMD="python $R/bench.py --steps 3 --warmup 1 --cpu-queries 0"
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/${v}_p1 -o run -- $CMD > $OUT/p.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum -d $OUT/${v}_p2 -o run -- $CMD >> $OUT/p.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM -d $OUT/${v}_p3 -o run -- $CMD >> $OUT/p.log 2>&1
done
python - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/*/run_counter_collection.csv")):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"].split("(")[0][:40]
        if "k_maxsim" not in k: continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
    for k in acc:
        print(p.split("/")[-2], {c: round(v/max(n[(k,c)],1),1) for c,v in acc[k].items()})
PY
head -c 1500 $OUT/counters.txt; tail -3 $OUT/p.log
