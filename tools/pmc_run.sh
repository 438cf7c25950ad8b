#!/bin/bash
# PMC passes for the two hot kernels (separate passes; --kernel-trace only, as the pool requires)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
CMD="python $R/bench.py --steps 3 --warmup 1 --cpu-queries 0"
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/p1 -o run -- $CMD > $OUT/p1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -d $OUT/p2 -o run -- $CMD > $OUT/p2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/p3 -o run -- $CMD > $OUT/p3.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/p4 -o run -- $CMD > $OUT/p4.log 2>&1
ls $OUT/*/
python - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/p*/run_counter_collection.csv")):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"].split("(")[0][:40]
        if not any(s in k for s in ("k_maxsim","k_approx","k_centroid","k_probe_partial","k_sel_collect","k_ivf_mark")): continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
    for k in acc:
        print(p.split("/")[-2], k, {c: round(v/max(n[(k,c)],1),1) for c,v in acc[k].items()})
PY
