#!/bin/bash
# S1 epilogue rewrite, second pass: GPU suite first, kernel times, pruning statistics, PMC of the streaming kernel, small batches
R=$GRAFT_REPO_ROOT
cd $R && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
kt() {
  rm -rf /tmp/p_at
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_at -o run -- python $R/bench.py --steps 5 --warmup 2 --cpu-queries 0 > /tmp/kt.out 2>/tmp/kt.err
  python - <<PY
import csv, json
try:
    for r in csv.DictReader(open("/tmp/p_at/run_kernel_stats.csv")):
        if "k_centroid_scores" in r["Name"]: print("  $1", r["Name"].split("(")[0][:48], "avg_us=%.1f calls=%s" % (float(r["AverageNs"])/1e3, r["Calls"]))
except Exception as e:
    print("  $1 FAILED", e); print(open("/tmp/kt.err").read()[-1500:])
PY
}
for v in 0 1; do
  FP_S1_STREAM=$v kt "stream=$v"
  FP_S1_STREAM=$v timeout 200 python $R/bench.py --steps 30 --warmup 5 --cpu-queries 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  stream=$v value', d['value'], 'ms', d['ms_per_step'], 'rescored', d['roofline'].get('docs_rescored_exactly_per_batch'), d.get('stages_ms'))"
done
for b in 1 8; do
  timeout 200 python $R/bench.py --batch $b --steps 40 --warmup 5 --cpu-queries 0 --workload cfg2_b$b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  B=$b ms', d['ms_per_step'], d.get('stages_ms'))"
done
bash $R/tools/pmc_scan.sh k_centroid_scores_stream
cd /tmp
rm -rf /tmp/p_m; timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_m -o run -- python $R/bench.py --steps 2 --warmup 1 --cpu-queries 0 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for p in glob.glob("/tmp/p_m/**/run_counter_collection.csv", recursive=True):
    acc=collections.defaultdict(float); n=collections.defaultdict(int)
    for r in csv.DictReader(open(p)):
        if "k_centroid_scores_stream" in r["Kernel_Name"]: acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    print("mfma", {c: round(v/n[c],1) for c,v in acc.items()})
PY
cd $R && timeout 400 python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default bench', d['value'], d['ms_per_step'], d.get('parity_vs_cpu'), d['roofline'].get('docs_rescored_exactly_per_batch'))"
