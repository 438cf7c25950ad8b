#!/bin/bash
# S1 epilogue with less VALU work (raw packed max, scalar address bases, one-rounding floor): kernel time per variant, pruning
# statistics, ablation floors, GPU suite, default bench line (parity against the CPU oracle)
R=$GRAFT_REPO_ROOT
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
FP_S1_WAVES=4 FP_S1_STREAM=0 kt "stream=0 waves=4"
cp $R/fast-plaid_amd/libfastplaid_hip.so /tmp/lib_orig.so
for f in $R/tools/libs/lib_s1abl*.so; do
  cp $f $R/fast-plaid_amd/libfastplaid_hip.so
  kt "$(basename $f)"
  FP_S1_STREAM=0 kt "$(basename $f) stream=0"
done
cp /tmp/lib_orig.so $R/fast-plaid_amd/libfastplaid_hip.so
cd $R && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 400 python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default bench', d['value'], d['ms_per_step'], d.get('parity_vs_cpu'), d['roofline'].get('docs_rescored_exactly_per_batch'))"
