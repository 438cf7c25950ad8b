#!/bin/bash
# S1 with the XCD-aware tile order: kernel time (rocprofv3) and bench value for FP_S1_XCD=0/1, ablation floors, parity subset
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
kt() {  # kernel time of k_centroid_scores under the current env
  rm -rf /tmp/p_as
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_as -o run -- python $R/bench.py --steps 5 --warmup 2 --cpu-queries 0 > /dev/null 2>&1
  python - <<PY
import csv
for r in csv.DictReader(open("/tmp/p_as/run_kernel_stats.csv")):
    if "k_centroid_scores" in r["Name"]: print("  $1", r["Name"].split("(")[0][:48], "avg_us=%.1f calls=%s" % (float(r["AverageNs"])/1e3, r["Calls"]))
PY
}
for x in 0 1; do
  export FP_S1_XCD=$x
  echo "== FP_S1_XCD=$x"
  kt "xcd=$x"
  FP_S1_WAVES=4 kt "xcd=$x waves=4"
  timeout 200 python $R/bench.py --steps 30 --warmup 5 --cpu-queries 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  value', d['value'], 'ms', d['ms_per_step'], {k:round(v,3) for k,v in d.get('stages_ms',{}).items()} if isinstance(d.get('stages_ms'),dict) else '')"
done
unset FP_S1_XCD
cp $R/fast-plaid_amd/libfastplaid_hip.so /tmp/lib_orig.so
for f in $R/tools/libs/lib_s1abl*.so; do
  cp $f $R/fast-plaid_amd/libfastplaid_hip.so
  kt "$(basename $f)"
done
cp /tmp/lib_orig.so $R/fast-plaid_amd/libfastplaid_hip.so
cd $R && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
