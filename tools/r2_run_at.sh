#!/bin/bash
# S1 streaming kernel (LDS-direct loads): DMA semantics probe, kernel time per variant, ablation floors, GPU suite
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 60 $R/tools/probe/lds_dma_probe
kt() {
  rm -rf /tmp/p_at
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_at -o run -- python $R/bench.py --steps 5 --warmup 2 --cpu-queries 0 > /tmp/kt.out 2>/tmp/kt.err
  python - <<PY
import csv, json
try:
    for r in csv.DictReader(open("/tmp/p_at/run_kernel_stats.csv")):
        if "k_centroid_scores" in r["Name"]: print("  $1", r["Name"].split("(")[0][:48], "avg_us=%.1f calls=%s" % (float(r["AverageNs"])/1e3, r["Calls"]))
    d = json.loads(open("/tmp/kt.out").read().strip().splitlines()[-1])
    print("  $1 parity", d.get("parity_vs_cpu"))
except Exception as e:
    print("  $1 FAILED", e); print(open("/tmp/kt.err").read()[-1500:])
PY
}
for v in 0 1 4 16 32; do
  export FP_S1_STREAM=$v
  kt "stream=$v"
done
unset FP_S1_STREAM
for v in 0 1; do
  FP_S1_STREAM=$v timeout 200 python $R/bench.py --steps 30 --warmup 5 --cpu-queries 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  stream=$v value', d['value'], 'ms', d['ms_per_step'], d.get('stages_ms'))"
done
cp $R/fast-plaid_amd/libfastplaid_hip.so /tmp/lib_orig.so
for f in $R/tools/libs/lib_s1abl*.so; do
  cp $f $R/fast-plaid_amd/libfastplaid_hip.so
  kt "$(basename $f)"
  FP_S1_STREAM=0 kt "$(basename $f) stream=0"
done
cp /tmp/lib_orig.so $R/fast-plaid_amd/libfastplaid_hip.so
cd $R && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
