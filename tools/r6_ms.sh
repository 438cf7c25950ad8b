#!/bin/bash
# round 6, MaxSim: parity of the touched kernels, then A/B of library variants on the SAME box (bench line + rocprofv3 kernel time of k_maxsim6 / repair).
# usage: bash tools/r6_ms.sh TAG [variant.so ...]   (variants under tools/libs/, selected with FP_LIB_PATH; the tree's library is always measured)
TAG=${1:-r6ms}; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "arith or maxsim or repair or golden or decompress or synthetic_vs_oracle or randomized or token_score or lazy or full_size_cfg2" 2>&1 | tail -4
kt() {  # kernel times of the bench workload for the library in $1 ("" = the tree's)
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt
  FP_LIB_PATH=$1 FP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o run -- python $R/bench.py --steps 10 --warmup 3 --cpu-queries 0 > /dev/null 2>&1
  python - <<'PY'
import csv, glob, re
f = glob.glob("/tmp/kt/**/run_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")[:50]
    if re.search("k_maxsim|k_final_mark|k_centroid_scores_stream|k_l0_scan|k_approx", n) and int(r["Calls"]) >= 10:
        print("   %-44s calls %4s avg %8.1f us  min %8.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
  cd $R
}
for lib in "" "$@" "" "$@"; do
  echo "== ${lib:-tree}"
  FP_LIB_PATH=${lib:+$R/$lib} timeout 300 python bench.py --cpu-queries 16 --steps 40 --warmup 8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('  ', round(d['ms_per_step'],4), 'p50', round(d.get('p50_ms',0),4), 'dev', round(d['value_device_io']['ms_per_step'],4), d.get('parity_vs_cpu'), {k:round(v,4) for k,v in d['stages_ms'].items() if 'maxsim' in k or 'repair' in k})"
  kt ${lib:+$R/$lib}
done 2>&1 | tee $OUT/${TAG}_ab.txt
