#!/bin/bash
# round 6: cfg2 with zero-padded query rows (the last 8 of 32 token rows of every query zeroed), the tree's library against variants   usage: bash tools/r6_zero_rows.sh TAG [variant.so ...]
TAG=${1:-r6zr}; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R; mkdir -p $OUT
for lib in "" "$@" "" "$@"; do
  echo "== ${lib:-tree}"
  FP_LIB_PATH=${lib:+$R/$lib} timeout 300 python bench.py --cpu-queries 16 --steps 40 --warmup 8 --zero-rows 8 --workload cfg2_zero_padded 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('  ', round(d['ms_per_step'],4), 'p50', round(d.get('p50_ms',0),4), d.get('parity_vs_cpu'), d.get('s1_form'), {k:round(v,3) for k,v in d['stages_ms'].items()})"
done 2>&1 | tee $OUT/${TAG}_zero_rows.txt
