#!/bin/bash
# round 6, half-batch overlap (NO-GO, profiles/r06_split_go_nogo.txt): needs the library built with tools/probe/r6_split_overlap.patch applied.
# (1) the graph worker with every batch split, (2) the bench line without / with the overlap and with different edges between the halves on the SAME box.   usage: bash tools/r6_split.sh TAG
TAG=${1:-r6split}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R; mkdir -p $OUT
FP_GRAPH=1 FP_TEST=split_min=2 timeout 600 python tests/graph_worker.py 2>&1 | tail -2
FP_GRAPH=1 FP_TEST=split_min=2,spec_cap_pct=50 timeout 600 python tests/graph_worker.py 2>&1 | tail -2
for e in "FP_SPLIT=0" "FP_SPLIT=1" "FP_TEST=split_dep_a=-1" "FP_TEST=split_dep_a=2,split_dep_b=1" "FP_TEST=split_dep_a=7,split_dep_b=7" "FP_TEST=split_dep_a=4,split_dep_b=2" "FP_SPLIT=1 FP_GRAPH=0" "FP_SPLIT=0"; do
  echo "== $e"
  env $e timeout 300 python bench.py --cpu-queries 0 --steps 40 --warmup 8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'p50', round(d.get('p50_ms',0),4), 'dev', round(d['value_device_io']['ms_per_step'],4), 'sum stages', round(sum(d['stages_ms'].values()),4))"
done | tee $OUT/${TAG}_ab.txt
