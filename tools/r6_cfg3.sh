#!/bin/bash
# round 6: tables beyond 2^17 centroids on ranges of 2^18 with a 4-bit table (the tree) against 2^17 ranges with the byte table (FP_TEST=l0_rsh=17):
# the level-0 tests that use such tables, then cfg3 on one GPU both ways.   usage: bash tools/r6_cfg3.sh TAG
TAG=${1:-r6cfg3}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "level0 or lazy or full_size_properties" 2>&1 | tail -4
for e in "X=0" "FP_TEST=l0_rsh=17"; do
  echo "== $e"
  env $e timeout 900 python bench.py --gpus 1 --config cfg3 --steps 5 --warmup 2 --cpu-queries 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=[v for v in [d['roofline']]+list(d['roofline_by_kernel'].values()) if 'candidate_docs_per_batch' in v][0]
print('   ms/batch=%.2f' % d['ms_per_step'], 'cand=%d rescored=%d' % (k['candidate_docs_per_batch'], k['docs_rescored_exactly_per_batch']), {k: round(v,2) for k,v in d['stages_ms'].items() if v >= 0.3})"
done 2>&1 | tee $OUT/${TAG}_cfg3.txt
