#!/bin/bash
# cfg4 stage times under a list of environments (one bench run each)   usage: bash tools/r6_cfg4_knobs.sh TAG "ENV=.." ...
TAG=${1:-r6cfg4k}; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R; mkdir -p $OUT
for e in "$@"; do
  echo "== $e"
  env $e timeout 300 python $R/bench.py --steps 10 --warmup 3 --cpu-queries 0 --workload cfg4 --docs 100000 --doc-len 1024 --batch 32 --topk 100 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
k=[v for v in [d['roofline']]+list(d['roofline_by_kernel'].values()) if 'candidate_docs_per_batch' in v][0]
print('   ms/batch=%.3f' % d['ms_per_step'], 'rescored=%d' % k['docs_rescored_exactly_per_batch'], {k: round(v,3) for k,v in d['stages_ms'].items() if k.startswith('S4') or k.startswith('S5')})"
done 2>&1 | tee $OUT/${TAG}_cfg4.txt
