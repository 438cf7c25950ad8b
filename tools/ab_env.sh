#!/bin/bash
# A/B of environment switches on the bench workload: bash tools/ab_env.sh "FP_X=0" "FP_X=1" ...  (each argument = one run's env, space-separated)
cd $GRAFT_REPO_ROOT
for e in "$@"; do
  echo "== $e"
  env $e python bench.py --cpu-queries 0 --steps 20 --warmup 5 ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'p50', round(d.get('p50_ms',0),4), {k:round(v,4) for k,v in d['stages_ms'].items()})"
done
