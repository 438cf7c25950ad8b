#!/bin/bash
TAG=${1:-r02_aj}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -k "golden or level0 or full_size" 2>&1 | tail -3
for x in 1 0 1 0; do
FP_L0_XCD=$x timeout 300 python bench.py --cpu-queries 0 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('xcd=$x', 'qps %.0f ms %.3f' % (d['value'], d['ms_per_step']), 'scan %.3f' % d['stages_ms']['S4 approx'])"
done
