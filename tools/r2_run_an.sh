#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "golden or level0 or random or synthetic or bound_and" 2>&1 | tail -3
for w in 8 4 8 4; do
FP_S1_WAVES=$w timeout 300 python bench.py --cpu-queries 0 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('waves=$w', 'qps %.0f ms %.3f' % (d['value'], d['ms_per_step']), 'S1 %.3f' % d['stages_ms']['S1 centroid_gemm'])"
done
