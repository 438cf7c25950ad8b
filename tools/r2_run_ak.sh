#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for x in 2 0 2 0; do
FP_L0_XCD=$x timeout 300 python bench.py --cpu-queries 0 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('order=$x', 'qps %.0f ms %.3f' % (d['value'], d['ms_per_step']), 'scan %.3f' % d['stages_ms']['S4 approx'])"
done
FP_L0_XCD=2 timeout 600 python -m pytest tests -m gpu -q -x -k "golden or level0" 2>&1 | tail -2
