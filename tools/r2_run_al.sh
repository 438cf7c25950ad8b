#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests -m gpu -q -x -k "golden or level0" 2>&1 | tail -2
for x in 1 2; do
timeout 300 python bench.py --cpu-queries 0 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('run $x', 'qps %.0f ms %.3f' % (d['value'], d['ms_per_step']), 'scan %.3f' % d['stages_ms']['S4 approx'])"
done
