#!/bin/bash
# round 3, lab 9: full GPU suite (shape LRU, certification at 2.1 M columns, new shard paths) + default bench
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/r3_lab9_tests.log 2>&1; tail -8 $OUT/r3_lab9_tests.log | cut -c1-800
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-queries 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('cfg2', round(d['value'],1), round(d['ms_per_step'],3), d['p50_ms'], d['stages_ms'])"
