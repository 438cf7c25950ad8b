#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-queries 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('cfg2', round(d['value'],1), round(d['ms_per_step'],3), d['p50_ms'], d['stages_ms'])"
timeout 600 python -m pytest tests -m gpu -x -q -k "graph or repair or golden" 2>&1 | tail -3
