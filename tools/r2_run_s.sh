#!/bin/bash
TAG=${1:-r02_s}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -k "level0 or bound_and_refine or golden" 2>&1 | tail -30 > $OUT/${TAG}_tests.log
grep -E "^E   |passed|failed|^FAILED" $OUT/${TAG}_tests.log | cut -c1-300 | head -30
run() { name=$1; shift; timeout 300 python $R/bench.py --steps 5 --warmup 2 --cpu-queries 0 --workload "$name" "$@" 2>$OUT/cfg_$name.err | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$name', 'qps=%.1f ms/batch=%.2f' % (d['value'], d['ms_per_step']), d['stages_ms'])"; }
run cfg4 --docs 100000 --doc-len 1024 --batch 32 --topk 100
run cfg5_nfull64k --docs 5000000 --centroids 65536 --batch 128 --nfull 65536
