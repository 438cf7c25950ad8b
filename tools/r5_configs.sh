#!/bin/bash
# stage breakdown of the other BASELINE configs and of cfg2 variants (q_len, dim) -> gpurun_out/r05_configs.txt
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
T=${1:-r05_a}
bash tools/bench_configs.sh 2>&1 | tee $OUT/${T}_configs.txt
run2() { name="$1"; shift; timeout 300 python bench.py --steps 5 --warmup 4 --cpu-queries 0 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg2 $name: ms/batch=%.3f' % d['ms_per_step'], {k: round(v,3) for k,v in d['stages_ms'].items()})" | tee -a $OUT/${T}_configs.txt; }
run2 "--qlen 64" --qlen 64
run2 "--qlen 70" --qlen 70
run2 "--qlen 128" --qlen 128
run2 "--dim 96" --dim 96
run2 "--dim 64" --dim 64
run2 "--dim 256 --docs 300000" --dim 256 --docs 300000
