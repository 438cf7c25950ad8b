#!/bin/bash
# round 3, lab 8: level-0 scan with 4 lanes per candidate for multi-range tables: tests + cfg3 anchor + cfg2 check
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "level0 or q8 or bound or synthetic or graph or spec" > $OUT/r3_lab8_tests.log 2>&1; tail -6 $OUT/r3_lab8_tests.log | cut -c1-600
for ppl in 0 8; do
FP_L0_PPL=$ppl timeout 900 python bench.py --gpus 1 --config cfg3 --steps 10 --warmup 3 > $OUT/r3_lab8_cfg3_ppl$ppl.json 2> $OUT/r3_lab8_cfg3.err; python -c "
import json
d=json.loads(open('$OUT/r3_lab8_cfg3_ppl$ppl.json').read().strip().splitlines()[-1]); print('cfg3 1gpu ppl=$ppl', round(d['value'],1), round(d['ms_per_step'],2), d['stages_ms'], d['config']['index_bytes_per_gpu'])" || tail -5 $OUT/r3_lab8_cfg3.err
done
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-queries 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('cfg2', round(d['value'],1), round(d['ms_per_step'],3), d['stages_ms'])"
