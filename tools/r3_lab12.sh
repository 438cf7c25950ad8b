#!/bin/bash
# round 3, lab 12: argmax-step tracking in k_maxsim6 + the stepped repair: tests, A/B against the whole-document repair
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
for v in 1 0; do FP_REPAIR_STEP=$v timeout 300 python bench.py --steps 20 --warmup 5 --cpu-queries 64 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('step=$v', round(d['value'],1), round(d['ms_per_step'],3), 'maxsim', d['stages_ms']['S6+S7 maxsim'], 'repair', d['stages_ms']['S7 order repair'], d.get('parity_vs_cpu'))"; done
timeout 1500 python -m pytest tests -m gpu -x -q -k "repair or golden or certification or shard or native or full_size or synthetic" 2>&1 | tail -4
