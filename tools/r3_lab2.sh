#!/bin/bash
# round 3, lab 2: k_maxsim6 first run: GPU tests, bench A/B against k_maxsim5
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
for impl in 6 5; do
  FP_MAXSIM_IMPL=$impl timeout 300 python bench.py --steps 10 --warmup 3 --cpu-queries 0 2>$OUT/r3_lab2_bench$impl.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('impl $impl', 'maxsim_ms=%.4f repair_ms=%.4f step_ms=%.3f qps=%.0f' % (d['stages_ms']['S6+S7 maxsim'], d['stages_ms']['S7 order repair'], d['ms_per_step'], d['value']))" || tail -5 $OUT/r3_lab2_bench$impl.err
done
FP_MS_XCD=1 timeout 300 python bench.py --steps 10 --warmup 3 --cpu-queries 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('impl 6 xcd', 'maxsim_ms=%.4f repair_ms=%.4f step_ms=%.3f qps=%.0f' % (d['stages_ms']['S6+S7 maxsim'], d['stages_ms']['S7 order repair'], d['ms_per_step'], d['value']))"
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/r3_lab2_tests.log 2>&1; tail -15 $OUT/r3_lab2_tests.log | cut -c1-300
