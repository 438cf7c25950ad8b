#!/bin/bash
# round 3, lab 6: in-place native residual order, more k_maxsim6 shapes: GPU tests + bench
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/r3_lab6_tests.log 2>&1; tail -15 $OUT/r3_lab6_tests.log | cut -c1-400
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-queries 64 2>$OUT/r3_lab6_bench.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('maxsim_ms=%.4f repair_ms=%.4f step_ms=%.3f qps=%.0f' % (d['stages_ms']['S6+S7 maxsim'], d['stages_ms']['S7 order repair'], d['ms_per_step'], d['value']), d.get('parity_vs_cpu'), d['config']['index_bytes_per_gpu'], d['stages_ms'])" || tail -5 $OUT/r3_lab6_bench.err
