#!/bin/bash
# opt-in graph replay (FP_GRAPH=1): GPU suite (the graph test runs last), latency at B = 64 / 8 / 1 with and without it
R=$GRAFT_REPO_ROOT
cd $R && timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r02_h_gpu_tests.log 2>&1; tail -25 gpurun_out/r02_h_gpu_tests.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
for g in 1 0; do
  for b in 1 8 64; do
    FP_GRAPH=$g timeout 120 python $R/bench.py --batch $b --steps 60 --warmup 8 --cpu-queries 0 --workload cfg2_b$b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  graph=$g B=$b ms', round(d['ms_per_step'],4), 'p50', round(d['p50_ms'],4), 'value', round(d['value'],1), 'dev_io', d.get('value_device_io',{}).get('ms_per_step'))"
  done
done
