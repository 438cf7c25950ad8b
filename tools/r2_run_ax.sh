#!/bin/bash
# candidate-capacity speculation: GPU suite (with the new worker test), bench at B = 64 / 8 / 1 with and without it
R=$GRAFT_REPO_ROOT
cd $R && timeout 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
for sp in 1 0; do
  for b in 64 8 1; do
    FP_SPECULATE=$sp timeout 200 python $R/bench.py --batch $b --steps 40 --warmup 5 --cpu-queries 0 --workload cfg2_b$b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  spec=$sp B=$b ms', round(d['ms_per_step'],4), 'value', round(d['value'],1), 'dev_io', d.get('value_device_io',{}).get('ms_per_step'), d.get('stages_ms'))"
  done
done
