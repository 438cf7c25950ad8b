#!/bin/bash
# graph replay on by default (tag r02_i): GPU suite, smoke, default bench line
R=$GRAFT_REPO_ROOT
cd $R && timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r02_i_gpu_tests.log 2>&1; tail -12 gpurun_out/r02_i_gpu_tests.log | cut -c1-600
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-300
timeout 500 python bench.py > gpurun_out/r02_i_bench.json 2> gpurun_out/r02_i_bench.err; python -c "
import json
d = json.loads(open('gpurun_out/r02_i_bench.json').read().strip().splitlines()[-1])
print(round(d['value'], 1), round(d['ms_per_step'], 4), d['p50_ms'], d['scaling'], d.get('parity_vs_cpu'), d['stages_ms'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['value_device_io'])" || tail -5 gpurun_out/r02_i_bench.err
