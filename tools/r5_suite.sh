#!/bin/bash
# full GPU suite + default bench line
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
T=${1:-r5s}
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/${T}_gpu_tests.log
cat $OUT/${T}_gpu_tests.log
