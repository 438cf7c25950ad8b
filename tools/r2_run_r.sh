#!/bin/bash
TAG=${1:-r02_r}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
bash tools/bench_configs.sh 2>&1 | tee $OUT/${TAG}_configs.txt | cut -c1-700
for f in $OUT/cfg_*.err; do echo "== $f"; tail -3 $f | cut -c1-300; done
