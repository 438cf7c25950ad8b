#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== dim 64"; FP_S1_STATS=1 FP_GRAPH=0 timeout 300 python tools/s1_stats_cfg2.py 6 1000000 64 2>&1 | tail -7
