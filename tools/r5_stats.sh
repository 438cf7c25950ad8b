#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
FP_S1_STATS=1 FP_GRAPH=0 timeout 300 python tools/s1_stats_cfg2.py 4 2>&1 | tail -5
