#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
LAZY_EXPECT=1 timeout 600 python tests/lazy_worker.py 2>&1 | tail -2
FP_S1_EXACT=1 LAZY_EXPECT=0 timeout 600 python tests/lazy_worker.py 2>&1 | tail -2
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "probe or golden_stagewise or synthetic_vs_oracle or randomized" 2>&1 | tail -3
bash tools/launch_sequence.sh > gpurun_out/r05_launch_sequence.txt 2>&1; grep -E "^B=|k_" gpurun_out/r05_launch_sequence.txt | head -80
