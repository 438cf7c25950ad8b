#!/bin/bash
# How much margin does S1's certification window have?  FP_S1_EXACT=2 re-evaluates EVERY centroid score with the reference's
# ascending chain and counts the entries the window did NOT flag whose chain value differs from the staged MFMA value
# (unflagged_differences: must be 0 at the shipped window).  The sweep shrinks both terms of the window by 2^-s, s = 0..6, on
# BASELINE cfg2 (1 M documents, 131072 centroids, 64 x 32 query columns per batch): where the first differences appear -- and how
# fast their count grows -- is the measured tail of |MFMA accumulator - chain accumulator|.
# usage (GPU box): bash tools/s1_window_sweep.sh [batches] [docs] [dim] [scales] > gpurun_out/s1_window_sweep.txt
NB=${1:-10}; DOCS=${2:-1000000}; DIM=${3:-128}; SCALES=${4:-"0 1 2 3 4 5 6"}
cd ${GRAFT_REPO_ROOT:-.}
for s in $SCALES; do
  w0=$(python -c "print(-21.5 - $s)"); ka=$(python -c "print(-20.0 - $s)")
  echo "## window scale 2^-$s: w0 = 2^$w0, kappa = 2^$ka"
  FP_S1_EXACT=2 FP_S1_STATS=1 FP_TEST="s1_w0_log2=$w0,s1_kappa_log2=$ka" timeout 200 python tools/s1_stats_cfg2.py $NB $DOCS $DIM 2>&1 | tail -1
done
