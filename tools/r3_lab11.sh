#!/bin/bash
# round 3, lab 11: build-from-vectors corpus (Gaussian mixture -> k-means -> codec -> fp_compress): level 0 vs the 8-bit stage; n_full sweep test
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests -m gpu -x -q -k "n_full_scores_sweep" 2>&1 | tail -3
rm -f /tmp/gmm_corpus.npz
timeout 900 python tools/bench_gmm.py --parity 50000 --tag auto 2>$OUT/r3_gmm_auto.err | tee $OUT/r3_gmm.jsonl | cut -c1-900; tail -3 $OUT/r3_gmm_auto.err
FP_APPROX_IMPL=l0 timeout 600 python tools/bench_gmm.py 2>/dev/null | tee -a $OUT/r3_gmm.jsonl | cut -c1-900
FP_APPROX_IMPL=q8 timeout 600 python tools/bench_gmm.py 2>/dev/null | tee -a $OUT/r3_gmm.jsonl | cut -c1-900
