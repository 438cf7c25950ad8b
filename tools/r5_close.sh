#!/bin/bash
# evidence for a last small change: the lazy worker + the graph / selection tests on the shipped build, then profile_round.sh's
# passes WITHOUT the full suite (bench line, rocprofv3 kernel summary, PMC traffic keyed by the library hash)
T=${1:-r05_c}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
( LAZY_EXPECT=1 timeout 300 python tests/lazy_worker.py 2>&1 | tail -2
  timeout 300 python -m pytest tests/test_zz_graph_replay.py tests/test_hip_parity.py -m gpu -x -q -k "graph or golden_batched or ties" 2>&1 | tail -2 ) > $OUT/${T}_gpu_tests.log 2>&1
cat $OUT/${T}_gpu_tests.log
sed -e 's/^timeout 1500 python -m pytest tests -m gpu.*$/true/' tools/profile_round.sh > /tmp/profile_round_nosuite.sh
bash /tmp/profile_round_nosuite.sh $T > $OUT/${T}_profile_round.log 2>&1
tail -2 $OUT/${T}_profile_round.log | cut -c1-400
