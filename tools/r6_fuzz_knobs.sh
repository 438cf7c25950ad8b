#!/bin/bash
# the search-surface fuzz under the engine's test knobs: forced sub-batching, forced candidate-capacity overflows, every S1 mode, no graphs
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout 900 env "$@" 2>&1 | grep -v amdgpu.ids | grep "FUZZ_" | tail -4 | cut -c1-400; }
run FP_TEST=s_budget_kb=256 python tests/fuzz_worker.py 600 51
run FP_TEST=s_budget_kb=2048 python tests/fuzz_worker.py 150 52 0 big
run FP_TEST=spec_cap_pct=50 python tests/fuzz_worker.py 600 53
run FP_TEST=spec_cap_pct=50 python tests/fuzz_worker.py 150 54 0 big
run FP_TEST=spec_cap_pct=50 python tests/fuzz_worker.py 300 55 0 stateful
for m in 0 1 2 3; do run FP_S1_EXACT=$m python tests/fuzz_worker.py 400 56; run FP_S1_EXACT=$m python tests/fuzz_worker.py 100 57 0 big; done
run FP_GRAPH=0 python tests/fuzz_worker.py 300 58 0 stateful
run FP_S1_STATS=1 python tests/fuzz_worker.py 300 59
