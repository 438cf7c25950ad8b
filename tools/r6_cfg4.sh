#!/bin/bash
# round 6: cfg4 (ColPali shape: 100k docs x 1024 tokens, batch 32, top_k 100 -- the hot-code form of S4's level 0) for the tree's library and
# variants on the SAME box, after the tests that drive that form.   usage: bash tools/r6_cfg4.sh TAG [variant.so ...]
TAG=${1:-r6cfg4}; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "level0 or build_from_vectors or bound_and_refine or full_size_properties" 2>&1 | tail -4
run() { timeout 300 python $R/bench.py --steps 10 --warmup 3 --cpu-queries 0 --workload cfg4 --docs 100000 --doc-len 1024 --batch 32 --topk 100 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('   ms/batch=%.3f' % d['ms_per_step'], {k: round(v,3) for k,v in d['stages_ms'].items()})"; }
for lib in "" "$@" "" "$@"; do
  echo "== ${lib:-tree}"
  FP_LIB_PATH=${lib:+$R/$lib} run
done 2>&1 | tee $OUT/${TAG}_cfg4.txt
