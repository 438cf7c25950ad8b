#!/bin/bash
# round 6: the build-from-vectors corpus (tools/bench_gmm.py) under the forms of S4, one line each -> gpurun_out/${TAG}_gmm.jsonl
# usage: bash tools/r6_gmm.sh TAG [docs] ["ENV=.. ENV=.." ...]     (each further argument = one run's environment)
TAG=${1:-r6gmm}; DOCS=${2:-250000}; shift; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R; mkdir -p $OUT
: > $OUT/${TAG}_gmm.jsonl
if [ $# -eq 0 ]; then set -- "X=0" "FP_APPROX_IMPL=l0" "FP_APPROX_IMPL=l0h" "FP_APPROX_IMPL=q8"; fi
for e in "$@"; do
  env $e timeout 600 python tools/bench_gmm.py --docs $DOCS --tag "$e" 2>/dev/null | tail -1 >> $OUT/${TAG}_gmm.jsonl
done
python - <<PY
import json
for l in open("$OUT/${TAG}_gmm.jsonl"):
    l = l.strip()
    if not l: continue
    d = json.loads(l)
    print(d["tag"], "ms", d["ms_per_batch"], "cand/q", round(d["candidates_per_query"]), "exact/q", round(d["rescored_exactly_per_query"]), "codes/doc", round(d["unique_codes_per_doc"], 1))
    print("    ", {k: round(v, 3) for k, v in d["stages_ms"].items()})
PY
