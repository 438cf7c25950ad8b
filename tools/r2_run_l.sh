#!/bin/bash
TAG=${1:-r02_l}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -k "level0 or bound_and_refine or full_size or golden or random" 2>&1 | tail -30 > $OUT/${TAG}_tests.log
grep -E "^E   |passed|failed|^FAILED" $OUT/${TAG}_tests.log | cut -c1-300 | head -30
timeout 400 python bench.py --cpu-queries 0 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python - <<PY
import json
d = json.load(open("$OUT/${TAG}_bench.json"))
print("qps %.0f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in d["stages_ms"].items()}, d["roofline"].get("docs_rescored_exactly_per_batch"))
PY
bash tools/r2_run_k.sh ${TAG} | tail -75
