#!/bin/bash
TAG=${1:-r02_aa}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $OUT/${TAG}_tests.log
grep -E "^E   |passed|failed|^FAILED" $OUT/${TAG}_tests.log | cut -c1-300 | head -30
timeout 400 python bench.py --cpu-queries 0 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python - <<PY
import json
d = json.load(open("$OUT/${TAG}_bench.json"))
print("qps %.0f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in d["stages_ms"].items()})
PY
bash tools/r2_run_z.sh ${TAG} 2>&1 | tail -64
