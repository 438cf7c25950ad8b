#!/bin/bash
TAG=${1:-r02_ai}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -k "golden or random or level0 or synthetic or full_size or token" 2>&1 | tail -5
timeout 300 python bench.py --cpu-queries 0 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
FP_S1_QREG=0 timeout 300 python bench.py --cpu-queries 0 > $OUT/${TAG}_bench_old.json 2> /dev/null
python - <<PY
import json, glob, os
for p in sorted(glob.glob("$OUT/${TAG}_bench*.json")):
    d = json.load(open(p))
    print(os.path.basename(p), "qps %.0f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in d["stages_ms"].items()})
PY
