#!/bin/bash
TAG=${1:-r02_ag}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
for sm in 8192 4096 2048 1024; do
  FP_L0_SAMPLE=$sm timeout 300 python bench.py --cpu-queries 0 --steps 20 --warmup 5 > $OUT/${TAG}_bench_s$sm.json 2> /dev/null
done
python - <<PY
import json, glob, os
for p in sorted(glob.glob("$OUT/${TAG}_bench*.json")):
    d = json.load(open(p))
    st = d.get("stages_ms", {})
    print(os.path.basename(p), "qps %.0f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in st.items() if k.startswith("S1") or k.startswith("S4")}, d["roofline"].get("docs_rescored_exactly_per_batch"))
PY
