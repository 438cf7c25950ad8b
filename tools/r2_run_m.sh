#!/bin/bash
TAG=${1:-r02_m}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
for abl in 1 2 3; do
  FP_L0_ABL=$abl timeout 300 python bench.py --cpu-queries 0 --steps 10 --warmup 3 > $OUT/${TAG}_bench_abl$abl.json 2> /dev/null
done
for pm in 6 8; do
  FP_L0_PILOT=$pm timeout 300 python bench.py --cpu-queries 0 --steps 10 --warmup 3 > $OUT/${TAG}_bench_pilot$pm.json 2> /dev/null
done
python - <<PY
import json, glob, os
for p in sorted(glob.glob("$OUT/${TAG}_bench*.json")):
    try:
        d = json.load(open(p))
    except Exception as e:
        print(os.path.basename(p), "unreadable", e); continue
    st = d.get("stages_ms", {})
    print(os.path.basename(p), "qps %.0f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in st.items() if k.startswith("S4") or k.startswith("S5")}, d["roofline"].get("docs_rescored_exactly_per_batch"))
PY
