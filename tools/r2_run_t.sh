#!/bin/bash
TAG=${1:-r02_t}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -k "level0 or bound_and_refine or golden or full_size or random" 2>&1 | tail -30 > $OUT/${TAG}_tests.log
grep -E "^E   |passed|failed|^FAILED" $OUT/${TAG}_tests.log | cut -c1-300 | head -30
timeout 400 python bench.py --cpu-queries 0 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
for tail in 0.01 0.05 0.1; do
  FP_L0_TAIL=$tail timeout 300 python bench.py --cpu-queries 0 --steps 10 --warmup 3 > $OUT/${TAG}_bench_tail$tail.json 2> /dev/null
done
python - <<PY
import json, glob, os
for p in sorted(glob.glob("$OUT/${TAG}_bench*.json")):
    try:
        d = json.load(open(p))
    except Exception as e:
        print(os.path.basename(p), "unreadable", e); continue
    st = d.get("stages_ms", {})
    print(os.path.basename(p), "qps %.0f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in st.items()}, d["roofline"].get("docs_rescored_exactly_per_batch"))
PY
bash tools/pmc_scan.sh k_l0_scan 2>&1 | tail -12 | cut -c1-400
