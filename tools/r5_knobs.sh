#!/bin/bash
# bench line + stage times under a list of FP_TEST settings:  bash tools/r5_knobs.sh "sel_gx=4" "sel_gx=16" ...
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R; mkdir -p $OUT
for v in "" "$@"; do
  FP_TEST="$v" timeout 300 python bench.py --cpu-queries 0 --steps 30 --warmup 5 > $OUT/knob.json 2> $OUT/knob.err
  python - "$v" <<PY
import json, sys
d = json.load(open("$OUT/knob.json"))
st = d["stages_ms"]
print("%-22s ms/step %.4f p50 %.4f | %s" % (sys.argv[1] or "(default)", d["ms_per_step"], d["p50_ms"], " ".join("%s %.3f" % (k.split()[0], v) for k, v in st.items())))
PY
done
