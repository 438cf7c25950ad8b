#!/bin/bash
# bench line + stage times for variant libraries (FP_LIB_PATH):  bash tools/r5_libs.sh a.so b.so ...
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R; mkdir -p $OUT
for v in "" "$@"; do
  if [ -n "$v" ]; then export FP_LIB_PATH=$R/$v; else unset FP_LIB_PATH; fi
  timeout 300 python bench.py --cpu-queries 0 --steps 30 --warmup 5 > $OUT/knob.json 2> $OUT/knob.err
  python - "$v" <<PY
import json, sys
d = json.load(open("$OUT/knob.json"))
st = d["stages_ms"]
print("%-34s ms/step %.4f p50 %.4f | %s" % (sys.argv[1] or "(default)", d["ms_per_step"], d["p50_ms"], " ".join("%s %.3f" % (k.split()[0], v) for k, v in st.items())))
PY
done
