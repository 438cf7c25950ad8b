#!/bin/bash
# S3 of cfg5 / cfg3 / cfg2 for the shipped library and variant libraries:  bash tools/r5_s3ab.sh [variant.so ...]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R; mkdir -p $OUT
run() { lib="$1"; knob="$2"; shift 2
  if [ -n "$lib" ]; then export FP_LIB_PATH=$R/$lib; else unset FP_LIB_PATH; fi
  FP_TEST="$knob" timeout 400 python bench.py --cpu-queries 0 --steps 6 --warmup 3 "$@" > $OUT/knob.json 2> $OUT/knob.err
  python - "$lib" "$knob" "$*" <<PY
import json, sys
d = json.load(open("$OUT/knob.json"))
st = d["stages_ms"]
print("%-30s %-12s %-40s ms/step %.3f | S3 mark %.4f compact %.4f S2 %.4f S5 %.4f" % (sys.argv[1] or "(shipped)", sys.argv[2], sys.argv[3], d["ms_per_step"], st["S3 ivf_mark+count"], st["S3 compact"], st["S2 probe_topk"], st["S5 select"]))
PY
}
C5="--docs 5000000 --centroids 65536 --batch 128"
for lib in "" "$@"; do
  run "$lib" "" $C5
  run "$lib" ""
done
