#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
T=${1:-r5m}
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "golden_stagewise or maxsim_column or golden_batched or synthetic_vs_oracle or maxsim_repair" 2>&1 | tail -12
timeout 400 python bench.py > $OUT/${T}_bench.json 2> $OUT/${T}_bench.err; echo "bench rc=$?"
python - $T <<'PY'
import json, sys
T = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/{T}_bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "dev", d["value_device_io"]["ms_per_step"], "parity", d.get("parity_vs_cpu"))
    print("stages", d.get("stages_ms"))
    for k, v in d["roofline_by_kernel"].items(): print(k, v.get("frac"), v.get("avg_launch_ms"), v.get("frac_with_repair"))
except Exception as e:
    print("bench parse failed", e); print(open(f"gpurun_out/{T}_bench.err").read()[-3000:])
PY
FP_TEST=maxsim7=0 timeout 400 python bench.py --cpu-queries 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('maxsim6:', d['ms_per_step'], d['stages_ms']['S6+S7 maxsim'])"
