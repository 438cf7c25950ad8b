#!/bin/bash
# round 5, launch-tail pass: correctness of the touched kernels (lazy worker, graph replay, probe / golden / random shapes, level 0,
# subset + speculative paths), then the bench line and the rocprofv3 kernel summary.   usage: bash tools/r5_micro.sh TAG
TAG=${1:-r5mic}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R; mkdir -p $OUT
LAZY_EXPECT=1 timeout 600 python tests/lazy_worker.py 2>&1 | tail -2
timeout 900 python -m pytest tests/test_zz_graph_replay.py tests/test_hip_parity.py -m gpu -x -q -k "graph or lazy or probe or golden or synthetic_vs_oracle or randomized or level0 or subset or spec or n_full or ties" 2>&1 | tail -3
timeout 300 python bench.py --cpu-queries 0 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python - <<PY
import json
d = json.load(open("$OUT/${TAG}_bench.json"))
print("ms/step", d["ms_per_step"], "p50", d["p50_ms"], "dev", d["value_device_io"]["ms_per_step"], d.get("parity_vs_cpu"))
print(d["stages_ms"])
PY
cd /tmp && export TMPDIR=/tmp
FP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o run -- python $R/bench.py --steps 8 --warmup 3 --cpu-queries 0 > $OUT/${TAG}_prof.log 2>&1
python $R/tools/summarize_prof.py $(ls $OUT/${TAG}_prof/*/run_kernel_stats.csv $OUT/${TAG}_prof/run_kernel_stats.csv 2>/dev/null | head -1) $OUT/${TAG}_kernel_stats.csv "bench.py --steps 8 --warmup 3, FP_GRAPH=0"
grep -E "k_lz_exact|k_sel_|k_ivf_mark|k_final|k_cand|k_probe|k_l0_thr" $OUT/${TAG}_kernel_stats.csv
