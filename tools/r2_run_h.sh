#!/bin/bash
TAG=${1:-r02_h}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -k "one_shot or reference_restatement or maxsim_repair or sharded" 2>&1 | tail -150 > $OUT/${TAG}_focus_tests.log
grep -E "^E  |passed|failed|^FAILED" $OUT/${TAG}_focus_tests.log | cut -c1-300 | head -60
FP_MAXSIM_REPAIR=0 DBG_TAG=r0 timeout 400 python tools/debug_parity2.py 2>&1 | tail -1
FP_MAXSIM_REPAIR=1 DBG_TAG=r1 timeout 300 python tools/debug_parity2.py 2>&1 | tail -1
for c in 1 3; do
  FP_REPAIR_CHAIN=$c FP_MAXSIM_REPAIR=2 DBG_TAG=r2 timeout 300 python tools/debug_parity2.py 2>&1 | tail -1
  echo "=== chain $c"
  DBG_TAG=compare python tools/debug_parity2.py 2>&1 | tee $OUT/${TAG}_debug2_chain$c.txt | grep -E "repair|doc " | head -24 | cut -c1-300
done
timeout 600 python -m pytest tests/test_hip_parity.py::test_full_size_cfg2_id_lists_vs_oracle -q -x 2>&1 | tail -15 | cut -c1-300
timeout 400 python bench.py --cpu-queries 0 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --force-dist --dist-mode shard --no-alt-mode --steps 10 --warmup 3 --cpu-queries 0 > $OUT/${TAG}_bench_dist1_native.json 2> $OUT/${TAG}_dist_native.err
grep -E "rank0|Error|error" $OUT/${TAG}_dist_native.err | head -20 | cut -c1-300
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 1 --force-dist --dist-mode shard --dist-impl torch --no-alt-mode --steps 10 --warmup 3 --cpu-queries 0 > $OUT/${TAG}_bench_dist1_torch.json 2> $OUT/${TAG}_dist_torch.err
grep -E "rank0|Error|error" $OUT/${TAG}_dist_torch.err | head -20 | cut -c1-300
python - <<PY
import json, glob, os
for p in sorted(glob.glob("$OUT/${TAG}_bench*.json")):
    try:
        d = json.load(open(p))
    except Exception as e:
        print(os.path.basename(p), "unreadable", e); continue
    st = d.get("stages_ms", {})
    print(os.path.basename(p), "qps %.0f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in st.items()},
          "repaired", d.get("docs_repaired_per_batch"), d.get("config", {}).get("parallelism"), "ident", d.get("parity", d.get("identical_id_lists")))
PY
