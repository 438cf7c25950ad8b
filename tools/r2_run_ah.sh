#!/bin/bash
TAG=${1:-r02_ah}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | cut -c1-300
for mode in replica shard; do
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29547 bench.py --gpus 2 --dist-backend gloo --dist-mode $mode --steps 5 --warmup 2 --cpu-queries 0 > $OUT/${TAG}_gloo2_$mode.json 2> $OUT/${TAG}_gloo2_$mode.err
echo "rc=$?"; grep -E "rank[01]\]:|Error" $OUT/${TAG}_gloo2_$mode.err | head -8 | cut -c1-300; cat $OUT/${TAG}_gloo2_$mode.json | cut -c1-500
done
timeout 900 python -m pytest tests -m gpu -q -x -k "roundtrip or one_shot" 2>&1 | tail -3
