#!/bin/bash
TAG=${1:-r02_y}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 1 --force-dist --dist-mode replicate --no-alt-mode --steps 10 --warmup 3 --cpu-queries 0 > $OUT/${TAG}_rep.json 2> $OUT/${TAG}_rep.err
echo "rc=$?"; grep -E "rank0\]:|Error" $OUT/${TAG}_rep.err | head -20 | cut -c1-300; cat $OUT/${TAG}_rep.json | cut -c1-300
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29546 bench.py --gpus 1 --force-dist --steps 10 --warmup 3 --cpu-queries 0 > $OUT/${TAG}_auto.json 2> $OUT/${TAG}_auto.err
echo "rc=$?"; grep -E "rank0\]:|Error" $OUT/${TAG}_auto.err | head -20 | cut -c1-300; cat $OUT/${TAG}_auto.json | cut -c1-600
