#!/bin/bash
# round 3, lab 7: sharded search rework (status words, sort-free cut / union, sub-batches, failure injection) + bench modes
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "shard or native or replicated" > $OUT/r3_lab7_tests.log 2>&1; tail -8 $OUT/r3_lab7_tests.log | cut -c1-600
# one rank through the RCCL path at cfg2 (the code path of N > 1) and the cfg3 one-GPU anchor
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --force-dist --dist-mode shard --config cfg2 --steps 20 --warmup 5 > $OUT/r3_lab7_dist1.json 2> $OUT/r3_lab7_dist1.err; tail -c 600 $OUT/r3_lab7_dist1.json; tail -3 $OUT/r3_lab7_dist1.err
timeout 900 python bench.py --gpus 1 --config cfg3 --steps 10 --warmup 3 > $OUT/r3_lab7_cfg3_1gpu.json 2> $OUT/r3_lab7_cfg3.err; python -c "
import json
d=json.loads(open('$OUT/r3_lab7_cfg3_1gpu.json').read().strip().splitlines()[-1]); print('cfg3 1gpu', round(d['value'],1), round(d['ms_per_step'],2), d['stages_ms'], d['config']['index_bytes_per_gpu'])" || tail -5 $OUT/r3_lab7_cfg3.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 1 --force-dist --config cfg3 --steps 10 --warmup 3 > $OUT/r3_lab7_cfg3_dist1.json 2> $OUT/r3_lab7_cfg3_dist1.err; tail -c 700 $OUT/r3_lab7_cfg3_dist1.json; tail -3 $OUT/r3_lab7_cfg3_dist1.err
