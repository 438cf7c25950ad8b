#!/bin/bash
# round-2 closing check (tag r02_g): GPU suite, default bench line, one-rank lines of the two multi-GPU modes
R=$GRAFT_REPO_ROOT
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r02_g_gpu_tests.log 2>&1; tail -5 gpurun_out/r02_g_gpu_tests.log
timeout 500 python bench.py > gpurun_out/r02_g_bench.json 2> gpurun_out/r02_g_bench.err; tail -c 400 gpurun_out/r02_g_bench.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --force-dist --dist-mode shard --steps 20 --warmup 5 > gpurun_out/r02_g_bench_dist1_native.json 2> /dev/null
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 1 --force-dist --steps 20 --warmup 5 > gpurun_out/r02_g_bench_dist1_auto.json 2> /dev/null
for f in gpurun_out/r02_g_bench.json gpurun_out/r02_g_bench_dist1_native.json gpurun_out/r02_g_bench_dist1_auto.json; do python -c "
import json
d = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['value'], 1), round(d['ms_per_step'], 4), d['scaling'], d.get('parity_vs_cpu'))"; done
