#!/bin/bash
# bench.py's multi-GPU modes after the switch to weak-scaling replicas: one rank over RCCL, two ranks over gloo sharing the GPU
R=$GRAFT_REPO_ROOT
cd $R
run() { echo "== $*"; timeout 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $2 bench.py --gpus $1 ${@:3} 2>/tmp/err.txt | python -c "
import sys, json
t = sys.stdin.read().strip().splitlines()
if not t: print('NO OUTPUT'); sys.exit(0)
d = json.loads(t[-1])
print({k: d.get(k) for k in ('value', 'ms_per_step', 'n_gpus', 'scaling')}, d['config']['parallelism'][:90], d['config'].get('global_batch'), 'roofline' in d, 'stages_ms' in d, d.get('alt_mode'))
" || tail -5 /tmp/err.txt; }
run 1 29551 --force-dist --steps 20 --warmup 5
run 2 29552 --dist-backend gloo --steps 6 --warmup 2
run 2 29553 --dist-backend gloo --dist-mode split --steps 6 --warmup 2
run 2 29554 --dist-backend gloo --dist-mode shard --steps 6 --warmup 2
