#!/bin/bash
# Stage breakdown of the other BASELINE configs (informational; the bench line is cfg2).
# (warm-up 4: a shape's plain call, its learnt-capacity call and the capturing call (+ anything the runtime initialises at a process's first graph:
# profiles/r06_first_graph_cost.txt) lie before the timed region; until r06_c the warm-up was 2)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
run() { name=$1; shift; timeout 300 python $R/bench.py --steps 5 --warmup 4 --cpu-queries 0 --workload "$name" "$@" 2>$OUT/cfg_$name.err | python -c "
import sys,json
d=json.loads(sys.stdin.read())
allk=[d['roofline']]+list(d['roofline_by_kernel'].values())
k=[v for v in allk if 'candidate_docs_per_batch' in v][0]
m=[v for v in allk if str(v.get('kernel','')).startswith('k_maxsim')][0]
print('$name', 'qps=%.1f ms/batch=%.2f' % (d['value'], d['ms_per_step']), d['stages_ms'], 'dominant=%s' % d['roofline']['kernel'].split(' ')[0], 'maxsim_frac=%.3f' % m['frac'], 'cand/batch=%.0f rescored=%.0f' % (k['candidate_docs_per_batch'], k['docs_rescored_exactly_per_batch']))"; }
run cfg1 --docs 1000 --doc-len 300 --batch 16 --qlen 50 --topk 10
run cfg4 --docs 100000 --doc-len 1024 --batch 32 --topk 100
run cfg5 --docs 5000000 --centroids 65536 --batch 128
run cfg5_nfull16k --docs 5000000 --centroids 65536 --batch 128 --nfull 16384
run cfg5_nfull64k --docs 5000000 --centroids 65536 --batch 128 --nfull 65536
run cfg3_1gpu --config cfg3
