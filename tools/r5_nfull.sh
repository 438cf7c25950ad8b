#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "n_full_scores_sweep or lazy_centroid or massive_score_ties or golden_batched" 2>&1 | tail -5
for nf in 16384 32768; do
timeout 300 python bench.py --steps 8 --warmup 3 --cpu-queries 0 --docs 5000000 --centroids 65536 --batch 128 --nfull $nf 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg5 nfull $nf: ms/batch=%.3f' % d['ms_per_step'], {k: round(v,3) for k,v in d['stages_ms'].items()}, d['roofline_by_kernel']['k_centroid_scores'].get('s1_form'))"
done
