#!/bin/bash
# round 6: S1's epilogue, the one-fma (round-toward--inf) form of the level-0 excess against the floor / clamp / subtract form
# (FP_TEST=s1_rd=0): the level-0 / lazy / golden parity tests, then bench A B A B on one box, cfg2 and cfg4   usage: bash tools/r6_s1rd.sh TAG
TAG=${1:-r6s1rd}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "level0 or lazy or golden or synthetic_vs_oracle or non_finite or full_size_cfg2 or fuzz_vs_oracle or centroid_scores" 2>&1 | tail -3
for cfg in cfg2 cfg4; do
for rd in 1 0 1 0; do
  echo "== $cfg s1_rd=$rd"
  FP_TEST=s1_rd=$rd timeout 400 python bench.py --config $cfg --cpu-queries 16 --steps 40 --warmup 8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('  ', round(d['value']), round(d['ms_per_step'],4), 'p50', round(d['p50_ms'],4), d['repeat_ms_per_step'], d.get('parity_vs_cpu'), {k:round(v,4) for k,v in d['stages_ms'].items() if 'S1' in k or 'S4' in k})"
done; done 2>&1 | tee $OUT/${TAG}_ab.txt
