#!/bin/bash
# round 6: the fused lazy selection (k_sel_lz_fused): the lazy worker + the selection / graph / level-0 tests, then bench A B A B against a variant on one box   usage: bash tools/r6_sel.sh TAG variant.so
TAG=${1:-r6sel}; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R; mkdir -p $OUT
LAZY_EXPECT=1 timeout 600 python tests/lazy_worker.py 2>&1 | tail -2
timeout 900 python -m pytest tests/test_zz_graph_replay.py tests/test_hip_parity.py -m gpu -x -q -k "graph or lazy or selection or ties or n_full or golden or synthetic_vs_oracle or level0 or full_size_cfg2" 2>&1 | tail -3
for lib in "" "$@" "" "$@"; do
  echo "== ${lib:-tree}"
  FP_LIB_PATH=${lib:+$R/$lib} timeout 300 python bench.py --cpu-queries 16 --steps 40 --warmup 8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('  ', round(d['value']), round(d['ms_per_step'],4), 'p50', round(d['p50_ms'],4), d['repeat_ms_per_step'], d.get('parity_vs_cpu'), {k:round(v,4) for k,v in d['stages_ms'].items() if 'S5' in k})"
done 2>&1 | tee $OUT/${TAG}_ab.txt
