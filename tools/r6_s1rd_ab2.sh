cd $GRAFT_REPO_ROOT
for rd in 0 1 0 1 0 1 0 1 0 1; do
  FP_TEST=s1_rd=$rd timeout 400 python bench.py --cpu-queries 0 --steps 60 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('rd=$rd', round(d['ms_per_step'],4), 'p50', round(d['p50_ms'],4), d['repeat_ms_per_step'])"
done
