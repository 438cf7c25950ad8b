#!/bin/bash
# per-kernel averages of the cfg2 batch with S1's one-fma excess form on / off (rocprofv3 kernel summary of each)   usage: bash tools/r6_s1rd_prof.sh TAG
TAG=${1:-r6s1rd}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rd in 1 0; do
  rm -rf /tmp/prd$rd
  FP_TEST=s1_rd=$rd timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prd$rd -o run -- python $R/bench.py --steps 20 --warmup 5 --cpu-queries 0 > /dev/null 2>&1
  python $R/tools/summarize_prof.py $(find /tmp/prd$rd -name run_kernel_stats.csv | head -1) $OUT/${TAG}_rd${rd}_kernel_stats.csv "bench.py --steps 20 --warmup 5 (cfg2), FP_TEST=s1_rd=$rd, MI355X"
  echo "== s1_rd=$rd"; grep -E "k_centroid_scores_stream|k_l0_scan|k_approx|k_maxsim6|k_ivf_mark|k_l0_floor|k_l0_pilot|k_l0_compact" $OUT/${TAG}_rd${rd}_kernel_stats.csv
done
