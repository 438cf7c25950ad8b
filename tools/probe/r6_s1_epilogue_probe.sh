#!/bin/bash
# timing-only probes of S1's epilogue (results wrong by construction): s1_rd bit 1 = no floor loads, bit 2 = no table store,
# bit 3 = table store without the byte encoding.  Level 0 forced so that the broken table does not change the engine's choice.
# (The bits were temporary edits of s1_writeout / the engine's s1_rd key and are not in the tree: profiles/r06_s1_rd.txt says what they did.)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for rd in 1 3 5 9 7; do
  rm -rf /tmp/pp$rd
  FP_APPROX_IMPL=l0 FP_TEST=s1_rd=$rd timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp$rd -o run -- python $R/bench.py --steps 12 --warmup 3 --cpu-queries 0 > /tmp/pp$rd.log 2>&1
  f=$(find /tmp/pp$rd -name run_kernel_stats.csv | head -1)
  if [ -z "$f" ]; then tail -5 /tmp/pp$rd.log; continue; fi
  python $R/tools/summarize_prof.py $f /tmp/pp$rd.csv "probe" > /dev/null
  echo "s1_rd=$rd $(grep -h k_centroid_scores_stream /tmp/pp$rd.csv)"
done
