#!/bin/bash
# round 6's closing evidence on ONE box: profile_round.sh (GPU suite, bench line, rocprofv3 kernel summary, PMC traffic keyed by the
# library hash), the other BASELINE configs, one-query / eight-query latency, the kernel sequence of one call, the call shapes of
# tools/r6_usage.py, zero-padded queries, the build-from-vectors corpus.   usage: bash tools/r6_final.sh TAG
T=${1:-r06_a}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
bash tools/profile_round.sh $T > $OUT/${T}_profile_round.log 2>&1
tail -4 $OUT/${T}_profile_round.log | cut -c1-600
cd $R
for B in 1 8; do
  timeout 200 python bench.py --batch $B --steps 60 --warmup 10 --cpu-queries 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=$B: mean %.4f ms  p50 %.4f ms  p90 %.4f ms (fp_search through the replayed graph, 60 steps)' % (d['ms_per_step'], d['p50_ms'], d['p90_ms']))"
done | tee $OUT/${T}_latency.txt
bash tools/launch_sequence.sh > $OUT/${T}_launch_sequence.txt 2>&1
bash tools/r5_configs.sh $T > /dev/null 2>&1
cut -c1-200 $OUT/${T}_configs.txt
timeout 600 python tools/r6_usage.py > $OUT/${T}_usage_shapes.jsonl 2>/dev/null
timeout 300 python bench.py --cpu-queries 16 --zero-rows 8 --workload cfg2_zero_padded > $OUT/${T}_zero_padded.json 2>/dev/null
timeout 900 python tools/bench_gmm.py --docs 250000 --parity 50000 --tag "round 6 (auto)" 2>/dev/null | tail -1 > $OUT/${T}_gmm.jsonl
cat $OUT/${T}_gmm.jsonl | cut -c1-400
