#!/bin/bash
# rocprofv3 kernel summary of cfg4 (100k docs x 1024 tokens, batch 32, top_k 100) -> gpurun_out/${TAG}_cfg4_kernel_stats.csv
TAG=${1:-r06_c}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p4
FP_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p4 -o run -- python $R/bench.py --steps 10 --warmup 3 --cpu-queries 0 --workload cfg4 --docs 100000 --doc-len 1024 --batch 32 --topk 100 > /dev/null 2>&1
python $R/tools/summarize_prof.py $(find /tmp/p4 -name run_kernel_stats.csv | head -1) $OUT/${TAG}_cfg4_kernel_stats.csv "bench.py --steps 10 --warmup 3 --docs 100000 --doc-len 1024 --batch 32 --topk 100 (cfg4), FP_GRAPH=0, MI355X"
head -16 $OUT/${TAG}_cfg4_kernel_stats.csv
