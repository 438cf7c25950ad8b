#!/bin/bash
# round 3, lab 1: instruction issue rates, gather patterns, timing-only ablations of k_maxsim5 (tools/libs/lib_msabl*.so)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R/tools/probe
timeout 120 ./issue_rate.bin > $OUT/r3_issue_rate.txt 2>&1; cat $OUT/r3_issue_rate.txt
timeout 200 ./gather_lab.bin > $OUT/r3_gather_lab.txt 2>&1; cat $OUT/r3_gather_lab.txt
cd $R
cp fast-plaid_amd/libfastplaid_hip.so /tmp/lib_orig.so
for f in /tmp/lib_orig.so tools/libs/lib_msabl*.so; do
  cp $f fast-plaid_amd/libfastplaid_hip.so
  timeout 200 python bench.py --steps 10 --warmup 3 --cpu-queries 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$(basename $f)', 'maxsim_ms=%.4f repair_ms=%.4f step_ms=%.3f' % (d['stages_ms']['S6+S7 maxsim'], d['stages_ms']['S7 order repair'], d['ms_per_step']))" | tee -a $OUT/r3_msabl.txt
done
cp /tmp/lib_orig.so fast-plaid_amd/libfastplaid_hip.so
