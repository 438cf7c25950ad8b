#!/bin/bash
# round 3, lab 3: where does k_maxsim6's time go: timing-only ablations (1 no normalisation, 2 no decode + no normalisation,
# 3 no MFMA, 4 no decode / normalisation / MFMA = loads + epilogue only) and SQ counters
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
cp fast-plaid_amd/libfastplaid_hip.so /tmp/lib_orig.so
for f in /tmp/lib_orig.so tools/libs/lib_ms6abl*.so; do
  cp $f fast-plaid_amd/libfastplaid_hip.so
  timeout 200 python bench.py --steps 10 --warmup 3 --cpu-queries 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$(basename $f)', 'maxsim_ms=%.4f repair_ms=%.4f step_ms=%.3f' % (d['stages_ms']['S6+S7 maxsim'], d['stages_ms']['S7 order repair'], d['ms_per_step']))" | tee -a $OUT/r3_ms6abl.txt
done
cp /tmp/lib_orig.so fast-plaid_amd/libfastplaid_hip.so
bash tools/pmc_scan.sh k_maxsim6 2>&1 | tee $OUT/r3_ms6_pmc.txt | cut -c1-400
