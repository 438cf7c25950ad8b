#!/bin/bash
# round 3, lab 5: k_maxsim6 occupancy variants (12 waves; 2 workgroups of 8 waves per CU), XCD mapping, ablations and counters on the rinv kernel
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
cp fast-plaid_amd/libfastplaid_hip.so /tmp/lib_orig.so
one() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 10 --warmup 3 --cpu-queries 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$tag', 'maxsim_ms=%.4f repair_ms=%.4f step_ms=%.3f' % (d['stages_ms']['S6+S7 maxsim'], d['stages_ms']['S7 order repair'], d['ms_per_step']))" | tee -a $OUT/r3_lab5.txt; }
one w16 FP_X=1
one w16_xcd FP_MS_XCD=1
for f in tools/libs/lib_ms6_*.so; do cp $f fast-plaid_amd/libfastplaid_hip.so; one $(basename $f .so) FP_X=1; done
cp /tmp/lib_orig.so fast-plaid_amd/libfastplaid_hip.so
bash tools/pmc_scan.sh k_maxsim6 2>&1 | tee $OUT/r3_ms6_pmc2.txt | cut -c1-400
