#!/bin/bash
# Builds the variant library with the rows-through-LDS MaxSim kernel (tools/probe/maxsim7_lab.hip) and times both on cfg2.
# usage (GPU box): bash tools/maxsim7_lab.sh      -> gpurun_out/maxsim7_lab.txt
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
C=$R/fast-plaid_amd/csrc; V=/tmp/fp_variant; mkdir -p $V $R/gpurun_out
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$C -c $R/tools/probe/maxsim7_lab.hip -o $V/fp_maxsim_lab.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/fp_kernels.o $V/fp_maxsim_lab.o $C/fp_synth.o $C/fp_engine.o -o $V/libfastplaid_lab.so || exit 1
cd $R
( echo "== product (k_maxsim6)"; python bench.py --cpu-queries 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms']['S6+S7 maxsim'], d['parity_vs_cpu'] if 'parity_vs_cpu' in d else '')"
  echo "== variant (k_maxsim7)"; FP_LIB_PATH=$V/libfastplaid_lab.so python bench.py --cpu-queries 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms']['S6+S7 maxsim'])"
) | tee $R/gpurun_out/maxsim7_lab.txt
