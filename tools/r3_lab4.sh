#!/bin/bash
# round 3, lab 4: k_maxsim6 with the one-multiply normalisation: bench A/B (FP_MS_RINV=0: compensated quotient), hard-token count, GPU tests
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --cpu-queries ${CPUQ:-0} 2>$OUT/r3_lab4_$tag.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$tag', 'maxsim_ms=%.4f repair_ms=%.4f step_ms=%.3f qps=%.0f' % (d['stages_ms']['S6+S7 maxsim'], d['stages_ms']['S7 order repair'], d['ms_per_step'], d['value']), d.get('parity_vs_cpu'))" || tail -5 $OUT/r3_lab4_$tag.err; }
CPUQ=64 run rinv FP_X=1
run norinv FP_MS_RINV=0
run v5 FP_MAXSIM_IMPL=5
python - <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import fast_plaid_amd as fp
R = fp.fast_plaid_rust
spec = fp.synth.SynthSpec(n_docs=1_000_000, doc_len=128, n_centroids=131072, dim=128, nbits=4, seed=42)
ix = R.construct_synthetic_index(spec, "cuda:0", centroids=fp.synth.centroids(spec), bucket_weights=fp.synth.bucket_weights(spec))
print("hard tokens:", ix.n_hard_tokens, "of", ix.n_docs * 128, "bytes", ix.device_bytes)
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/r3_lab4_tests.log 2>&1; tail -15 $OUT/r3_lab4_tests.log | cut -c1-300
