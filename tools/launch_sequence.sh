#!/bin/bash
# kernel sequence of ONE fp_search call at B = 1 and B = 64 (graph replay off), in launch order
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import fast_plaid_amd as fp
R = fp.fast_plaid_rust
B = int(sys.argv[1])
spec = fp.synth.SynthSpec(n_docs=1_000_000, doc_len=128, n_centroids=131072, dim=128, nbits=4, seed=42)
cent = fp.synth.centroids(spec)
ix = R.construct_synthetic_index(spec, "cuda:0", centroids=cent, bucket_weights=fp.synth.bucket_weights(spec))
R.set_graph_replay(False)
p = R.SearchParameters(2000, 4096, 1000, 8)
for i in range(4):
    R.search_arrays(ix, fp.synth.make_queries(spec, cent, B, 32, seed=10 + i), p)
PY
for B in 1 64; do
rm -rf /tmp/tr$B; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$B -o run -- python /tmp/one.py $B > /dev/null 2>&1
python - <<PY
import csv, glob, re
f = glob.glob("/tmp/tr$B/**/run_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:44] for r in rows]
# the last call: from the last k_pack_queries on
last = max(i for i, n in enumerate(names) if n.startswith("k_pack_queries"))
t0 = int(rows[last]["Start_Timestamp"])
print("B=$B: %d kernels in the last call" % (len(rows) - last))
for r, n in zip(rows[last:], names[last:]):
    print("%8.1f us +%6.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, n))
PY
done
