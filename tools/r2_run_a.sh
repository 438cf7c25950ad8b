#!/bin/bash
# round-2 evidence run A: GPU tests, bench with the level-0 S4, parameter sweep, kernel trace, FETCH_SIZE calibration
TAG=${1:-r02_a}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/${TAG}_gpu_tests.log
cat $OUT/${TAG}_gpu_tests.log
timeout 400 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -3 $OUT/${TAG}_bench.err; cat $OUT/${TAG}_bench.json
for impl in q8; do
  FP_APPROX_IMPL=$impl timeout 200 python bench.py --cpu-queries 0 > $OUT/${TAG}_bench_$impl.json 2>> $OUT/${TAG}_bench.err
done
for tail in 0.01 0.05 0.1; do
  FP_L0_TAIL=$tail timeout 200 python bench.py --cpu-queries 0 --steps 10 --warmup 3 > $OUT/${TAG}_bench_tail$tail.json 2>> $OUT/${TAG}_bench.err
done
for cpw in 256 512 2048 4096; do
  FP_L0_CPW=$cpw timeout 200 python bench.py --cpu-queries 0 --steps 10 --warmup 3 > $OUT/${TAG}_bench_cpw$cpw.json 2>> $OUT/${TAG}_bench.err
done
FP_L0_PILOT=2 timeout 200 python bench.py --cpu-queries 0 --steps 10 --warmup 3 > $OUT/${TAG}_bench_pilot2.json 2>> $OUT/${TAG}_bench.err
python - <<PY
import json, glob, os
for p in sorted(glob.glob("$OUT/${TAG}_bench*.json")):
    try:
        d = json.load(open(p))
    except Exception as e:
        print(os.path.basename(p), "unreadable", e); continue
    st = d.get("stages_ms", {})
    print(os.path.basename(p), "qps %.0f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in st.items()},
          "exact-rescored", d.get("roofline", {}).get("docs_rescored_exactly_per_batch"))
PY
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 5 --cpu-queries 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o run -- $CMD > $OUT/${TAG}_prof.log 2>&1
python $R/tools/summarize_prof.py $(ls $OUT/${TAG}_prof/*/run_kernel_stats.csv $OUT/${TAG}_prof/run_kernel_stats.csv 2>/dev/null | head -1) \
    $OUT/${TAG}_kernel_stats.csv "bench.py --steps 20 --warmup 5 (cfg2: 1M docs, 64 queries/batch), MI355X"
head -30 $OUT/${TAG}_kernel_stats.csv
# FETCH_SIZE calibration on known byte counts
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value $R/tools/probe/fetch_calib.hip -o /tmp/fetch_calib.bin
timeout 120 /tmp/fetch_calib.bin > $OUT/${TAG}_fetch_calib.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/${TAG}_calib_pmc -o run -- /tmp/fetch_calib.bin > $OUT/${TAG}_calib_pmc.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $OUT/${TAG}_calib_pmc2 -o run -- /tmp/fetch_calib.bin >> $OUT/${TAG}_calib_pmc.log 2>&1
python - <<PY
import csv, glob, re, json
known = {}
for line in open("$OUT/${TAG}_fetch_calib.txt"):
    m = re.match(r"(k_calib_gather<\d+>) rows=(\d+) row_bytes=(\d+) line_stride=(\d+)", line)
    if m: known[m.group(1)] = {"rows": int(m.group(2)), "row_bytes": int(m.group(3)), "line_stride": int(m.group(4)), "line": line.strip()}
    m = re.match(r"(k_calib_stream16) bytes=(\d+)", line)
    if m: known[m.group(1)] = {"bytes": int(m.group(2)), "line": line.strip()}
res = {}
for p in sorted(glob.glob("$OUT/${TAG}_calib_pmc*/**/run_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(p)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if k in known: res.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
out = {}
for k, c in res.items():
    e = dict(known[k]); e.update(c)
    if "FETCH_SIZE" in c:
        raw = c["FETCH_SIZE"] * 1024
        if "rows" in e:
            e["raw_bytes_per_row"] = raw / e["rows"]
        else:
            e["raw_over_known"] = raw / e["bytes"]
    out[k] = e
json.dump(out, open("$OUT/${TAG}_fetch_calibration.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
# PMC traffic of the bench kernels (separate passes)
CMD3="python $R/bench.py --steps 3 --warmup 1 --cpu-queries 0"
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/${TAG}_pmc_fetch -o run -- $CMD3 > $OUT/${TAG}_pmc.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/${TAG}_pmc_write -o run -- $CMD3 >> $OUT/${TAG}_pmc.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $OUT/${TAG}_pmc_mfma -o run -- $CMD3 >> $OUT/${TAG}_pmc.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/${TAG}_pmc_sq -o run -- $CMD3 >> $OUT/${TAG}_pmc.log 2>&1
python - <<PY
import csv, glob, collections, json, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for p in sorted(glob.glob("$OUT/${TAG}_pmc_*/**/run_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(p)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:60]
        if not k.startswith("k_"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
out = {}
for k, cs in acc.items():
    d = {c: v / max(n[(k, c)], 1) for c, v in cs.items()}
    e = {"launches_sampled": max(n[(k, c)] for c in cs)}
    if "FETCH_SIZE" in d:
        e["fetch_size_kb_raw"] = round(d["FETCH_SIZE"], 1)
        e["hbm_read_bytes_corrected"] = int(d["FETCH_SIZE"] * 1024 * 2)
    if "WRITE_SIZE" in d:
        e["write_size_kb_raw"] = round(d["WRITE_SIZE"], 1)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("GRBM_GUI_ACTIVE", 0) > 0:
        e["mfma_busy_cycles"] = int(d["SQ_VALU_MFMA_BUSY_CYCLES"])
        e["gpu_active_cycles_sum_over_8_xcd"] = int(d["GRBM_GUI_ACTIVE"])
        e["mfma_util"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
    if "TCC_HIT_sum" in d:
        e["l2_hit_rate"] = round(d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d.get("TCC_MISS_sum", 0.0), 1.0), 4)
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
        if c in d: e[c.lower()] = int(d[c])
    out[k] = e
json.dump({"command": "$CMD3", "note": "per-launch averages; FETCH_SIZE doubled per the guide's gfx950 correction (see *_fetch_calibration.json for row gathers); WRITE_SIZE uncalibrated",
           "kernels": out}, open("$OUT/${TAG}_pmc_traffic.json", "w"), indent=1, sort_keys=True)
for k in sorted(out): print(k, out[k])
PY
