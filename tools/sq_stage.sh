#!/bin/bash
# SQ counters of ONE stage at the bench size (run on the GPU box): tools/sq_stage.sh <stage> <tag> -> gpurun_out/sqs_<tag>/summary.txt
# (instruction counts by type, the wait / stall / issue split of the wave cycles, LDS bank conflicts; quad-cycles)
set -e
stage=${1:-quadratics}; tag=${2:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sqs_$tag
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT \
  --output-format csv -d "$out/sq" -o s -- python3 tools/time_stage.py "$stage" 2 > "$out/run.log" 2>&1
cp "$(find "$out/sq" -name '*counter_collection.csv' | head -1)" "$out/counters.csv"
rm -rf "$out/sq"
python3 - "$out/counters.csv" > "$out/summary.txt" <<'PY'
import collections, csv, sys
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("ilqr::", "").replace("void ", "")
    rows[(name, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
best = {}
for (name, d), c in rows.items():
    if "SQ_WAVE_CYCLES" in c and (name not in best or c["SQ_WAVE_CYCLES"] > best[name]["SQ_WAVE_CYCLES"]):
        best[name] = c
print("%-26s %11s %6s %6s %6s %11s %11s %11s %10s" % ("kernel (largest launch)", "wave_qcyc", "wait%", "stall%", "issue%", "valu", "salu", "lds", "bankconf"))
for name, c in sorted(best.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    if not name.startswith("k_"):
        continue
    w = c["SQ_WAVE_CYCLES"] or 1.0
    g = lambda k: c.get(k, 0.0)
    print("%-26s %11.4g %6.1f %6.1f %6.1f %11.4g %11.4g %11.4g %10.4g" % (name[:26], w, 100 * g("SQ_WAIT_ANY") / w, 100 * g("SQ_WAIT_INST_ANY") / w,
          100 * g("SQ_ACTIVE_INST_ANY") / w, g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_INSTS_LDS"), g("SQ_LDS_BANK_CONFLICT")))
PY
cat "$out/summary.txt"
