#!/bin/bash
# rocprofv3 per-kernel stats of one stage at the bench size: tools/prof_stage.sh <stage> [outdir]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=${2:-gpurun_out/prof_stage}
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o s -- python3 tools/time_stage.py "$1" 5 > "$out/run.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(fs[0])):
    print("%-70s calls %4s avg %10.1f us max %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
