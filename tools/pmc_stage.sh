#!/bin/bash
# PMC counters of one stage at the bench size: tools/pmc_stage.sh <stage> "<counters>" [outname]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
stage=$1; ctrs=$2; name=${3:-pmc_$stage}
out=gpurun_out/$name
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --pmc $ctrs --output-format csv -d "$out" -o c -- python3 tools/time_stage.py "$stage" 2 > "$out/run.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    agg[r["Kernel_Name"].replace("void ", "").replace("ilqr::", "").split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if k.startswith("k_"):
        print(k, " ".join("%s=%.4g" % (c, max(x)) for c, x in sorted(v.items())))
PY
