#!/bin/bash
# PMC counters of every kernel of one bench step (the solve's own launches: operand layout, concurrent region):
# tools/pmc_bench.sh "<counters>" [outname] [bench args...]   (at most two counters of one block per pass; bounded by `timeout`)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ctrs=$1; name=${2:-pmc_bench}; shift; shift
out=gpurun_out/$name
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 ${PMC_TIMEOUT:-200} rocprofv3 --pmc $ctrs --output-format csv -d "$out" -o c -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-contact-line --no-fd-line "$@" > "$out/run.log" 2>&1
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
