#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for LIM in 0 1; do
  out=gpurun_out/limprof_$LIM; rm -rf $out; mkdir -p $out
  CM=0 LIM=$LIM rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k -- python3 tools/probes/limit_cost_probe.py > $out/run.log 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "== LIM=$LIM"; tail -1 $out/run.log
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:9]:
    print("  %-46s calls %4s avg %8.1f us tot %7.1f ms"%(r['Name'].replace('void ','').replace('ilqr::','')[:46],r['Calls'],float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/1e6))
PY
done
