#!/bin/bash
# usage: tl.sh outname [env assignments...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
name=$1; shift
for kv in "$@"; do export "$kv"; done
export ILQR_SPLIT=0
out=gpurun_out/tl_$name
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/tr -o t -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-contact-line --no-fd-line > $out/bench.log 2>&1
f=$(find $out/tr -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py "$f" 16 > gpurun_out/tl_$name.txt
rm -rf $out
