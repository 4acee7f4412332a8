#!/bin/bash
# Issue / wait characterisation of every kernel of the bench command (run on the GPU box):
#   tools/sq_profile.sh <tag>   -> gpurun_out/sq_<tag>/sq_summary.txt
# SQ counters in ONE pass (8 slots, MI355X_MICROARCH.md "rocprofv3 PMC slots"): SQ_WAVE_CYCLES = SQ_WAIT_ANY (parked on
# s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY (issuing), all in quad-cycles.
set -e
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sq_$tag
rm -rf "$out"; mkdir -p "$out"
BENCH1="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-contact-line --no-fd-line ${SQ_BENCH_ARGS}"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
  --output-format csv -d "$out/sq" -o s -- $BENCH1 > "$out/bench_sq.log" 2>&1
cp "$(find "$out/sq" -name '*counter_collection.csv' | head -1)" "$out/sq_counter_collection.csv"
rm -rf "$out/sq"
python3 tools/sq_summary.py "$out/sq_counter_collection.csv" > "$out/sq_summary.txt"
cat "$out/sq_summary.txt"
