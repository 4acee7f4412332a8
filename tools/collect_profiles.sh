#!/bin/bash
# Collects the rocprofv3 summaries the bench numbers are judged against (run on the GPU box):
#   tools/collect_profiles.sh <tag>      -> gpurun_out/profiles_<tag>/  (copy what is to be kept into profiles/)
# 1. per-kernel time of the bench command; 2. HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes, as the microarch guide
# prescribes) of the same command -> traffic JSON stamped with the run signature; 3. MFMA counters of the Riccati kernel.
set -e
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles_$tag
rm -rf "$out"; mkdir -p "$out"
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-contact-line --no-fd-line"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o k -- $BENCH > "$out/bench_stats.log" 2>&1
cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_kernel_stats.csv"
BENCH1="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-contact-line --no-fd-line"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o f -- $BENCH1 > "$out/bench_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -o w -- $BENCH1 > "$out/bench_write.log" 2>&1
mkdir -p "$out/${tag}_pmc"
cp "$(find "$out/fetch" -name '*counter_collection.csv' | head -1)" "$out/${tag}_pmc/fetch_size_counter_collection.csv"
cp "$(find "$out/write" -name '*counter_collection.csv' | head -1)" "$out/${tag}_pmc/write_size_counter_collection.csv"
python3 tools/pmc_summary.py --traffic-json "$out/traffic_latest.json" --stamp "$(python3 bench.py --print-signature)" \
  "$out/${tag}_pmc/fetch_size_counter_collection.csv" "$out/${tag}_pmc/write_size_counter_collection.csv" > "$out/traffic.log"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES --output-format csv \
  -d "$out/mfma" -o m -- python3 tools/time_stage.py backward 3 > "$out/mfma.log" 2>&1
cp "$(find "$out/mfma" -name '*counter_collection.csv' | head -1)" "$out/${tag}_pmc/mfma_backward_wave_counter_collection.csv"
python3 tools/pmc_summary.py "$out/${tag}_pmc/mfma_backward_wave_counter_collection.csv" | grep "k_backward_\(wave\|pack\)" > "$out/mfma_summary.txt" || true
rm -rf "$out/stats" "$out/fetch" "$out/write" "$out/mfma"
tail -1 "$out/bench_stats.log" | cut -c1-200
head -8 "$out/${tag}_kernel_stats.csv" | cut -c1-160
cat "$out/mfma_summary.txt"
