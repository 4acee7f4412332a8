#!/bin/bash
# Everything profiles/ keeps for a round, in one go on the GPU box (about ten minutes):  ILQR_GIT_HEAD=<sha> tools/collect_round.sh r05
# (the repository's .git does not travel to the box: the build container passes `git rev-parse --short HEAD` in for the stamps)
#   -> gpurun_out/profiles_<tag>/ : <tag>_kernel_stats.csv, traffic_latest.json, <tag>_pmc/*, <tag>_sq_issue_wait_summary.txt,
#      <tag>_fp64_instruction_mix.txt, <tag>_iteration_timeline.txt, <tag>_iteration_spans_fixed_and_early_exit.txt, <tag>_bench_lines/*.json
set -e
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles_$tag
bash tools/collect_profiles.sh "$tag" > /dev/null 2>&1 || true
echo "[collect] kernel stats, traffic, mfma done"
bash tools/sq_profile.sh "$tag" > /dev/null 2>&1 || true
cp gpurun_out/sq_$tag/sq_summary.txt "$out/${tag}_sq_issue_wait_summary.txt"
cp gpurun_out/sq_$tag/sq_counter_collection.csv "$out/${tag}_pmc/sq_counter_collection.csv"
echo "[collect] sq done"
BENCH1="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-contact-line --no-fd-line"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 \
  --output-format csv -d "$out/mix" -o x -- $BENCH1 > "$out/bench_mix.log" 2>&1
cp "$(find "$out/mix" -name '*counter_collection.csv' | head -1)" "$out/${tag}_pmc/fp64_mix_counter_collection.csv"; rm -rf "$out/mix"
python3 - "$out/${tag}_pmc/fp64_mix_counter_collection.csv" > "$out/${tag}_fp64_instruction_mix.txt" <<'PY'
import collections, csv, sys
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("ilqr::", "").replace("void ", "")
    rows[(name, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
best = {}
for (name, d), c in rows.items():
    if name not in best or c.get("SQ_INSTS_VALU", 0) > best[name].get("SQ_INSTS_VALU", 0):
        best[name] = c
print("# fp64 instruction mix of the largest (full-batch) launch of every kernel, one rocprofv3 pass (wave-level instruction counts;")
print("# exec_gflop = (ADD + MUL + 2 FMA) x 64 lanes + MOPS x 512: what the hardware issued if every lane were active)")
print("%-30s %11s %11s %11s %11s %11s %11s %6s %10s" % ("kernel", "valu", "add_f64", "mul_f64", "fma_f64", "trans_f64", "mfma_mops", "f64%", "exec_gflop"))
for name, c in sorted(best.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    if not name.startswith("k_"):
        continue
    g = lambda k: c.get(k, 0.0)
    a, m, f, t, v, mo = g("SQ_INSTS_VALU_ADD_F64"), g("SQ_INSTS_VALU_MUL_F64"), g("SQ_INSTS_VALU_FMA_F64"), g("SQ_INSTS_VALU_TRANS_F64"), g("SQ_INSTS_VALU"), g("SQ_INSTS_VALU_MFMA_MOPS_F64")
    print("%-30s %11.4g %11.4g %11.4g %11.4g %11.4g %11.4g %6.1f %10.1f" % (name[:30], v, a, m, f, t, mo, 100 * (a + m + f + t) / max(v, 1), ((a + m + 2 * f) * 64 + mo * 512) / 1e9))
PY
echo "[collect] fp64 mix done"
# (one group per concurrent region for this trace: with the early continuation of the convergence-exit leg the regions of consecutive
# iterations overlap and tools/timeline.py, which cuts at k_quad_kin, cannot attribute kernels to iterations)
export ILQR_SPLIT=0
rocprofv3 --kernel-trace --output-format csv -d "$out/tr" -o t -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-contact-line --no-fd-line > "$out/bench_trace.log" 2>&1
unset ILQR_SPLIT
f=$(find "$out/tr" -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py "$f" 16 > "$out/${tag}_iteration_timeline.txt"
{ echo "# ILQR_SPLIT=0 (one group per concurrent region, see tools/collect_round.sh); iterations 30.. are the convergence-exit leg:"; echo "# from the iteration whose pass holds <= 512 rollouts on, both lambda passes run side by side (two backward_wav / line_search_ entries, overlapping)"; python3 tools/timeline.py "$f" 5 all | sed -n '/^iter/,$p'; } > "$out/${tag}_iteration_spans_fixed_and_early_exit.txt"
rm -rf "$out/tr"
echo "[collect] timelines done"
# the contact workload's own passes: per-kernel time and HBM traffic of `bench.py --contact`
CB="python3 bench.py --contact --steps 1 --warmup 0 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/cst" -o k -- python3 bench.py --contact --steps 2 --warmup 1 --no-cpu-baseline > "$out/bench_contact_stats.log" 2>&1
cp "$(find "$out/cst" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_contact_kernel_stats.csv"; rm -rf "$out/cst"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/cf" -o f -- $CB > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/cw" -o w -- $CB > /dev/null 2>&1
cp "$(find "$out/cf" -name '*counter_collection.csv' | head -1)" "$out/${tag}_pmc/contact_fetch_size_counter_collection.csv"
cp "$(find "$out/cw" -name '*counter_collection.csv' | head -1)" "$out/${tag}_pmc/contact_write_size_counter_collection.csv"
rm -rf "$out/cf" "$out/cw"
python3 tools/pmc_summary.py --traffic-json "$out/${tag}_contact_traffic.json" --stamp "$(python3 bench.py --contact --print-signature)" \
  "$out/${tag}_pmc/contact_fetch_size_counter_collection.csv" "$out/${tag}_pmc/contact_write_size_counter_collection.csv" > /dev/null
echo "[collect] contact passes done"
mkdir -p "$out/${tag}_bench_lines"
L="$out/${tag}_bench_lines"
python3 bench.py 2>/dev/null | grep '^{' > "$L/bench_default.json"; echo "[collect] default line done"
python3 bench.py --contact --no-cpu-baseline 2>/dev/null | grep '^{' > "$L/bench_contact.json"
python3 bench.py --batch 1 --no-cpu-baseline --no-contact-line --steps 20 --warmup 3 2>/dev/null | grep '^{' > "$L/bench_b1.json"
ILQR_SPEC=0 python3 bench.py --batch 1 --no-cpu-baseline --no-contact-line --steps 20 --warmup 3 2>/dev/null | grep '^{' > "$L/bench_b1_sequential_retry.json"
python3 bench.py --batch 256 --no-cpu-baseline --no-contact-line 2>/dev/null | grep '^{' > "$L/bench_b256.json"
ILQR_SPEC=0 python3 bench.py --batch 256 --no-cpu-baseline --no-contact-line 2>/dev/null | grep '^{' > "$L/bench_b256_sequential_retry.json"
ILQR_SPEC=0 ILQR_SPLIT=0 python3 bench.py --no-cpu-baseline --no-contact-line 2>/dev/null | grep '^{' > "$L/bench_default_sequential_retry_one_group.json"
python3 bench.py --stage rollout_jacobians --batch 1024 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | grep '^{' > "$L/bench_cfg1.json"
python3 bench.py --batch 1024 --no-cpu-baseline --no-contact-line 2>/dev/null | grep '^{' > "$L/bench_b1024.json"
python3 bench.py --batch 1024 --horizon 50 --no-cpu-baseline --no-contact-line 2>/dev/null | grep '^{' > "$L/bench_b1024_n50.json"
python3 bench.py --batch 8192 --no-cpu-baseline --no-contact-line 2>/dev/null | grep '^{' > "$L/bench_b8192.json"
echo "[collect] small lines done"
python3 bench.py --workload config3 --steps 3 --no-cpu-baseline 2>/dev/null | grep '^{' > "$L/bench_config3_global_batch_one_gpu.json"
python3 bench.py --workload config4 --steps 3 2>/dev/null | grep '^{' > "$L/bench_config4_global_batch_one_gpu.json"
# every stage kernel alone, full batch (stage API): what "alone ms" in the summary means
rm -f "$out/${tag}_alone_kernel_stats.csv"
for st in linearize quadratics backward line_search rollout; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/alone_$st" -o a -- python3 tools/time_stage.py $st 5 > "$out/alone_$st.log" 2>&1 || true
  f=$(find "$out/alone_$st" -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$st" "$out/${tag}_alone_kernel_stats.csv" <<'PY'
import csv, os, sys
# keep the rows of the kernels the stage itself launches (the set-up before it launches other stages' kernels once)
keep = {"linearize": ("k_lin_primal", "k_lin_tangent"), "quadratics": ("k_quad_kin", "k_cost_quadratics"), "backward": ("k_backward",),
        "line_search": ("k_line_search", "k_traj_knot_cost"), "rollout": ("k_rollout", "k_traj_cost_sum")}[sys.argv[2]]
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Name"] for k in keep)]
new = not os.path.exists(sys.argv[3])
w = csv.DictWriter(open(sys.argv[3], "a", newline=""), fieldnames=["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"], quoting=csv.QUOTE_NONNUMERIC)
if new: w.writeheader()
have = set()
if not new: have = {r["Name"] for r in csv.DictReader(open(sys.argv[3]))}
for r in rows:
    if r["Name"] not in have: w.writerow({k: r[k] for k in w.fieldnames})
PY
  rm -rf "$out/alone_$st"
done
echo "[collect] stage kernels alone done"
# address-unit counters of every kernel of a bench step (two passes: the TA block offers two counters at a time)
bash tools/pmc_bench.sh "TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" ta1_$tag > /dev/null 2>&1 || true
bash tools/pmc_bench.sh "TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" ta2_$tag > /dev/null 2>&1 || true
python3 tools/ta_table.py gpurun_out/ta1_$tag/c_counter_collection.csv gpurun_out/ta2_$tag/c_counter_collection.csv > "$out/${tag}_address_unit_counters.txt" 2>/dev/null || true
echo "[collect] address-unit counters done"
python3 tools/kernel_resources.py > "$out/${tag}_kernel_resources.txt" 2>/dev/null || true
( cd mpc-ilqr-mujoco_amd/lib && ls -l libilqr_hip.so libilqr_hip_legacy.so | awk '{print $5, $9}' ) > "$out/${tag}_library_sizes.txt"
python3 tools/round_summary.py "$out" "$tag" "${ILQR_GIT_HEAD:-unknown}" > "$out/${tag}_summary.md"
echo "[collect] all done"
ls "$out" "$L"
