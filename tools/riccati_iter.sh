#!/bin/bash
# One round of the Riccati kernel's edit-measure loop on the GPU box: parity against the generic kernel, in-kernel phase stamps, and the
# kernel's duration at the bench size from a rocprofv3 kernel trace.   tools/riccati_iter.sh <outdir under gpurun_out>
out=gpurun_out/${1:-riccati_iter}
mkdir -p "$out"
python3 tools/pack_check.py 64 25 > "$out/check_free.txt" 2>&1; tail -2 "$out/check_free.txt"
ILQR_CONTACT=2 python3 tools/pack_check.py 16 25 > "$out/check_c2.txt" 2>&1; tail -1 "$out/check_c2.txt"
if [ -f mpc-ilqr-mujoco_amd/lib/libilqr_hip_wstamp.so ]; then
  ILQR_HIP_LIB=$PWD/mpc-ilqr-mujoco_amd/lib/libilqr_hip_wstamp.so ILQR_WSTAMPS=1 python3 tools/time_stage.py backward 5 > "$out/stamps.txt" 2>&1; tail -13 "$out/stamps.txt"
fi
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/prof" -o bw -- python3 "$root/tools/time_stage.py" backward 20 > "$root/$out/ts.txt" 2>&1
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/prof/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "k_backward" in r["Name"]:
        print("%-40s calls %s avg %.4f ms min %.4f max %.4f" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
PY
