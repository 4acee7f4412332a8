#!/bin/bash
# Kernel duration of every timing-only variant library of the operand-layout Riccati kernel (make VAR=... pkvariant), bench size.
out=gpurun_out/${1:-riccati_variants}
mkdir -p "$out"
root=$PWD
cd /tmp && export TMPDIR=/tmp
for lib in "$root"/mpc-ilqr-mujoco_amd/lib/libilqr_hip.so "$root"/mpc-ilqr-mujoco_amd/lib/libilqr_hip_pk_*.so; do
  tag=$(basename "$lib" .so)
  ILQR_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/$tag" -o bw -- python3 "$root/tools/time_stage.py" backward 10 > "$root/$out/$tag.txt" 2>&1
  python3 - "$root/$out/$tag" "$tag" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "k_backward" in r["Name"]:
        print("%-28s %-24s avg %.4f ms min %.4f" % (sys.argv[2], r["Name"][:24], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6))
PY
done
