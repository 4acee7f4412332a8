#!/usr/bin/env python3
"""Per-launch durations of one kernel from a rocprofv3 kernel_trace.csv, in launch order.

  python tools/trace_launches.py <kernel_trace.csv> k_backward_wave [n_fixed] > profiles/rNN_<kernel>_launches.csv

bench.py's `roofline.avg_launch_ms` is the average of the FULL-BATCH launches of the dominant kernel (the first pass of every
iteration); `rocprofv3 --stats` averages those together with the lambda-retry launches (a subset of the rollouts) and with the
launches of the early-exit step.  This lists every launch so that the two can be compared: launches alternate first pass /
retry; the first `n_fixed` launches (default: 2 per iteration x 10 iterations x 3 steps of the profiled command) belong to
the fixed-iteration steps.
"""
import csv, sys, statistics as st
path, kernel = sys.argv[1], sys.argv[2]
n_fixed = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rows = [r for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print("launch,phase,pass,duration_ms")
for i, v in enumerate(d):
    print("%d,%s,%s,%.6f" % (i, "fixed-iterations" if i < n_fixed else "early-exit", "first" if i % 2 == 0 else "retry", v))
full = d[:n_fixed][0::2]
sys.stderr.write("%s: %d launches; first pass of the fixed-iteration steps: avg %.3f ms (min %.3f, max %.3f); retry avg %.3f ms; all %.3f ms\n"
                 % (kernel, len(d), st.mean(full), min(full), max(full), st.mean(d[:n_fixed][1::2]), st.mean(d)))
