#!/usr/bin/env python3
"""Kernel timeline of one iLQR iteration from a rocprofv3 kernel_trace.csv: start / end relative to the iteration's first
kernel, the gap since the previous kernel END on any stream, and the union of busy time.

  python tools/timeline.py <kernel_trace.csv> [iteration index, default 5]
"""
import csv, sys
path = sys.argv[1]
it = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = [r for r in csv.DictReader(open(path))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# an iteration starts at each k_cost_quadratics launch that follows a k_control
marks = [i for i, r in enumerate(rows) if "k_quad_kin" in r["Kernel_Name"]]
lo = marks[it]
# include the kernels that started a little before (lin primal / rollout of the same fork)
while lo > 0 and "k_control" not in rows[lo - 1]["Kernel_Name"] and "k_solve_begin" not in rows[lo - 1]["Kernel_Name"]:
    lo -= 1
hi = marks[it + 1]
while hi > lo and "k_control" not in rows[hi - 1]["Kernel_Name"]:
    hi -= 1
t0 = int(rows[lo]["Start_Timestamp"])
last_end = t0
busy, cur_s, cur_e = 0, None, None
print("%-34s %9s %9s %9s %8s" % ("kernel", "start_us", "end_us", "dur_us", "gap_us"))
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("void ", "").replace("ilqr::", "").split("(")[0]
    print("%-34s %9.1f %9.1f %9.1f %8.1f" % (name[:34], s / 1e3, e / 1e3, (e - s) / 1e3, (s - (last_end - t0)) / 1e3))
    last_end = max(last_end, e + t0)
    if cur_s is None: cur_s, cur_e = s, e
    elif s <= cur_e: cur_e = max(cur_e, e)
    else: busy += cur_e - cur_s; cur_s, cur_e = s, e
busy += cur_e - cur_s
print("iteration span %.1f us, union of kernel time %.1f us, idle %.1f us" % ((last_end - t0) / 1e3, busy / 1e3, (last_end - t0 - busy) / 1e3))
if len(sys.argv) > 3 and sys.argv[3] == "all":
    # one line per iteration: span and the duration of each full-batch kernel in it
    print()
    print("%4s %9s  %s" % ("iter", "span_us", "kernel durations (us)"))
    for k in range(len(marks) - 1):
        lo, hi = marks[k], marks[k + 1]
        while lo > 0 and "k_control" not in rows[lo - 1]["Kernel_Name"] and "k_solve_begin" not in rows[lo - 1]["Kernel_Name"]:
            lo -= 1
        while hi > lo and "k_control" not in rows[hi - 1]["Kernel_Name"]:
            hi -= 1
        seg = rows[lo:hi]
        if not seg:
            continue
        a, b = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
        parts = []
        for r in seg:
            nm = r["Kernel_Name"].replace("void ", "").replace("ilqr::", "").split("(")[0].split("<")[0]
            if nm in ("k_lin_primal_r", "k_lin_primal_s", "k_lin_tangent", "k_lin_tangent2", "k_lin_tangent2c", "k_quad_kin", "k_cost_quadratics", "k_backward_wave", "k_backward_pack", "k_line_search_s", "k_rollout_s"):
                parts.append("%s %.0f" % (nm[2:14], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        print("%4d %9.1f  %s" % (k, (b - a) / 1e3, "  ".join(parts)))
