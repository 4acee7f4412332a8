#!/usr/bin/env python3
"""Per-kernel issue / wait split from one rocprofv3 --pmc pass of the SQ counters (tools/sq_profile.sh).

For the LARGEST launch of every kernel (= the full-batch launch): wave quad-cycles, the share parked on s_waitcnt / barriers
(SQ_WAIT_ANY), stalled at issue (SQ_WAIT_INST_ANY), issuing (SQ_ACTIVE_INST_ANY), the VALU and LDS shares of the issuing
time, VALU instructions and LDS bank-conflict cycles.
"""
import collections
import csv
import sys

rows = collections.defaultdict(dict)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("ilqr::", "")
        if name.startswith("void "):
            name = name[5:]
        rows[(name, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
best = {}
for (name, disp), c in rows.items():
    if "SQ_WAVE_CYCLES" in c and (name not in best or c["SQ_WAVE_CYCLES"] > best[name]["SQ_WAVE_CYCLES"]):
        best[name] = c
print("%-28s %12s %7s %7s %7s %7s %7s %12s %10s" % ("kernel (largest launch)", "wave_qcyc", "wait%", "stall%", "issue%", "valu%", "lds%", "insts_valu", "bankconf"))
for name, c in sorted(best.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    if not name.startswith("k_"):
        continue
    w = c["SQ_WAVE_CYCLES"] or 1.0
    g = lambda k: c.get(k, 0.0)
    print("%-28s %12.4g %7.1f %7.1f %7.1f %7.1f %7.1f %12.4g %10.4g" % (
        name[:28], w, 100 * g("SQ_WAIT_ANY") / w, 100 * g("SQ_WAIT_INST_ANY") / w, 100 * g("SQ_ACTIVE_INST_ANY") / w,
        100 * g("SQ_ACTIVE_INST_VALU") / w, 100 * g("SQ_ACTIVE_INST_LDS") / w, g("SQ_INSTS_VALU"), g("SQ_LDS_BANK_CONFLICT")))
