#!/usr/bin/env python3
"""Address-unit (TA) view of every kernel of one bench step, from two rocprofv3 --pmc passes of tools/pmc_bench.sh:
   pass 1: TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
   pass 2: TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum
   python tools/ta_table.py <pass1 counter_collection.csv> <pass2 counter_collection.csv>
Per kernel the launch with the largest GRBM_GUI_ACTIVE (the full-batch launch).  busy % = TA_TA_BUSY_sum / (256 TAs x GRBM_GUI_ACTIVE / 8):
GRBM_GUI_ACTIVE comes summed over the 8 XCDs, the TA counters over the 256 CUs' address units."""
import csv, sys, collections
def largest(path, key):
    acc = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].split("(")[0].replace("ilqr::", "").replace("void ", "")
        acc[(n, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    best = {}
    for (n, d), c in acc.items():
        if n not in best or c.get(key, 0) > best[n].get(key, 0):
            best[n] = c
    return best
a = largest(sys.argv[1], "GRBM_GUI_ACTIVE")
b = largest(sys.argv[2], "TCP_TCC_WRITE_REQ_sum")
print("# address unit (TA) and L1 -> L2 requests per kernel, largest launch of one bench step (tools/ta_table.py)")
print("%-30s %9s %8s %14s %14s %12s %12s" % ("kernel", "cycles", "TA busy%", "data stalled%", "addr stalled%", "L2 write req", "L2 read req"))
for n, c in sorted(a.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    if not n.startswith("k_") or c.get("GRBM_GUI_ACTIVE", 0) < 2e5:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    d = b.get(n, {})
    ta = 256.0 * cyc
    print("%-30s %9.3g %8.1f %14.1f %14.1f %12.4g %12.4g" % (n[:30], cyc, 100 * c.get("TA_TA_BUSY_sum", 0) / ta, 100 * c.get("TA_DATA_STALLED_BY_TC_CYCLES_sum", 0) / ta,
          100 * d.get("TA_ADDR_STALLED_BY_TC_CYCLES_sum", 0) / ta, d.get("TCP_TCC_WRITE_REQ_sum", 0), d.get("TCP_TCC_READ_REQ_sum", 0)))
