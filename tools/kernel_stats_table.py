"""Per-kernel table of a rocprofv3 --kernel-trace --stats CSV: python tools/kernel_stats_table.py <..._kernel_stats.csv>"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("ilqr::", "").replace("void ", "")
    if n.startswith("k_"):
        print("%-32s calls %4s avg %9.1f us min %9.1f max %9.1f" % (n[:32], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
