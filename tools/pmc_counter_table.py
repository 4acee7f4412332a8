"""Largest per-launch value of every counter of a rocprofv3 --pmc CSV, per kernel: python tools/pmc_counter_table.py <..._counter_collection.csv>"""
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("ilqr::", "").replace("void ", "")
    acc[(n, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (n, c), v in sorted(acc.items()):
    if n.startswith("k_"):
        print("%-30s %-12s launches %3d max %.4g  (x32 B = %.3f GB; x64 B = %.3f GB)" % (n[:30], c, len(v), max(v), max(v) * 32 / 1e9, max(v) * 64 / 1e9))
