#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters: python tools/pmc_summary.py <counter_collection.csv> [...]"""
import csv, sys, collections
for path in sys.argv[1:]:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        v = sorted(v)
        print("%-40s %-22s n=%4d  mean=%.6g  max=%.6g" % (k, c, len(v), sum(v) / len(v), v[-1]))
