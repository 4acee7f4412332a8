"""Instruction-cache hit / miss counters per kernel (largest launch) of a pass
   rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ ...:
   python tools/icache_table.py <..._counter_collection.csv>   (round 4: every kernel hits in > 99.8 % of its fetches -- none is fetch-bound)"""
import csv, sys, collections
acc = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("ilqr::", "").replace("void ", "")
    key = (n, r["Dispatch_Id"])
    acc[key][r["Counter_Name"]] = float(r["Counter_Value"])
best = {}
for (n, d), c in acc.items():
    if n not in best or c.get("SQC_ICACHE_REQ", 0) > best[n].get("SQC_ICACHE_REQ", 0):
        best[n] = c
print("%-30s %12s %12s %12s %8s %12s %12s" % ("kernel (largest launch)", "icache_req", "hits", "misses", "miss%", "miss_dup", "tc_inst_req"))
for n, c in sorted(best.items(), key=lambda kv: -kv[1].get("SQC_ICACHE_REQ", 0)):
    if n.startswith("k_"):
        g = lambda k: c.get(k, 0.0)
        print("%-30s %12.4g %12.4g %12.4g %8.1f %12.4g %12.4g" % (n[:30], g("SQC_ICACHE_REQ"), g("SQC_ICACHE_HITS"), g("SQC_ICACHE_MISSES"), 100 * g("SQC_ICACHE_MISSES") / max(g("SQC_ICACHE_REQ"), 1), g("SQC_ICACHE_MISSES_DUPLICATE"), g("SQC_TC_INST_REQ")))
