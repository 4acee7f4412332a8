#!/usr/bin/env python3
"""Key numbers of a bench.py JSON line: python tools/bench_summary.py <log file>"""
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print("value %.0f it/s  %.2f ms/step  early-exit %.0f  contact %s" % (d["value"], d["ms_per_step"], d["early_exit"]["value"], d.get("contact", {}).get("value")))
        print({k: (round(v["total_ms_per_step"], 2), round(v["avg_launch_ms"], 3)) for k, v in d["kernels"].items()})
        print({k: round(v, 2) for k, v in d["stage_ms_per_step"].items()})
        print("roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), "whole-iteration", round(d["roofline"]["whole_iteration_frac"], 3))
        if "cpu_baseline" in d:
            print(d["cpu_baseline"])
