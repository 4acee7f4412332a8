#!/usr/bin/env python3
"""Key numbers of a bench.py JSON line (any of its forms: default, --stage, fallback lines): python tools/bench_summary.py <log file>"""
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        ee, ct, fd = d.get("early_exit") or {}, d.get("contact") or {}, d.get("fd") or {}
        print("value %.0f %s  %.2f ms/step  early-exit %s  contact %s  fd %s" % (d.get("value", float("nan")), d.get("unit", ""), d.get("ms_per_step", float("nan")), ee.get("value"), ct.get("value"), fd.get("value")))
        if d.get("kernels"):
            print({k: (round(v.get("total_ms_per_step", 0.0), 2), round(v.get("avg_launch_ms", 0.0), 3)) for k, v in d["kernels"].items()})
        if d.get("stage_ms_per_step"):
            print({k: round(v, 2) for k, v in d["stage_ms_per_step"].items()})
        r = d.get("roofline") or {}
        print("roofline", r.get("kernel"), r.get("bound"), round(r.get("frac", float("nan")), 3), "whole-iteration", r.get("whole_iteration_frac"))
        if d.get("cpu_baseline"):
            print(d["cpu_baseline"])
