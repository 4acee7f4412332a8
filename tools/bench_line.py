"""One line per bench JSON: python tools/bench_line.py <file with the JSON line> -- value, ms per step, the early-exit and contact objects."""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
ee = d.get("early_exit") or {}
ct = d.get("contact") or {}
print(sys.argv[1], "value", round(d["value"]), "ms/step", round(d["ms_per_step"], 2), "ee", round(ee.get("value", 0)), "ee ms", ee.get("ms_per_step"), "contact", round(ct.get("value", 0)))
