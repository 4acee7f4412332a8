#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 --pmc counter_collection.csv files.

  python tools/pmc_summary.py <counter_collection.csv> [...]            # table: n, mean, max per (kernel, counter)
  python tools/pmc_summary.py --traffic-json OUT [--stamp "$(python bench.py --print-signature ...)"] <fetch.csv> <write.csv>
      # per kernel: FETCH_SIZE / WRITE_SIZE of the
      largest launch (= the full-batch launch; the lambda-retry launches of the same kernel run a subset), in KiB as
      rocprofv3 reports them.  bench.py turns them into bytes: 2 x FETCH_SIZE (gfx950 counts 64 B per 128-B request,
      MI355X_MICROARCH.md "HBM"; re-checked for 8-byte-per-lane accesses with tools/probes/pmc_calib.hip) + WRITE_SIZE.
"""
import os
import csv, sys, json, collections

def load(paths):
    acc = collections.defaultdict(list)
    for path in paths:
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("ilqr::", "")
            if name.startswith("void "):
                name = name[5:]
            name = name.split("<")[0]          # template instantiations (k_line_search_s<false>) under their plain name
            acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    return acc

if len(sys.argv) > 2 and sys.argv[1] == "--traffic-json":
    rest = sys.argv[3:]
    stamp = None
    if rest and rest[0] == "--stamp":      # signature of the bench run the counters were collected on (bench.py --print-signature)
        stamp = json.loads(rest[1]); rest = rest[2:]
    acc = load(rest)
    out = {}
    for (k, c), v in acc.items():
        if c in ("FETCH_SIZE", "WRITE_SIZE") and k.startswith("k_"):
            out.setdefault(k, {})[c + "_KiB"] = max(v)
    out = {k: v for k, v in out.items() if len(v) == 2}
    if stamp is not None:
        import subprocess, datetime
        # (the GPU box has no .git: the build container passes its HEAD in, tools/collect_round.sh)
        stamp["head"] = os.environ.get("ILQR_GIT_HEAD", "")
        if not stamp["head"]:
            try:
                stamp["head"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
            except Exception:
                stamp["head"] = "unknown"
        stamp["date"] = datetime.date.today().isoformat()
        out["_stamp"] = stamp
    json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))
else:
    for (k, c), v in sorted(load(sys.argv[1:]).items()):
        print("%-40s %-22s n=%4d  mean=%.6g  max=%.6g" % (k, c, len(v), sum(v) / len(v), max(v)))
