"""profiles/ is self-consistent by construction: the per-kernel table of the current round is written by
tools/round_summary.py from the CSVs beside it, and every kernel figure profiles/README.md quotes for the round is checked here
against profiles/rNN_kernel_stats.csv (3 %).  CPU test: it only parses committed files."""
import csv
import importlib.util
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
TAG = "r06"


def _summary_mod():
    spec = importlib.util.spec_from_file_location("round_summary", os.path.join(ROOT, "tools", "round_summary.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def _have():
    return os.path.exists(os.path.join(PROF, TAG + "_kernel_stats.csv")) and os.path.exists(os.path.join(PROF, TAG + "_summary.md"))


@pytest.mark.skipif(not _have(), reason="profiles of the current round not collected yet (tools/collect_round.sh)")
def test_summary_table_is_what_the_csvs_say():
    m = _summary_mod()
    rows = {n: (calls, avg, mx, alone) for n, calls, avg, mx, alone in m.kernel_rows(PROF, TAG)}
    assert rows, "no kernel rows"
    seen = 0
    for line in open(os.path.join(PROF, TAG + "_summary.md")):
        mm = re.match(r"^\| `(k_[^`]+)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]*) \|", line)
        if not mm:
            continue
        n = mm.group(1)
        assert n in rows, n
        calls, avg, mx, alone = rows[n]
        assert int(mm.group(2)) == calls
        assert abs(float(mm.group(3)) - avg) <= 0.03 * avg + 5e-4 and abs(float(mm.group(4)) - mx) <= 0.03 * mx + 5e-4, (n, line)
        if alone is not None:
            assert abs(float(mm.group(5)) - alone) <= 0.03 * alone + 5e-4, (n, line)
        seen += 1
    assert seen == len(rows)
    # the traffic stamp carries the HEAD the collection ran on (passed in from the build container)
    t = json.load(open(os.path.join(PROF, "traffic_latest.json")))
    assert t["_stamp"].get("head") not in (None, "", "unknown")


@pytest.mark.skipif(not _have(), reason="profiles of the current round not collected yet (tools/collect_round.sh)")
def test_kernel_figures_quoted_in_the_readme_for_this_round_match_the_kept_csvs():
    """README convention for the current round: a kernel figure is written  `k_name`: avg X ms  /  `k_name`: max X ms  /
    `k_name`: alone X ms  and nothing else counts as a quote of the CSVs (the prose cites rNN_summary.md for everything else)."""
    text = open(os.path.join(PROF, "README.md")).read()
    start = text.index("## Round 6")
    nxt = text.find("\n## Round ", start + 5)
    sec = text[start:nxt if nxt > 0 else len(text)]
    m = _summary_mod()
    st = m.read_stats(os.path.join(PROF, TAG + "_kernel_stats.csv"))
    al = m.read_stats(os.path.join(PROF, TAG + "_alone_kernel_stats.csv"))
    quotes = re.findall(r"`(k_[^`]+)`: (avg|max|alone) ([\d.]+) ms", sec)
    assert quotes, "the round's README section quotes no kernel figure in the checked form"
    for name, kind, val in quotes:
        src = al if kind == "alone" else st
        assert name in src, (name, kind)
        want = src[name]["avg" if kind in ("avg", "alone") else "max"]
        assert abs(float(val) - want) <= 0.03 * want, (name, kind, val, want)
    # headline figures quoted as  **N it/s**  in the section must match the kept default bench line
    bl = os.path.join(PROF, TAG + "_bench_lines", "bench_default.json")
    b = json.loads(open(bl).read().strip().split("\n")[-1])
    hm = re.search(r"headline \*\*([\d ]+) it/s\*\*", sec)
    assert hm and abs(float(hm.group(1).replace(" ", "")) - b["value"]) <= 0.03 * b["value"]
    fm = re.search(r"`roofline.frac` \*\*([\d.]+)\*\*", sec)
    assert fm and abs(float(fm.group(1)) - b["roofline"]["frac"]) <= 0.03 * b["roofline"]["frac"]
