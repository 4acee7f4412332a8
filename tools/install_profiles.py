#!/usr/bin/env python3
"""Copy a round's collection (gpurun_out/profiles_<tag>, written on the GPU box by tools/collect_round.sh) into profiles/ and refresh the
figures profiles/README.md quotes for that round in the checked forms (`kernel`: avg|max|alone X ms, headline **N it/s**, `roofline.frac`
**x**) from the CSVs and the kept default bench line -- so that tests/test_profiles_consistency.py holds by construction.
   python3 tools/install_profiles.py r05"""
import csv, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
src, dst = os.path.join(ROOT, "gpurun_out", "profiles_" + tag), os.path.join(ROOT, "profiles")
head = json.load(open(os.path.join(src, "traffic_latest.json")))["_stamp"].get("head", "unknown")
# the summary once more, here (same files, same tool): the GPU box may carry an older copy of the tool
with open(os.path.join(src, tag + "_summary.md"), "w") as f:
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "round_summary.py"), src, tag, head], stdout=f)
for name in os.listdir(src):
    p = os.path.join(src, name)
    if name.startswith(tag + "_") and os.path.isfile(p):
        shutil.copy(p, os.path.join(dst, name))
    elif name.startswith(tag + "_") and os.path.isdir(p):
        shutil.rmtree(os.path.join(dst, name), ignore_errors=True); shutil.copytree(p, os.path.join(dst, name))
shutil.copy(os.path.join(src, "traffic_latest.json"), os.path.join(dst, "traffic_latest.json"))
shutil.copy(os.path.join(src, "mfma_summary.txt"), os.path.join(dst, tag + "_mfma_backward_summary.txt"))


def stats(path):
    out = {}
    for r in csv.DictReader(open(path)):
        n = r["Name"].split("(")[0].replace("ilqr::", "").replace("void ", "").strip()
        out[n] = (float(r["AverageNs"]) / 1e6, float(r["MaxNs"]) / 1e6)
    return out


st, al = stats(os.path.join(dst, tag + "_kernel_stats.csv")), stats(os.path.join(dst, tag + "_alone_kernel_stats.csv"))
b = json.loads(open(os.path.join(dst, tag + "_bench_lines", "bench_default.json")).read().strip().split("\n")[-1])
readme = os.path.join(dst, "README.md")
text = open(readme).read()
start = text.index("## Round %d" % int(tag[1:]))
nxt = text.find("\n## Round ", start + 5)
sec = text[start:nxt if nxt > 0 else len(text)]


def requote(m):
    name, kind = m.group(1), m.group(2)
    tab = al if kind == "alone" else st
    if name not in tab:
        sys.exit("profiles/README.md quotes %s (%s), which the new collection does not have" % (name, kind))
    return "`%s`: %s %.3f ms" % (name, kind, tab[name][1 if kind == "max" else 0])


sec = re.sub(r"`(k_[^`]+)`: (avg|max|alone) [\d.]+ ms", requote, sec)
sec = re.sub(r"headline \*\*[\d ]+ it/s\*\*", "headline **%s it/s**" % format(int(round(b["value"])), ",").replace(",", " "), sec)
sec = re.sub(r"`roofline.frac` \*\*[\d.]+\*\*", "`roofline.frac` **%.3f**" % b["roofline"]["frac"], sec)
sec = re.sub(r"\([\d.]+ per full-batch first pass inside the bench", "(%.2f per full-batch first pass inside the bench" % b["roofline"]["avg_launch_ms"], sec)
open(readme, "w").write(text[:start] + sec + (text[nxt:] if nxt > 0 else ""))
print("installed %s from HEAD %s: headline %.0f it/s, roofline.frac %.3f" % (tag, head, b["value"], b["roofline"]["frac"]))
