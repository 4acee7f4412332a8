#!/usr/bin/env python3
"""Walk the knot loop of a kernel in a hipcc -save-temps ISA listing along the common (no LLT failure) path and count
instructions by class.  usage: isa_path.py file.s kernel_symbol_substring [--dump out.txt] [--take LABEL ...]
Rules: s_cbranch_execz falls through (some lane is active), s_branch is followed, any other conditional branch falls
through unless its target label is listed after --take; s_cbranch_execnz is taken."""
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
sym = sys.argv[2]
take = set()
loop_label = None
dump = None
a = sys.argv[3:]
while a:
    if a[0] == "--dump": dump = a[1]; a = a[2:]
    elif a[0] == "--take": take.add(a[1]); a = a[2:]
    elif a[0] == "--loop": loop_label = a[1]; a = a[2:]
    else: raise SystemExit("bad arg " + a[0])
start = next(i for i, l in enumerate(src) if l.startswith("_ZN") and sym in l and l.rstrip().endswith(sym) is False and ":" in l and sym in l.split(":")[0])
end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
labels = {}
for i in range(start, end):
    m = re.match(r"^(\.LBB\d+_\d+):", src[i])
    if m: labels[m.group(1)] = i
# loop header = the label that is the target of the last backward s_branch
back = [(i, re.search(r"(\.LBB\d+_\d+)", src[i]).group(1)) for i in range(start, end) if re.match(r"\s+s_c?branch", src[i]) and re.search(r"(\.LBB\d+_\d+)", src[i])]
back = [(i, t) for i, t in back if labels[t] < i]
hdr = max(back, key=lambda p: p[0] - labels[p[1]])
if loop_label: hdr = max([p for p in back if p[1] == loop_label], key=lambda p: p[0])
loop_start, loop_end = labels[hdr[1]], hdr[0]
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr_read"): return "acc_read"
    if op.startswith("v_accvgpr_write"): return "acc_write"
    if op.startswith("v_accvgpr_mov"): return "acc_mov"
    if op.startswith("v_mov"): return "v_mov"
    if op.startswith("v_cndmask"): return "v_cndmask"
    if op.startswith("v_cmp"): return "v_cmp"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"): return "v_lane"
    if re.match(r"v_(fma|add|mul|rsq|rcp|max|min|div|ldexp|frexp|trig|sqrt)_f64", op) or op.startswith("v_pk_"): return "v_f64"
    if op.startswith("v_"): return "v_other"
    if op.startswith("ds_bpermute") or op.startswith("ds_swizzle"): return "ds_perm"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "ds_write"
    if op.startswith("global_load") and "lds" in op: return "g_load_lds"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"): return "g_load"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store"): return "g_store"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_barrier"): return "s_barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "s_branch"
    if op.startswith("s_"): return "s_alu"
    return "other"
cnt = collections.Counter()
seg = collections.OrderedDict()
out = []
i = loop_start
nm = 0
steps = 0
seen = []
segcnt = collections.Counter()
while True:
    steps += 1
    if steps > 60000:
        for x in seen[:40]: print(x)
        raise SystemExit("runaway")
    l = src[i]
    s = l.strip()
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":") or re.match(r"^\.LBB", l):
        i += 1; continue
    op = s.split()[0]
    c = cls(op)
    cnt[c] += 1
    out.append("%6d %-10s %s" % (i + 1, c, s))
    if c == "s_branch":
        tgt = re.search(r"(\.LBB\d+_\d+)", s).group(1)
        if i == loop_end: break
        if op != "s_branch" and op != "s_cbranch_execz": seen.append((i + 1, op, tgt, tgt in take))
        if op == "s_branch" or op == "s_cbranch_execnz" or tgt in take:
            i = labels[tgt]; continue
    i += 1
for x in seen: print("  cond branch", x)
print("loop %s: lines %d..%d" % (hdr[1], loop_start + 1, loop_end + 1))
tot = sum(cnt.values())
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]): print("%-12s %5d" % (k, v))
print("%-12s %5d   non-MFMA %d" % ("total", tot, tot - cnt["mfma"]))
if dump: open(dump, "w").write("\n".join(out) + "\n")
