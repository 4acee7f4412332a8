#!/usr/bin/env python3
"""Scratch traffic of a kernel's main loop from the compiler's assembly: spills stored before the loop (hoisted loop
invariants) and loads / stores inside it.   hipcc -S --cuda-device-only ... -o x.s ; python tools/spill_report.py x.s <mangled prefix>"""
import sys, collections
txt = open(sys.argv[1]).read().split("\n")
for name in sys.argv[2:]:
    s = [i for i, l in enumerate(txt) if l.startswith(name)][0]
    nxt = [i for i, l in enumerate(txt) if l.startswith("_ZN") and i > s]
    L = txt[s:(nxt[0] if nxt else len(txt))]
    hdr = [i for i, l in enumerate(L) if "Loop Header" in l or "This Inner Loop" in l]
    if not hdr:
        print(name, "no loop"); continue
    h = hdr[0]
    end = [i for i, l in enumerate(L) if "s_cbranch" in l and i > h][-1]
    cnt = lambda seg, key: sum(1 for l in seg if key in l)
    ins = lambda seg: sum(1 for l in seg if l.strip() and l.strip()[0] not in ";.")
    print("%s: before the loop %d instructions, %d scratch stores; loop %d instructions, %d scratch loads, %d scratch stores"
          % (name, ins(L[:h]), cnt(L[:h], "scratch_store"), ins(L[h:end]), cnt(L[h:end], "scratch_load"), cnt(L[h:end], "scratch_store")))
