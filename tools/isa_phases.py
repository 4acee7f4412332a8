#!/usr/bin/env python3
"""Instruction classes per phase of a path dump written by tools/isa_path.py --dump: phases are cut at cumulative MFMA counts.
usage: isa_phases.py dump.txt name:count name:count ..."""
import sys, collections
L = [l for l in open(sys.argv[1]).read().split('\n') if l.strip()]
bounds = [(0, 'top')] + [(int(a.split(':')[1]), a.split(':')[0]) for a in sys.argv[2:]] + [(10**9, 'tail')]
seg = collections.OrderedDict((b[1], collections.Counter()) for b in bounds)
nm = 0; bi = 0
for l in L:
    c = l.split()[1]
    if c == 'mfma':
        nm += 1; continue
    while bi + 1 < len(bounds) and nm >= bounds[bi + 1][0]: bi += 1
    seg[bounds[bi][1]][c] += 1
cats = ['acc_read', 'acc_write', 'acc_mov', 'v_mov', 'v_cndmask', 'v_cmp', 'v_other', 'v_f64', 'v_lane', 'ds_read', 'ds_write', 'ds_perm', 'g_load', 'g_load_lds', 'g_store', 's_alu', 's_branch', 's_nop', 's_waitcnt']
print('%-14s' % 'after mfma#' + ''.join('%7s' % c[:6] for c in cats) + '  total')
for k, v in seg.items():
    print('%-14s' % k + ''.join('%7d' % v[c] for c in cats) + '  %5d' % sum(v.values()))
