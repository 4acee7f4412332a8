"""Diagnostic: does the nominal rollout kernel reproduce the trajectory the line-search kernel accepted, bit for bit?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge._load_package()
from mpc_ilqr_mujoco_amd import solver as sv
sc = pkg.scenario
B, N = 4, 25
prob = sc.make_problem(sv.reference_kinematics, N=N)
ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
x0, ui = sc.synthetic_batch(B, N, 31, ug)
s = sv.BatchedILQR(B, N=N); s.set_problem(prob)
s.initialize(x0, ui)
s.stage_linearize(); s.stage_cost_quadratics(); s.stage_backward_pass()
imp, cost, alpha = s.stage_line_search()
xa, ua = s.xbar(), s.ubar()
s.stage_rollout()
xr = s.xbar()
d = xa != xr
print("accepted", imp, alpha, "mismatching elements", int(d.sum()))
for b in range(B):
    idx = np.argwhere(d[b])
    if len(idx):
        print(" rollout", b, "first knots", sorted(set(idx[:, 0].tolist()))[:6], "coords at first knot", idx[idx[:, 0] == idx[:, 0].min()][:, 1].tolist(), "max rel", np.abs(xa[b] - xr[b]).max())
import oracle_lib as ol
o = ol.Oracle(N, prob["dt"]); o.set_problem(prob)
for b in range(B):
    idx = np.argwhere(d[b])
    if not len(idx):
        continue
    t = int(idx[:, 0].min()); cs = idx[idx[:, 0] == t][:, 1].tolist()
    # identical inputs at knot t - 1?
    print("rollout", b, "knot", t, "inputs equal", np.array_equal(xa[b, t - 1], xr[b, t - 1]), "u arms", ua[b, t - 1, 11:])
    want = o.step(xa[b, t - 1], ua[b, t - 1])
    for c in cs:
        print("   coord", c, "LS %.17g  rollout %.17g  oracle %.17g" % (xa[b, t, c], xr[b, t, c], want[c]))
