"""Where does the nominal re-rollout of an accepted line-search candidate differ from the candidate?  (bitwise; stage API)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge._load_package()
from mpc_ilqr_mujoco_amd import solver as sv
sc = pkg.scenario
B, N = 64, 25
prob = sc.make_problem(sv.reference_kinematics, N=N)
ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
x0, ui = sc.synthetic_batch(B, N, 31, ug)
s = sv.BatchedILQR(B, N=N); s.set_problem(prob)
s.initialize(x0, ui)
s.stage_linearize(); s.stage_cost_quadratics(); s.stage_backward_pass(); s.stage_line_search()
xa = s.xbar().copy()
s.stage_rollout()
xb = s.xbar().copy()
d = xa.view(np.int64) != xb.view(np.int64)
print("differing elements:", d.sum(), "of", d.size, " max |diff|", np.abs(xa - xb).max())
print("by state index:", {int(i): int(c) for i, c in enumerate(d.sum(axis=(0, 1))) if c})
print("first knot with a difference per rollout:", [int(np.argmax(d[b].any(axis=1))) if d[b].any() else -1 for b in range(min(B, 16))])
b = int(np.argmax(d.any(axis=(1, 2))))
t = int(np.argmax(d[b].any(axis=1)))
print("rollout", b, "knot", t, "indices", np.nonzero(d[b, t])[0], "ulps", (xa[b, t].view(np.int64) - xb[b, t].view(np.int64))[d[b, t]])
