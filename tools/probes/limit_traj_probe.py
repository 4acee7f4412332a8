#!/usr/bin/env python3
"""Per-knot analytic Jacobians with joint-limit rows along a nominal trajectory, against the oracle's AD (debugging aid)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge._load_package()
from mpc_ilqr_mujoco_amd import solver as sv
import oracle_lib as ol
import test_gpu_parity as tp
cm = int(sys.argv[1]) if len(sys.argv) > 1 else 2
Bs = 3
prob, x0, ui = tp.make(Bs, seed=31, gravity=[0.0, 0.0, -9.81] if cm else None, walking=True)
x0 = x0.copy(); x0[:, 7 + 3] = 2.09; x0[:, 32 + 3] = 1.5; x0[1:, 7 + 18] = -1.28; x0[1:, 32 + 18] = -2.0
s = sv.BatchedILQR(Bs, N=25, dt=prob["dt"]); s.set_problem(prob); s.set_contact_mode(cm); s.set_joint_limits(True); s.set_options(jacobian_mode=0)
if len(sys.argv) > 2:      # a solve first: what does it leave behind?
    s.set_options(jacobian_mode=0, early_exit=False); s.set_max_iterations(int(sys.argv[2])); s.initialize(x0, ui); s.solve(x0)
s.initialize(x0, ui); s.stage_linearize()
A, Bm = s.linearization(); xb, ub = s.xbar(), s.ubar()
jr = ol.joint_ranges()
for b in range(Bs):
    o = tp.oracle_for(prob, jac_mode=0); o.set_contact_mode(cm); o.set_joint_limits(True)
    o.set_trajectory(xb[b], ub[b]); o.linearize()
    Ao, Bo = o.get("A"), o.get("B")
    for t in range(25):
        ea = np.abs(A[b][t] - Ao[t]).max(); eb = np.abs(Bm[b][t] - Bo[t]).max()
        out = [j for j in range(19) if xb[b][t][7 + j] > jr[j, 1] or xb[b][t][7 + j] < jr[j, 0]]
        stopped = [j for j in out if abs(xb[b][t + 1][32 + j]) < 1e-9]
        if ea > 1e-8 or eb > 1e-8:
            r, c = np.unravel_index(np.abs(A[b][t] - Ao[t]).argmax(), Ao[t].shape)
            print("b %d t %2d stance %s out-of-range %s stopped %s: A err %.2e (row %d col %d) B err %.2e" % (b, t, np.asarray(prob["stance"]).reshape(-1, 26, 2)[0][t].tolist(), out, stopped, ea, r, c, eb), flush=True)
s.close()
