#!/usr/bin/env python3
"""The step with joint-limit rows on the two-lane kernels against the committed dense-KKT vectors and the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge._load_package()
from mpc_ilqr_mujoco_amd import solver as sv
import oracle_lib as ol
sc = pkg.scenario
g = np.load(os.path.join(ROOT, "tests", "golden", "joint_limit_golden.npz"))
n, N = len(g["x"]), 5
prob = sc.make_problem(sv.reference_kinematics, N=N, gravity=list(g["gravity"]))
s = sv.BatchedILQR(2, N=N, dt=float(g["h"])); s.set_problem(prob)
o = ol.Oracle(N, float(g["h"])); o.set_problem(prob)
for i in range(n):
    cm = int(g["contact"][i])
    s.set_contact_mode(cm, float(g["soft"])); s.set_joint_limits(True)
    xs = np.tile(g["x"][i], (2, 1)); us = np.tile(g["u"][i], (2, 1))
    got = s.step_stance(xs, us, int(g["stance"][i][0]), int(g["stance"][i][1]))
    s.set_joint_limits(False)
    got0 = s.step_stance(xs, us, int(g["stance"][i][0]), int(g["stance"][i][1]))
    o.set_contact_mode(cm, float(g["soft"])); o.set_joint_limits(True)
    want = o.step_stance(g["x"][i], g["u"][i], g["stance"][i])
    print("case %d contact %d lock %s: vs golden %.2e  vs oracle %.2e  unlimited vs golden %.2e  lanes alike %s" % (
        i, cm, np.flatnonzero(g["lock"][i]).tolist(), np.abs(got[0] - g["x_next"][i]).max(), np.abs(got[0] - want).max(),
        np.abs(got0[0] - g["x_next_unlimited"][i]).max(), np.array_equal(got[0], got[1])), flush=True)
s.close()
