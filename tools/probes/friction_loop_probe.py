"""Diagnostic: closed loop on the friction-limited stance row (contact mode 3) against mode 2, still and pushed.  python tools/probes/friction_loop_probe.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import __graft_entry__ as ge
pkg = ge._load_package()
from mpc_ilqr_mujoco_amd import mpc_loop as ml, references as rf, solver as sv
sc = pkg.scenario
B, N = 2, 25
base = sc.make_problem(sv.reference_kinematics, N=N, gravity=(0.0, 0.0, -9.81))
rd = rf.ReferenceData(sv.reference_kinematics, sv.reference_com_velocity)
rd.set_states(np.tile(sc.standing_state(), (60, 1))); rd.contact = np.ones((60, 2), dtype=np.int32)
ug = sv.gravity_compensation(sc.standing_state(), base["gravity"])
x_still = np.tile(sc.standing_state(), (B, 1)); u_still = np.tile(ug, (B, N, 1))
for mode, mu, push in ((2, 1.0, 0.0), (3, 1.0, 0.0), (3, 0.3, 0.0), (2, 1.0, 0.6), (3, 1.0, 0.6), (3, 0.3, 0.6), (3, 0.02, 0.6), (3, 0.02, 0.0), (4, 1.0, 0.0), (4, 1.0, 0.6), (4, 0.3, 0.6), (4, 0.02, 0.6)):
    s = sv.BatchedILQR(B, N=N); s.set_max_iterations(3); s.set_contact_mode(mode); s.set_friction(mu); s.set_options(jacobian_mode=1, fd_eps=1e-5)
    xp = x_still.copy(); xp[:, 27] += push
    xs, us = ml.MPCRunner(s, rd, base).run(xp, 6, u_init=u_still)
    ee0 = sv.reference_kinematics(xs[0, 0])[1]; ee1 = sv.reference_kinematics(xs[-1, 0])[1]
    print("mode %d mu %.2f push %.1f: finite %s  pelvis z min %.4f  feet move xy %.4f z %.4f  max |v| end %.3f  |u| max %.1f" % (
        mode, mu, push, bool(np.all(np.isfinite(xs))), xs[:, :, 2].min(), np.abs(ee1[:, :2] - ee0[:, :2]).max(), np.abs(ee1[:, 2] - ee0[:, 2]).max(), np.abs(xs[-1, :, 26:]).max(), np.abs(us).max()))
    s.close()
