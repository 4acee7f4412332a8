#!/usr/bin/env python3
"""Analytic Jacobians of the sliding-foot contact modes (3 / 4) against the oracle's forward-mode AD on the committed friction states."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge._load_package()
from mpc_ilqr_mujoco_amd import solver as sv
import oracle_lib as ol
sc = pkg.scenario
g = np.load(os.path.join(ROOT, "tests", "golden", "friction_golden.npz"))
n, N = len(g["x"]), 25
prob = sc.make_problem(sv.reference_kinematics, N=N, cfg=dict(sc.SHIPPED_CONFIG), stance=np.ones((N + 1, 2), dtype=np.int32), gravity=list(g["gravity"]))
ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
x0, ui = sc.synthetic_batch(n, N, 28, ug)
modes = [int(a) for a in sys.argv[1:]] or [2, 3, 4]
for mode in modes:
    for mu in sorted(set(float(m) for m in g["mu"])):
        idx = [i for i in range(n) if float(g["mu"][i]) == mu]
        s = sv.BatchedILQR(n, N=N, dt=prob["dt"]); s.set_problem(prob); s.set_contact_mode(mode, float(g["soft"])); s.set_friction(mu); s.set_options(jacobian_mode=0)
        s.initialize(x0, ui)
        X = np.repeat(g["x"][:, None, :], N + 1, axis=1); U = np.repeat(g["u"][:, None, :], N, axis=1)
        s.set_trajectory(X, U); s.stage_linearize()
        A, Bm = s.linearization()
        for i in idx:
            o = ol.Oracle(N, prob["dt"]); o.set_problem(prob, 0); o.set_options(jac_mode=0); o.set_contact_mode(mode, float(g["soft"])); o.set_friction(mu)
            o.set_trajectory(X[i], U[i]); o.linearize()
            Ao, Bo = o.get("A")[0], o.get("B")[0]
            ea, eb = np.abs(A[i][0] - Ao).max() / max(1.0, np.abs(Ao).max()), np.abs(Bm[i][0] - Bo).max() / max(1.0, np.abs(Bo).max())
            ra, ca = np.unravel_index(np.abs(A[i][0] - Ao).argmax(), Ao.shape)
            print("mode %d mu %.1f state %d slide %s act %s: A err %.2e (row %d col %d)  B err %.2e" % (mode, mu, i, g["slide"][i], g["act"][i], ea, ra, ca, eb), flush=True)
        s.close()
