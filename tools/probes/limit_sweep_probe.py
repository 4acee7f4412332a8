#!/usr/bin/env python3
"""Random sweep of the step with joint-limit rows, GPU against the oracle: states with hinges past their ranges (moving out / in), every contact
mode, both stance patterns.  Decisions taken within 1e-6 rad/s of zero (the unlimited step's next rate of a violated hinge) are set aside."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge._load_package()
from mpc_ilqr_mujoco_amd import solver as sv
import oracle_lib as ol
sc = pkg.scenario
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
rng = np.random.default_rng(7)
jr = ol.joint_ranges()
prob = sc.make_problem(sv.reference_kinematics, N=5, gravity=[0.0, 0.0, -9.81])
X = np.tile(sc.standing_state(), (n, 1)); U = rng.uniform(-30, 30, (n, 19))
X[:, 7:26] = rng.uniform(-0.2, 0.2, (n, 19)); X[:, 26:] = rng.uniform(-0.5, 0.5, (n, 25))
aa = rng.uniform(-0.08, 0.08, (n, 3)); ang = np.linalg.norm(aa, axis=1)
X[:, 3] = np.cos(ang / 2); X[:, 4:7] = (np.sin(ang / 2) / ang)[:, None] * aa
for i in range(n):
    for j in rng.choice(19, size=int(rng.integers(0, 5)), replace=False):
        up = rng.random() < 0.5
        X[i, 7 + j] = jr[j, 1] + rng.uniform(0.0, 0.1) if up else jr[j, 0] - rng.uniform(0.0, 0.1)
        X[i, 32 + j] = rng.uniform(-2.0, 2.0)
s = sv.BatchedILQR(n, N=5, dt=prob["dt"]); s.set_problem(prob)
o = ol.Oracle(5, prob["dt"]); o.set_problem(prob)
worst = {}
for cm in (0, 1, 2, 3, 4):
    for st in ((1, 1), (1, 0), (0, 1)):
        s.set_contact_mode(cm); s.set_friction(0.4); s.set_joint_limits(True)
        got = s.step_stance(X, U, st[0], st[1])
        o.set_contact_mode(cm); o.set_friction(0.4)
        bad, skipped, stopped, w = 0, 0, 0, 0.0
        for i in range(n):
            o.set_joint_limits(True); want = o.step_stance(X[i], U[i], list(st))
            e = np.abs(got[i] - want).max() / max(1.0, np.abs(want).max())
            if e > 1e-8:
                o.set_joint_limits(False); free = o.step_stance(X[i], U[i], list(st))
                viol = [j for j in range(19) if X[i, 7 + j] > jr[j, 1] or X[i, 7 + j] < jr[j, 0]]
                if any(abs(free[32 + j]) < 1e-6 for j in viol): skipped += 1; continue
                bad += 1
            else:
                w = max(w, e)
            stopped += int(np.any(np.abs(want[32:]) < 1e-13))
        print("contact %d stance %s: worst %.1e, mismatches %d, set aside %d, states with a stopped hinge %d / %d" % (cm, st, w, bad, skipped, stopped, n), flush=True)
s.close()
