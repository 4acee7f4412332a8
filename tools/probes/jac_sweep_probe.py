#!/usr/bin/env python3
"""Random sweep of the analytic Jacobians of the constrained step (sliding feet, joint-limit rows) against the oracle's forward-mode AD: every
knot of every rollout a different random state (hinges past their ranges, lateral pushes that make feet slide)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge._load_package()
from mpc_ilqr_mujoco_amd import solver as sv
import oracle_lib as ol
sc = pkg.scenario
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 25
rng = np.random.default_rng(11)
jr = ol.joint_ranges()
prob = sc.make_problem(sv.reference_kinematics, N=N, cfg=dict(sc.SHIPPED_CONFIG), stance=np.ones((N + 1, 2), dtype=np.int32), gravity=[0.0, 0.0, -9.81])
ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
x0, ui = sc.synthetic_batch(B, N, 3, ug)
n = B * (N + 1)
X = np.tile(sc.standing_state(), (n, 1))
X[:, 0:3] += rng.uniform(-0.02, 0.02, (n, 3))
X[:, 7:26] = rng.uniform(-0.15, 0.15, (n, 19)); X[:, 26:] = rng.uniform(-0.4, 0.4, (n, 25))
X[::2, 26:29] += rng.uniform(-0.6, 0.6, (n // 2 + n % 2, 3))[: len(X[::2])]
aa = rng.uniform(-0.05, 0.05, (n, 3)); ang = np.linalg.norm(aa, axis=1)
X[:, 3] = np.cos(ang / 2); X[:, 4:7] = (np.sin(ang / 2) / ang)[:, None] * aa
for i in range(n):
    for j in rng.choice(19, size=int(rng.integers(0, 3)), replace=False):
        up = rng.random() < 0.5
        X[i, 7 + j] = jr[j, 1] + rng.uniform(0.01, 0.08) if up else jr[j, 0] - rng.uniform(0.01, 0.08)
        X[i, 32 + j] = rng.uniform(0.3, 2.0) * (1 if up else -1) * (1 if rng.random() < 0.75 else -1)
X = X.reshape(B, N + 1, 51); U = rng.uniform(-20, 20, (B, N, 19))
for cm, lim, mu in ((2, True, 1.0), (3, False, 0.3), (4, False, 0.3), (3, True, 0.3), (4, True, 0.6), (0, True, 1.0)):
    s = sv.BatchedILQR(B, N=N, dt=prob["dt"]); s.set_problem(prob); s.set_contact_mode(cm); s.set_friction(mu); s.set_joint_limits(lim); s.set_options(jacobian_mode=0)
    s.initialize(x0, ui); s.set_trajectory(X, U); s.stage_linearize()
    A, Bm = s.linearization()
    worst, bad = 0.0, 0
    for b in range(B):
        o = ol.Oracle(N, prob["dt"]); o.set_problem(prob, 0); o.set_options(jac_mode=0); o.set_contact_mode(cm); o.set_friction(mu); o.set_joint_limits(lim)
        o.set_trajectory(X[b], U[b]); o.linearize()
        Ao, Bo = o.get("A"), o.get("B")
        for t in range(N):
            e = max(np.abs(A[b][t] - Ao[t]).max() / max(1.0, np.abs(Ao[t]).max()), np.abs(Bm[b][t] - Bo[t]).max() / max(1.0, np.abs(Bo[t]).max()))
            if e > 1e-8: bad += 1
            else: worst = max(worst, e)
    print("contact %d limits %d mu %.1f: %d knots, worst %.1e, knots off by more than 1e-8: %d" % (cm, lim, mu, B * N, worst, bad), flush=True)
    s.close()
