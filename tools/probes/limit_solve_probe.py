#!/usr/bin/env python3
"""Analytic-Jacobian solve with joint-limit rows against the oracle, per kernel family (debugging aid)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["ILQR_ENV_PER_CALL"] = "1"
import __graft_entry__ as ge
pkg = ge._load_package()
from mpc_ilqr_mujoco_amd import solver as sv
import oracle_lib as ol
import test_gpu_parity as tp
cm = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
Bs = 3
prob, x0, ui = tp.make(Bs, seed=31, gravity=[0.0, 0.0, -9.81] if cm else None, walking=True)
x0 = x0.copy(); x0[:, 7 + 3] = 2.09; x0[:, 32 + 3] = 1.5; x0[1:, 7 + 18] = -1.28; x0[1:, 32 + 18] = -2.0
ref = []
for b in range(Bs):
    ob = tp.oracle_for(prob, jac_mode=0, early_exit=0, max_iter=iters); ob.set_contact_mode(cm); ob.set_joint_limits(True)
    ob.initialize(x0[b], ui[b]); ok, c = ob.solve(x0[b]); ref.append((ob.trace(), ob.get("K"), ob.get("xbar")))
for fam in ("", "wave-generic"):
    if fam: os.environ["ILQR_BACKWARD"] = fam
    else: os.environ.pop("ILQR_BACKWARD", None)
    s = sv.BatchedILQR(Bs, N=25, dt=prob["dt"]); s.set_problem(prob); s.set_contact_mode(cm); s.set_joint_limits(True)
    s.set_options(jacobian_mode=0, early_exit=False); s.set_max_iterations(iters)
    s.initialize(x0, ui); cost = s.solve(x0); tc, ta, tl = s.trace()
    for b in range(Bs):
        (nn, oc, oa, ol_), K, xb = ref[b]
        print("family '%s' b %d: cost trace gpu %s oracle %s alphas gpu %s oracle %s  rel K %.2e" % (fam, b, np.round(tc[b], 3), np.round(oc, 3), ta[b], oa, tp.rel(s.gains_K()[b], K)), flush=True)
    s.close()
