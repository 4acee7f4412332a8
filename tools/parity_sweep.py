#!/usr/bin/env python3
"""Wider GPU <-> oracle parity sweep than the test-suite (one-off robustness check, needs the GPU and the oracle):
   python tools/parity_sweep.py [rollouts_per_case]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge._load_package()
from mpc_ilqr_mujoco_amd import solver as sv
import oracle_lib as ol
sc = pkg.scenario
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N = 25
worst = dict(cost=0.0, K=0.0, trace=0.0)
bad = 0
for case, (gravity, walking, seed) in enumerate([(None, False, 100), ((0, 0, -2.0), True, 200), ((0, 0, -0.5), False, 300)]):
    stance = None
    if walking:
        stance = np.ones((N + 1, 2), dtype=np.int32); stance[3:9, 0] = 0; stance[12:20, 1] = 0; stance[N, :] = 0
    prob = sc.make_problem(sv.reference_kinematics, N=N, stance=stance, gravity=gravity)
    ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
    x0, ui = sc.synthetic_batch(B, N, seed, ug)
    s = sv.BatchedILQR(B, N=N); s.set_problem(prob); s.set_max_iterations(10)
    s.initialize(x0, ui)
    cost = s.solve(x0)
    tc, ta, tl = s.trace(); it = s.iterations(); K = s.gains_K()
    t0 = time.time()
    for b in range(B):
        o = ol.Oracle(N, prob["dt"]); o.set_problem(prob); o.set_options(max_iter=10)
        o.initialize(x0[b], ui[b]); ok, c = o.solve(x0[b])
        n, oc, oa, ol_ = o.trace()
        same_iters = (n == it[b]); same_alpha = np.array_equal(ta[b][:n], oa[:n])
        rc = abs(c - cost[b]) / abs(c)
        rk = np.abs(K[b] - o.get("K")).max() / np.abs(o.get("K")).max()
        rt = np.nanmax(np.abs(tc[b][:n + 1] - oc[:n + 1]) / np.abs(oc[:n + 1]))
        worst["cost"] = max(worst["cost"], rc); worst["K"] = max(worst["K"], rk); worst["trace"] = max(worst["trace"], rt)
        if not (same_iters and same_alpha and rc < 1e-5 and rk < 1e-5):
            bad += 1
            print("MISMATCH case %d rollout %d: iters %d/%d alpha_equal %s rel cost %.2e rel K %.2e" % (case, b, it[b], n, same_alpha, rc, rk))
    print("case %d (gravity %s, walking %s): %d rollouts checked in %.0f s" % (case, prob["gravity"], walking, B, time.time() - t0), flush=True)
    s.close()
print("worst relative differences:", worst, " mismatches:", bad)
