"""Is a solve reproducible bit for bit from run to run (same inputs, same process)?  python tools/determinism.py [B] [iters] [early_exit]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge._load_package()
from mpc_ilqr_mujoco_amd import solver as sv
sc = pkg.scenario
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ee = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
N = 25
prob = sc.make_problem(sv.reference_kinematics, N=N)
ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
x0, ui = sc.synthetic_batch(B, N, 3, ug)
s = sv.BatchedILQR(B, N=N); s.set_problem(prob); s.set_max_iterations(iters); s.set_options(early_exit=ee)
ref = None
for rep in range(8):
    s.enable_profiling(rep % 2 == 1)
    s.set_regularization(1e-6)          # lambda is solver state that survives a solve (ilqr.cpp:16, as in the reference): reset it
    s.initialize(x0, ui); c = s.solve(x0)
    out = (c.copy(), s.gains_K().copy(), s.xbar().copy(), s.trace()[0].copy())
    if ref is None:
        ref = out
    else:
        d = [int((a.view(np.int64) != b.view(np.int64)).sum()) for a, b in zip(out, ref)]
        print("run", rep, "differing elements (cost, K, xbar, trace):", d, " max rel cost diff %.2e" % np.max(np.abs(c - ref[0]) / np.abs(ref[0])), "mismatch", s.adopt_mismatches())
