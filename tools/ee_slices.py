"""Early-exit and fixed-iteration solve times against the number of batch slices (ILQR_SLICES): python tools/ee_slices.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge._load_package()
from mpc_ilqr_mujoco_amd import solver as sv
sc = pkg.scenario
B, N = int(os.environ.get("ILQR_B", "4096")), 25
prob = sc.make_problem(sv.reference_kinematics, N=N)
ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
x0, ui = sc.synthetic_batch(B, N, 0, ug)
s = sv.BatchedILQR(B, N=N); s.set_problem(prob)
for ee in (True, False):
    s.set_options(early_exit=ee)
    for k in (1, 2, 3, 4, 6, 8, 16):
        os.environ["ILQR_SLICES"] = str(k)
        ts = []
        for rep in range(4):
            s.initialize(x0, ui)
            t0 = time.perf_counter(); s.solve(x0); ts.append(time.perf_counter() - t0)
        it = s.iterations().sum()
        print("early_exit=%d slices=%2d  %.2f ms  %.0f executed it/s" % (ee, k, 1e3 * min(ts[1:]), it / min(ts[1:])), flush=True)
