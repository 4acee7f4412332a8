import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import __graft_entry__ as ge
pkg = ge._load_package()
from mpc_ilqr_mujoco_amd import solver as sv
sc = pkg.scenario
B, N = 4096, 25
prob = sc.make_problem(sv.reference_kinematics, N=N)
ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
x0, ui = sc.synthetic_batch(B, N, 0, ug)
s = sv.BatchedILQR(B, N=N); s.set_problem(prob); s.set_options(early_exit=False)
for prof in (False, True, False, True):
    s.enable_profiling(prof)
    ts = []
    for rep in range(4):
        s.initialize(x0, ui)
        t0 = time.perf_counter(); s.solve(x0); ts.append(time.perf_counter() - t0)
    print("profiling", prof, "solve %.2f ms" % (1e3 * min(ts[1:])))
