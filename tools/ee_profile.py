import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as ge
pkg = ge._load_package()
from mpc_ilqr_mujoco_amd import solver as sv
sc = pkg.scenario
B, N = 4096, 25
prob = sc.make_problem(sv.reference_kinematics, N=N)
ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
x0, ui = sc.synthetic_batch(B, N, 0, ug)
s = sv.BatchedILQR(B, N=N); s.set_problem(prob); s.set_options(early_exit=True); s.enable_profiling(True)
for rep in range(2):
    s.initialize(x0, ui); s.solve(x0)
ms, n = s.stage_ms()
it = s.iterations()
print("iterations histogram", np.bincount(it, minlength=11))
print({k: round(v, 2) for k, v in ms.items()}, "sum", round(sum(ms.values()), 1))
tc, ta, tl = s.trace()
# active rollouts at the start of each iteration
print("active at iteration k:", [(it > k).sum() for k in range(10)])
