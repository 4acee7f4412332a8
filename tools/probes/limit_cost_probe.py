#!/usr/bin/env python3
"""What the joint-limit option costs when nothing is stopped: the contact workload of bench.py (B = 4096, N = 25, 10 fixed iterations), rows off /
on, for a `rocprofv3 --kernel-trace --stats` pass each:   LIM=0|1 python3 tools/probes/limit_cost_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge._load_package()
from mpc_ilqr_mujoco_amd import solver as sv
sc = pkg.scenario
B, N = int(os.environ.get("ILQR_B", "4096")), 25
prob = sc.make_problem(sv.reference_kinematics, N=N, gravity=((0.0, 0.0, -9.81) if int(os.environ.get("CM", "2")) else None))
ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
x0, ui = sc.synthetic_batch(B, N, 0, ug)
s = sv.BatchedILQR(B, N=N); s.set_problem(prob); s.set_contact_mode(int(os.environ.get("CM", "2"))); s.set_joint_limits(os.environ.get("LIM", "0") == "1")
s.set_options(early_exit=False); s.set_max_iterations(10)
s.initialize(x0, ui); s.solve(x0)
t0 = time.perf_counter()
for _ in range(2):
    s.initialize(x0, ui); s.solve(x0)
print("limits %s: %.1f ms per step" % (os.environ.get("LIM", "0"), 1e3 * (time.perf_counter() - t0) / 2))
s.close()
