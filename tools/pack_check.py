#!/usr/bin/env python3
"""Operand-layout Riccati kernel (riccati_pack.hip) against the generic and the folded one-wave kernels on the same stage data:
   python tools/pack_check.py [B] [N]      (ILQR_CONTACT=1/2 for the stance-constrained Jacobians)"""
import os, sys
os.environ.setdefault("ILQR_ENV_PER_CALL", "1")   # these tools switch kernel families around calls on one handle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge._load_package()
from mpc_ilqr_mujoco_amd import solver as sv
sc = pkg.scenario
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 25
CONTACT = int(os.environ.get("ILQR_CONTACT", "0"))
prob = sc.make_problem(sv.reference_kinematics, N=N, gravity=(0.0, 0.0, -9.81)) if CONTACT else sc.make_problem(sv.reference_kinematics, N=N)
ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
x0, ui = sc.synthetic_batch(B, N, 0, ug)
s = sv.BatchedILQR(B, N=N, lib_path=sv.LEGACY_LIB_PATH)      # (wave-fold: a cross-check family of the test library)
s.set_problem(prob)
if CONTACT:
    s.set_contact_mode(CONTACT)
s.initialize(x0, ui)
s.stage_linearize(); s.stage_cost_quadratics()
out = {}
for kind in ("wave-generic", "wave-fold", "wave"):
    os.environ["ILQR_BACKWARD"] = kind
    s.stage_backward_pass()
    Vx, Vxx = s.value_function()
    out[kind] = (s.gains_K().copy(), s.gains_kff().copy(), Vx.copy(), Vxx.copy())
A, Bm = s.linearization()
lx, lu, lxx, luu = s.quadratics()
os.environ["ILQR_BACKWARD"] = "wave-generic"
s.stage_linearize(); s.stage_cost_quadratics()
A2, B2 = s.linearization(); q2 = s.quadratics()
print("layout round trip: A %.1e B %.1e lxx %.1e lx %.1e" % (np.abs(A - A2).max(), np.abs(Bm - B2).max(), np.abs(lxx - q2[2]).max(), np.abs(lx - q2[0]).max()))
ref = out["wave-generic"]
ok = True
for kind in ("wave-fold", "wave"):
    d = [np.abs(a - b).max() / max(1e-300, np.abs(b).max()) for a, b in zip(out[kind], ref)]
    print("%-10s vs generic: K %.2e  k %.2e  Vx %.2e  Vxx %.2e" % ((kind,) + tuple(d)))
    ok = ok and max(d) < 1e-9
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
