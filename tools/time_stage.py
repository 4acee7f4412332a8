#!/usr/bin/env python3
"""Time one stage of the HIP path at the bench size (B=4096, N=25): python tools/time_stage.py backward [reps]"""
import os, sys, time
os.environ.setdefault("ILQR_ENV_PER_CALL", "1")   # this tool may switch kernel families around calls on one handle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge._load_package()
from mpc_ilqr_mujoco_amd import solver as sv
sc = pkg.scenario
stage = sys.argv[1] if len(sys.argv) > 1 else "backward"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B, N = int(os.environ.get("ILQR_B", "4096")), int(os.environ.get("ILQR_N", "25"))
CONTACT = int(os.environ.get("ILQR_CONTACT", "0"))      # 1 / 2: contact row f4 under physical gravity
prob = sc.make_problem(sv.reference_kinematics, N=N, gravity=(0.0, 0.0, -9.81)) if CONTACT else sc.make_problem(sv.reference_kinematics, N=N)
ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
x0, ui = sc.synthetic_batch(B, N, 0, ug)
s = sv.BatchedILQR(B, N=N)
s.set_problem(prob)
if CONTACT:
    s.set_contact_mode(CONTACT)
if os.environ.get("ILQR_LIMITS"):        # joint-limit rows of the plant (the constraint-free plant then runs its CONTACT 5 kernels)
    s.set_joint_limits(True)
if os.environ.get("ILQR_LAMBDA"):
    s.set_regularization(float(os.environ["ILQR_LAMBDA"]))
s.initialize(x0, ui)
if os.environ.get("ILQR_BENIGN"):   # every knot = the standing state with gravity-compensation controls
    s.set_trajectory(np.tile(sc.standing_state(), (B, N + 1, 1)), np.tile(ug, (B, N, 1)))
s.stage_linearize(); s.stage_cost_quadratics(); s.stage_backward_pass()
fn = {"backward": s.stage_backward_pass, "linearize": s.stage_linearize, "quadratics": s.stage_cost_quadratics,
      "rollout": s.stage_rollout, "line_search": s.stage_line_search}[stage]
fn()
t0 = time.perf_counter()
for _ in range(reps):
    fn()
dt = (time.perf_counter() - t0) / reps
print("%s: %.3f ms per call (host-timed, includes sync)  K checksum %.9e" % (stage, dt * 1e3, float(np.abs(s.gains_K()[::97]).sum())))
if os.environ.get("ILQR_QSTAMPS"):
    names = ["record -> LDS", "jacobian rows (both waves)", "gradient, balance rows || til x z, P'", "hessian: operand fetch", "hessian: second-order patch", "hessian: MFMA + store"]
    st = s.cost()[:6]
    for nme, v in zip(names, st):
        print("  %-44s %10.0f cycles  %5.1f %%" % (nme, v, 100 * v / st.sum()))
    print("  total %.0f cycles" % st.sum())
if os.environ.get("ILQR_SSTAMPS"):
    names = ["loads + dx exchange", "K dx + reduce", "cost (+CoM; one-lane kernel only)", "dynamics step", "store x"]
    st = s.cost()[:5]
    for nme, v in zip(names, st):
        print("  %-28s %10.0f cycles/step  %5.1f %%" % (nme, v / 25, 100 * v / st.sum()))
    print("  total %.0f cycles per step" % (st.sum() / 25))
if os.environ.get("ILQR_LSTAMPS") and CONTACT:
    names = ["load dump", "Minv + wrench sweeps || constraint rhs", "constraint solve", "correction, forces, prologue", "tangent sweeps + pelvis",
             "multiplier tangents", "apply Minv + G (MFMA)", "columns + store"]
    st = s.cost()[:8]
    for nme, v in zip(names, st):
        print("  %-44s %10.0f cycles  %5.1f %%" % (nme, v, 100 * v / st.sum()))
    print("  total %.0f cycles" % st.sum())
elif os.environ.get("ILQR_LSTAMPS") and os.environ.get("ILQR_LINT") != "1":
    names = ["load dump (two knots)", "accumulate forces (wave 0, 2 x 20 lanes)", "tangent sweeps (64 lanes) || Minv outward", "pelvis rows", "apply Minv (MFMA)", "columns + store"]
    st = s.cost()[:6]
    for nme, v in zip(names, st):
        print("  %-44s %10.0f cycles  %5.1f %%" % (nme, v, 100 * v / st.sum()))
    print("  total %.0f cycles per PAIR of knots" % st.sum())
elif os.environ.get("ILQR_LSTAMPS"):
    names = ["load dump", "accumulate forces (level-synchronous)", "prologue (one lane)", "Minv sweeps (25 lanes)", "tangent RNEA (47 lanes)", "apply Minv (MFMA)", "columns + store"]
    st = s.cost()[:7]
    for nme, v in zip(names, st):
        print("  %-28s %10.0f cycles  %5.1f %%" % (nme, v, 100 * v / st.sum()))
    print("  total %.0f cycles" % st.sum())
if os.environ.get("ILQR_WSTAMPS") and "fold" not in os.environ.get("ILQR_BACKWARD", "wave") and "generic" not in os.environ.get("ILQR_BACKWARD", "wave"):
    names = ["sym + fold (5 transposes), lxx loads, wait staging", "P2 + G0 transposes", "P4 + P5", "P1 tile 3, P3 (3,3)", "Quu -> LDS", "rest of P1", "rest of P3", "chol + Linv",
             "stage next + P6a", "P6b + K store", "P7"]
    st = s.cost()[:11]
    tot = st.sum()
    for nme, v in zip(names, st):
        print("  %-50s %10.0f cycles  %5.1f %%" % (nme, v / N, 100 * v / tot))
    print("  per knot total %.0f cycles (clock64 ticks)" % (tot / N))
elif os.environ.get("ILQR_WSTAMPS"):
    names = ["wait staging, fix, lxx loads", "P2", "P4+P5", "P1 tile 3, P3 (3,3)", "Qu, Quu -> LDS", "rest of P1, P3", "chol + Linv", "stage next + P6a", "P6b + K store", "P7", "transposes"]
    st = s.cost()[:11]
    tot = st.sum()
    for nme, v in zip(names, st):
        print("  %-24s %10.0f cycles  %5.1f %%" % (nme, v / 25, 100 * v / tot))
    print("  per knot total %.0f cycles (clock64 ticks)" % (tot / 25))
if os.environ.get("ILQR_STAMPS"):
    names = ["regs<-staging", "sync", "P1", "P2", "Qx/Qu", "sync", "P3", "P4+P5", "sync", "chol+Linv (wave0)", "sync", "P6a+P6b", "sync", "P7+Vx", "sync", "-"]
    st = s.cost()[:16]
    tot = st.sum()
    for nme, v in zip(names, st):
        print("  %-24s %10.0f cycles  %5.1f %%" % (nme, v / 25, 100 * v / tot))
    print("  per knot total %.0f cycles (clock64 ticks)" % (tot / 25))
s.close()
