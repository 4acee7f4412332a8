"""Round 3 adds solve_golden_n25.npz (full N = 25 horizon, EXACT Jacobians of the Kane step by complex-step differentiation: cost
trace to 1e-9, gains to 1e-7 instead of 1e-4; seed 0 walks accept -> accept -> fail -> retry -> fail -> break) and
solve_golden_contact.npz (contact row f4: gravity -9.81, the stance-constrained step as a NumPy KKT system with the unilateral
release rule, complex-step Jacobians of the whole constrained step).
The whole solve loop against the NumPy / torch restatement in tests/golden/solve_golden.npz (gen_golden.py gen_solve: Kane-step
rollout, central-difference Jacobians of it, torch-autograd cost quadratics, NumPy Riccati, line search, the reference's
lambda / retry / exit rules -- written independently of oracle/ and of the HIP kernels): cost trace, accepted step sizes, lambda
schedule, iteration count, final trajectory and gains.  CPU: pins the oracle's glue; -m gpu: the HIP path through the C ABI."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from conftest import load_package

G = os.path.join(os.path.dirname(__file__), "golden")
pkg = load_package()


def _problem(g):
    N = int(g["N"])
    return dict(N=N, dt=float(g["h"]), Q=g["Q"], R=g["R"], Qf=g["Qf"], task_weights=tuple(float(v) for v in g["task_weights"]),
                w_joint=float(g["w_joint"]), w_ctrl=float(g["w_ctrl"]), gravity=tuple(float(v) for v in g["gravity"]),
                x_ref=g["x_ref"][None], u_ref=g["u_ref"][None], com_ref=g["com_ref"][None], stance=g["stance"][None].astype(np.int32),
                ee_ref=g["ee_ref"][None], com_vel_ref=np.zeros((1, N + 1, 3)))


def _check(g, seed, it, tc, ta, tl, xbar, K, tol_cost=1e-5, tol_x=1e-5, tol_K=1e-4):
    n = int(g["iters_%d" % seed])
    assert it == n
    assert np.allclose(tc[: n + 1], g["trace_cost_%d" % seed], rtol=tol_cost, atol=0), (tc[: n + 1], g["trace_cost_%d" % seed])
    assert np.array_equal(ta[:n], g["trace_alpha_%d" % seed]) and np.allclose(tl[:n], g["trace_lambda_%d" % seed], rtol=1e-12)
    assert np.abs(xbar - g["xbar_%d" % seed]).max() < tol_x * np.abs(g["xbar_%d" % seed]).max()
    Kg = g["K_%d" % seed]
    assert np.abs(K - Kg).max() < tol_K * np.abs(Kg).max(), np.abs(K - Kg).max() / np.abs(Kg).max()


# the round-3 goldens: exact Jacobians in the restatement, so everything pins four to five orders tighter (north_star bar: 1e-5)
EXACT = dict(tol_cost=1e-9, tol_x=1e-9, tol_K=1e-7)
ROUND3 = [("solve_golden_n25.npz", 0), ("solve_golden_n25.npz", 1), ("solve_golden_contact.npz", 0)]


@pytest.mark.parametrize("name,seed", ROUND3)
def test_oracle_reproduces_the_exact_jacobian_restatements(name, seed):
    g = np.load(os.path.join(G, name))
    prob = _problem(g)
    o = ol.Oracle(prob["N"], prob["dt"]); o.set_problem(prob)
    if "contact" in g.files:
        assert prob["gravity"][2] == -9.81
        o.set_contact_mode(int(g["contact"]), float(g["soft"]))
    else:
        assert prob["N"] == 25 and prob["stance"].min() == 0
    o.set_options(max_iter=int(g["max_iter"]), tol=float(g["tol"]), jac_mode=0)
    x0, ui = g["x0_%d" % seed], g["u_init"][seed]
    o.initialize(x0, ui); ok, c = o.solve(x0)
    it, tc, ta, tl = o.trace()
    _check(g, seed, it, tc, ta, tl, o.get("xbar"), o.get("K"), **EXACT)
    if name == "solve_golden_n25.npz" and seed == 0:
        assert list(g["trace_alpha_0"]) == [0.8, 1.0, 0.0]         # the third iteration fails twice and leaves (ilqr.cpp:619-644)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["solve_golden_n25.npz", "solve_golden_contact.npz"])
def test_hip_path_reproduces_the_exact_jacobian_restatements(name):
    from mpc_ilqr_mujoco_amd import solver as sv
    g = np.load(os.path.join(G, name))
    prob = _problem(g)
    ns = g["u_init"].shape[0]
    x0 = np.stack([g["x0_%d" % k] for k in range(ns)]); ui = g["u_init"]
    s = sv.BatchedILQR(ns, N=prob["N"], dt=prob["dt"]); s.set_problem(prob)
    if "contact" in g.files:
        s.set_contact_mode(int(g["contact"]), float(g["soft"]))
    s.set_max_iterations(int(g["max_iter"])); s.set_tolerance(float(g["tol"]))
    s.initialize(x0, ui); cost = s.solve(x0)
    tc, ta, tl = s.trace(); it = s.iterations(); xb = s.xbar(); K = s.gains_K()
    for seed in range(ns):
        _check(g, seed, int(it[seed]), tc[seed], ta[seed], tl[seed], xb[seed], K[seed], **EXACT)
    assert s.adopt_mismatches() == 0
    s.close()


@pytest.mark.parametrize("seed", [0, 1])
def test_oracle_reproduces_the_numpy_solve(seed):
    g = np.load(os.path.join(G, "solve_golden.npz"))
    prob = _problem(g)
    assert prob["stance"].min() == 0                 # a swing phase: the foot-position term is active
    o = ol.Oracle(prob["N"], prob["dt"]); o.set_problem(prob)
    o.set_options(max_iter=int(g["max_iter"]), tol=float(g["tol"]), jac_mode=0)
    x0, ui = g["x0_%d" % seed], g["u_init"][seed]
    o.initialize(x0, ui); ok, c = o.solve(x0)
    it, tc, ta, tl = o.trace()
    assert abs(c - g["trace_cost_%d" % seed][-1]) <= 1e-5 * abs(c)
    _check(g, seed, it, tc, ta, tl, o.get("xbar"), o.get("K"))


@pytest.mark.gpu
def test_hip_path_reproduces_the_numpy_solve():
    from mpc_ilqr_mujoco_amd import solver as sv
    g = np.load(os.path.join(G, "solve_golden.npz"))
    prob = _problem(g)
    x0 = np.stack([g["x0_0"], g["x0_1"]]); ui = g["u_init"]
    s = sv.BatchedILQR(2, N=prob["N"], dt=prob["dt"]); s.set_problem(prob)
    s.set_max_iterations(int(g["max_iter"])); s.set_tolerance(float(g["tol"]))
    s.initialize(x0, ui); cost = s.solve(x0)
    tc, ta, tl = s.trace(); it = s.iterations(); xb = s.xbar(); K = s.gains_K()
    for seed in range(2):
        assert abs(cost[seed] - g["trace_cost_%d" % seed][-1]) <= 1e-5 * abs(cost[seed])
        _check(g, seed, int(it[seed]), tc[seed], ta[seed], tl[seed], xb[seed], K[seed])
    s.close()
