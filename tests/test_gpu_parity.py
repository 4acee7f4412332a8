"""GPU parity: the HIP path (through the C ABI, include/ilqr_hip.h) against the CPU oracle on the same
seeded inputs, against the committed goldens, and size-independent properties at the bench size.
Tolerance: 1e-5 relative on per-iteration costs and feedback gains (BASELINE.json north_star);
the individual stages are held to much tighter bounds."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from conftest import load_package

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
pkg = load_package()
sc = pkg.scenario


def _needs_legacy():
    """does the environment select a cross-check kernel family?  (compiled into lib/libilqr_hip_legacy.so only)"""
    e = os.environ
    return (e.get("ILQR_BACKWARD", "wave") not in ("wave", "wave-generic") or e.get("ILQR_LS", "s")[:1] != "s" or e.get("ILQR_ROLLOUT", "s")[:1] != "s"
            or e.get("ILQR_DYN", "")[:1] == "s" or e.get("ILQR_LINT", "0") == "1")


def _solver(*a, legacy=False, **k):
    """legacy=True: a handle of the test library, for tests that switch to a cross-check family around calls on ONE handle"""
    from mpc_ilqr_mujoco_amd import solver as sv
    if legacy or _needs_legacy():
        k["lib_path"] = sv.LEGACY_LIB_PATH
    return sv.BatchedILQR(*a, **k)


def rel(a, b):
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())


def make(B, N=25, seed=0, gravity=None, walking=False, w_com_vel=0.0):
    from mpc_ilqr_mujoco_amd import solver as sv
    cfg = dict(sc.SHIPPED_CONFIG)
    cfg["W_com_vel"] = w_com_vel
    stance = None
    if walking:  # contact pattern with swing phases so the foot-position term is exercised
        stance = np.ones((N + 1, 2), dtype=np.int32)
        stance[3:9, 0] = 0
        stance[12:20, 1] = 0
        stance[N, :] = 0 if N > 20 else 1
    prob = sc.make_problem(sv.reference_kinematics, N=N, cfg=cfg, stance=stance, gravity=gravity)
    if w_com_vel > 0:
        prob["com_vel_ref"][:] = np.array([0.05, -0.02, 0.01])
    ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
    x0, ui = sc.synthetic_batch(B, N, seed, ug)
    return prob, x0, ui


def oracle_for(prob, b=0, **opts):
    o = ol.Oracle(prob["N"], prob["dt"])
    o.set_problem(prob, b)
    o.set_options(**opts)
    return o


def test_host_helpers_match_oracle():
    from mpc_ilqr_mujoco_amd import solver as sv
    rng = np.random.default_rng(0)
    for _ in range(4):
        x = sc.standing_state(); x[7:26] = rng.uniform(-0.5, 0.5, 19); x[3:7] = sc._axis_angle_quat(rng.uniform(-1, 1, 3))
        c1, e1 = sv.reference_kinematics(x); c2, e2 = ol.reference_kinematics(x)
        assert np.abs(c1 - c2).max() < 1e-13 and np.abs(e1 - e2).max() < 1e-13
        for g in ((0, 0, -9.81), (0, 0, -1.0)):
            o = ol.Oracle(25, 0.02); p = sc.make_problem(ol.reference_kinematics, gravity=g); o.set_problem(p)
            assert np.abs(sv.gravity_compensation(x, g) - o.grav_comp(x)).max() < 1e-11


def test_step_matches_oracle_and_kane_golden():
    d = np.load(os.path.join(G, "dynamics_golden.npz"))
    for g in (np.array([0, 0, -9.81]), np.array([0.0, 0.0, -1.0])):
        idx = [i for i in range(len(d["x"])) if np.allclose(d["gravity"][i], g)]
        prob, x0, ui = make(4, gravity=tuple(g))
        s = _solver(4); s.set_problem(prob)
        got = s.step(d["x"][idx], d["u"][idx])
        assert np.abs(got - d["x_next"][idx]).max() < 1e-10
        o = oracle_for(prob)
        rng = np.random.default_rng(1)
        xs = np.tile(sc.standing_state(), (64, 1)); xs[:, 7:26] = rng.uniform(-0.6, 0.6, (64, 19))
        xs[:, 3:7] = sc._axis_angle_quat(rng.uniform(-2, 2, (64, 3))); xs[:, 26:] = rng.uniform(-2, 2, (64, 25)); xs[0, 26:] = 0
        us = rng.uniform(-60, 60, (64, 19))
        got = s.step(xs, us)
        want = np.array([o.step(x, u) for x, u in zip(xs, us)])
        assert np.abs(got - want).max() < 1e-11
        s.close()


@pytest.mark.parametrize("jac_mode", [0, 1])
def test_linearization_matches_oracle(jac_mode):
    prob, x0, ui = make(3, seed=2)
    s = _solver(3); s.set_problem(prob); s.set_options(jacobian_mode=jac_mode)
    s.initialize(x0, ui)
    s.stage_linearize()
    A, Bm = s.linearization()
    xb, ub = s.xbar(), s.ubar()
    for b in range(3):
        o = oracle_for(prob, jac_mode=jac_mode)
        o.set_trajectory(xb[b], ub[b]); o.linearize()
        tol = 1e-9 if jac_mode == 0 else 2e-5   # FD mode divides O(1e-16) round-off by eps = 1e-5
        assert np.abs(A[b] - o.get("A")).max() < tol and np.abs(Bm[b] - o.get("B")).max() < tol
    s.close()


@pytest.mark.parametrize("walking,w_com_vel", [(False, 0.0), (True, 3.0)])
def test_cost_quadratics_and_total_cost_match_oracle(walking, w_com_vel):
    prob, x0, ui = make(4, seed=3, walking=walking, w_com_vel=w_com_vel)
    ui[1, 5, 4] = 39.5; ui[2, 7, 13] = -17.9      # into the soft torque-limit margin
    x0[3, 7 + 3] = 1.95                            # knee into the soft joint-limit margin
    s = _solver(4); s.set_problem(prob)
    s.initialize(x0, ui)
    s.stage_cost_quadratics()
    lx, lu, lxx, luu = s.quadratics()
    cost = s.stage_total_cost()
    xb, ub = s.xbar(), s.ubar()
    for b in range(4):
        o = oracle_for(prob)
        o.set_trajectory(xb[b], ub[b]); o.cost_quadratics()
        for name, got in (("lx", lx[b]), ("lu", lu[b]), ("lxx", lxx[b]), ("luu", luu[b])):
            want = o.get(name)
            assert np.abs(got - want).max() <= 1e-10 * max(1.0, np.abs(want).max()), (name, b, np.abs(got - want).max())
        assert abs(cost[b] - o.total_cost()) <= 1e-11 * abs(cost[b])
    s.close()


@pytest.mark.parametrize("case", ["spd", "bump"])
def test_backward_pass_vs_numpy_golden(case):
    r = np.load(os.path.join(G, "riccati_golden.npz"))
    A, Bm = r[case + "_A"], r[case + "_B"]
    N = A.shape[0]
    s = _solver(2, N=N)
    s.set_regularization(float(r["lam"]))
    rep = lambda a: np.stack([a, a])
    s.set_linearization(rep(A), rep(Bm))
    s.set_quadratics(rep(r[case + "_lx"]), rep(r[case + "_lu"]), rep(r[case + "_lxx"]), rep(r[case + "_luu"]))
    s.stage_backward_pass()
    K, kff = s.gains_K(), s.gains_kff()
    Vx, Vxx = s.value_function()
    tol = 1e-9 if case == "spd" else 1e-6
    for got, key in ((K, "K"), (kff, "k"), (Vx, "Vx"), (Vxx, "Vxx")):
        want = r[case + "_" + key]
        for b in range(2):
            assert np.abs(got[b] - want).max() <= tol * max(1.0, np.abs(want).max()), (key, np.abs(got[b] - want).max())
    s.close()


def test_backward_pass_indefinite_quu_fallback():
    """Quu indefinite even after the +1e-4 bump (ilqr.cpp:278-281): the reference's ldlt() solves the symmetric
    indefinite system; the kernels fall back to an explicit pivoted inverse.  Checked against a NumPy restatement of
    ilqr.cpp:250-309 (long-form Vx / Vxx, symmetrised)."""
    rng = np.random.default_rng(11)
    N, n, m, lam = 4, 51, 19, 1e-6
    A = np.eye(n)[None] + 0.05 * rng.standard_normal((N, n, n))
    Bm = 0.1 * rng.standard_normal((N, n, m))
    lx = rng.standard_normal((N + 1, n)); lu = rng.standard_normal((N, m))
    lxx = np.stack([np.diag(rng.uniform(1.0, 3.0, n)) for _ in range(N + 1)])
    luu = rng.uniform(0.5, 1.5, (N, m))
    luu[2, 3] = -40.0; luu[1, 7] = -25.0            # indefinite Quu at knots 2 and 1
    Vx, Vxx = lx[N].copy(), lxx[N].copy()
    Kw, kw = np.zeros((N, m, n)), np.zeros((N, m))
    n_indef = 0
    for t in range(N - 1, -1, -1):
        Qx = lx[t] + A[t].T @ Vx; Qu = lu[t] + Bm[t].T @ Vx
        Qxx = lxx[t] + A[t].T @ Vxx @ A[t]
        Quu = np.diag(luu[t]) + Bm[t].T @ Vxx @ Bm[t] + lam * np.eye(m)
        Qxu = A[t].T @ Vxx @ Bm[t]
        if np.linalg.eigvalsh(Quu).min() <= 0:
            Quu = Quu + 1e-4 * np.eye(m)
            n_indef += int(np.linalg.eigvalsh(Quu).min() <= 0)
        K = -np.linalg.solve(Quu, Qxu.T); k = -np.linalg.solve(Quu, Qu)
        Vx = Qx + K.T @ Quu @ k + K.T @ Qu + Qxu @ k
        Vxx = Qxx + K.T @ Quu @ K + K.T @ Qxu.T + Qxu @ K
        Vxx = 0.5 * (Vxx + Vxx.T)
        Kw[t], kw[t] = K, k
    assert n_indef >= 2
    s = _solver(3, N=N)
    s.set_regularization(lam)
    rep = lambda a: np.stack([a] * 3)
    s.set_linearization(rep(A), rep(Bm))
    s.set_quadratics(rep(lx), rep(lu), rep(lxx), rep(luu))
    s.stage_backward_pass()
    K, kff = s.gains_K(), s.gains_kff()
    gVx, gVxx = s.value_function()
    for got, want, key in ((K, Kw, "K"), (kff, kw, "k"), (gVx, Vx, "Vx"), (gVxx, Vxx, "Vxx")):
        for b in range(3):
            assert np.abs(got[b] - want).max() <= 1e-7 * max(1.0, np.abs(want).max()), (key, np.abs(got[b] - want).max())
    s.close()


def test_backward_pass_and_line_search_match_oracle():
    prob, x0, ui = make(3, seed=4)
    s = _solver(3); s.set_problem(prob)
    s.initialize(x0, ui)
    s.stage_linearize(); s.stage_cost_quadratics(); s.stage_backward_pass()
    K, kff = s.gains_K(), s.gains_kff()
    xb, ub = s.xbar(), s.ubar()
    A, Bm = s.linearization(); lx, lu, lxx, luu = s.quadratics()
    imp, cost, alpha = s.stage_line_search()
    xn, un = s.xbar(), s.ubar()
    for b in range(3):
        o = oracle_for(prob)
        o.set_trajectory(xb[b], ub[b]); o.set_linearization(A[b], Bm[b]); o.set_quadratics(lx[b], lu[b], lxx[b], luu[b])
        o.backward_pass()
        assert rel(K[b], o.get("K")) < 1e-8 and rel(kff[b], o.get("kff")) < 1e-8
        ok, c, a = o.line_search(x0[b])
        assert ok == bool(imp[b]) and a == alpha[b] and abs(c - cost[b]) < 1e-8 * abs(c)
        assert rel(xn[b], o.get("xbar")) < 1e-8 and rel(un[b], o.get("ubar")) < 1e-7
    s.close()


@pytest.mark.parametrize("walking,gravity,seed", [(False, None, 0), (True, (0.0, 0.0, -2.0), 5)])
def test_full_solve_parity_trace_and_gains(walking, gravity, seed):
    """north_star: per-iteration costs and feedback gains within 1e-5 relative of the CPU reference."""
    B = 6
    prob, x0, ui = make(B, seed=seed, walking=walking, gravity=gravity)
    s = _solver(B); s.set_problem(prob)
    s.initialize(x0, ui)
    cost = s.solve(x0)
    tc, ta, tl = s.trace()
    K, kff, xb, ub, it, lam = s.gains_K(), s.gains_kff(), s.xbar(), s.ubar(), s.iterations(), s.lambdas()
    for b in range(B):
        o = oracle_for(prob)
        o.initialize(x0[b], ui[b])
        ok, c = o.solve(x0[b])
        n, oc, oa, olam = o.trace()
        assert n == it[b], (b, n, it[b])
        assert np.allclose(tc[b, : n + 1], oc[: n + 1], rtol=1e-5, atol=0), (b, tc[b, : n + 1], oc[: n + 1])
        assert np.array_equal(ta[b, :n], oa[:n]) and np.allclose(tl[b, :n], olam[:n], rtol=1e-12)
        assert abs(cost[b] - c) <= 1e-5 * abs(c)
        assert rel(K[b], o.get("K")) < 1e-5 and rel(kff[b], o.get("kff")) < 1e-5
        assert rel(xb[b], o.get("xbar")) < 1e-5 and rel(ub[b], o.get("ubar")) < 1e-5
        assert abs(lam[b] - o.get_lambda()) < 1e-18
    s.close()


def _bench_batch(seed=0, gravity=None, B=4096, N=25):
    """The batch bench.py times (scenario.synthetic_batch of the WHOLE batch: a rollout's inputs depend on its index in it)."""
    from mpc_ilqr_mujoco_amd import solver as sv
    prob = sc.make_problem(sv.reference_kinematics, N=N, gravity=gravity)
    ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
    x0, ui = sc.synthetic_batch(B, N, seed, ug)
    return prob, x0, ui


def test_the_mode_the_bench_times_fixed_ten_iterations_analytic_vs_oracle():
    """The headline configuration exactly as bench.py runs it -- analytic Jacobians, early_exit = 0, max_iter = 10, rollouts of the
    seed-0 bench batch -- against the oracle: the FULL cost trace, every accepted step size, the lambda schedule and the final
    gains.  Iterations 4..9 of these rollouts are the fail -> lambda x 10 -> retry -> fail -> `continue` chain of ilqr.cpp:619-644
    with lambda saturating at 1e-3: a quarter of the timed work."""
    prob, x0g, uig = _bench_batch()
    pick = list(range(12)) + [777, 2048, 3333, 4095]
    x0, ui = x0g[pick], uig[pick]
    B = len(pick)
    from mpc_ilqr_mujoco_amd import solver as sv
    s = _solver(B); s.set_problem(prob); s.set_max_iterations(10)
    s.set_options(jacobian_mode=sv.JAC_ANALYTIC, early_exit=False)
    s.initialize(x0, ui)
    cost = s.solve(x0)
    tc, ta, tl = s.trace()
    K, kff, xb, ub, it, lam = s.gains_K(), s.gains_kff(), s.xbar(), s.ubar(), s.iterations(), s.lambdas()
    assert np.all(it == 10) and s.adopt_mismatches() == 0
    saw_saturation = saw_retry_chain = False
    for b in range(B):
        o = oracle_for(prob, jac_mode=0, early_exit=0, max_iter=10)
        o.initialize(x0[b], ui[b]); ok, c = o.solve(x0[b])
        n, oc, oa, olam = o.trace()
        assert n == 10
        assert np.allclose(tc[b], oc, rtol=1e-5, atol=0), (b, tc[b], oc)
        assert np.array_equal(ta[b], oa), (b, ta[b], oa)                       # accepted step sizes (0 = both searches failed)
        assert np.allclose(tl[b], olam, rtol=1e-12, atol=0), (b, tl[b], olam)
        assert abs(cost[b] - c) <= 1e-5 * abs(c) and abs(lam[b] - o.get_lambda()) < 1e-18
        assert rel(K[b], o.get("K")) < 1e-5 and rel(kff[b], o.get("kff")) < 1e-5
        assert rel(xb[b], o.get("xbar")) < 1e-5 and rel(ub[b], o.get("ubar")) < 1e-5
        saw_saturation |= bool(np.any(olam >= 1e-3 * (1 - 1e-12)) or o.get_lambda() >= 1e-3 * (1 - 1e-12))
        saw_retry_chain |= bool(np.sum(oa[4:] == 0.0) >= 3)
    assert saw_saturation and saw_retry_chain      # the chain the docstring names is really exercised by these rollouts
    s.close()


def test_the_contact_leg_of_the_bench_fixed_ten_iterations_vs_oracle():
    """bench.py's `contact` object: gravity -9.81, both feet scheduled in stance, contact mode 2 (unilateral), analytic Jacobians of
    the constrained step, 10 fixed iterations -- the same rollouts on the oracle (its forward-mode AD through the constrained step)."""
    prob, x0g, uig = _bench_batch(gravity=(0.0, 0.0, -9.81))
    pick = [0, 1, 777, 4095]
    x0, ui = x0g[pick], uig[pick]
    B = len(pick)
    from mpc_ilqr_mujoco_amd import solver as sv
    s = _solver(B); s.set_problem(prob); s.set_contact_mode(2); s.set_max_iterations(10)
    s.set_options(jacobian_mode=sv.JAC_ANALYTIC, early_exit=False)
    s.initialize(x0, ui)
    cost = s.solve(x0)
    tc, ta, tl = s.trace()
    K, it = s.gains_K(), s.iterations()
    assert np.all(it == 10) and s.adopt_mismatches() == 0
    for b in range(B):
        o = oracle_for(prob, jac_mode=0, early_exit=0, max_iter=10); o.set_contact_mode(2)
        o.initialize(x0[b], ui[b]); ok, c = o.solve(x0[b])
        n, oc, oa, olam = o.trace()
        assert n == 10 and np.allclose(tc[b], oc, rtol=1e-5, atol=0), (b, tc[b], oc)
        assert np.array_equal(ta[b], oa) and np.allclose(tl[b], olam, rtol=1e-12, atol=0)
        assert abs(cost[b] - c) <= 1e-5 * abs(c) and rel(K[b], o.get("K")) < 1e-5
    s.close()


def test_fixed_iteration_mode_and_fd_mode_parity():
    B = 3
    prob, x0, ui = make(B, seed=6)
    s = _solver(B); s.set_problem(prob); s.set_options(jacobian_mode=1, fd_eps=1e-5, early_exit=False); s.set_max_iterations(4)
    s.initialize(x0, ui)
    cost = s.solve(x0)
    tc, ta, tl = s.trace()
    assert np.all(s.iterations() == 4)
    for b in range(B):
        o = oracle_for(prob, jac_mode=1, fd_eps=1e-5, early_exit=0, max_iter=4)
        o.initialize(x0[b], ui[b]); ok, c = o.solve(x0[b])
        n, oc, oa, ol_ = o.trace()
        assert n == 4 and np.allclose(tc[b], oc, rtol=1e-5) and np.array_equal(ta[b], oa)
        assert rel(s.gains_K()[b], o.get("K")) < 1e-4   # FD noise (1e-16/1e-5) amplified through the recursion
    # The 1e-4 above is round-off / eps of two INDEPENDENT forward-difference evaluations carried through four iterations, not solver
    # error.  north_star's 1e-5 on the gains, for the reference's own Jacobian scheme, is checked on the SAME forward-difference
    # (A_t, B_t): the GPU's FD Jacobians about the solved trajectory go into the oracle (set_linearization), both run their own cost
    # quadratics and backward pass with the rollout's own lambda.
    xb, ub, lam = s.xbar(), s.ubar(), s.lambdas()
    s.set_trajectory(xb, ub)
    s.stage_linearize(); s.stage_cost_quadratics(); s.stage_backward_pass()
    A, Bm = s.linearization(); K, kff = s.gains_K(), s.gains_kff()
    for b in range(B):
        o = oracle_for(prob, jac_mode=1, fd_eps=1e-5, early_exit=0, lam=float(lam[b]))
        o.set_trajectory(xb[b], ub[b])
        o.linearize()                                   # the oracle's own forward differences: the scheme itself, entry by entry
        assert np.abs(A[b] - o.get("A")).max() < 1e-6 and np.abs(Bm[b] - o.get("B")).max() < 1e-6
        o.set_linearization(A[b], Bm[b]); o.cost_quadratics(); o.backward_pass()
        assert rel(K[b], o.get("K")) < 1e-5 and rel(kff[b], o.get("kff")) < 1e-5, (b, rel(K[b], o.get("K")), rel(kff[b], o.get("kff")))
    s.close()


def test_contact_mode_step_and_solve_match_oracle():
    """rigid stance constraints on the scheduled feet.  (i) the stance-constrained step on the GPU equals
    the oracle's for every stance pattern; (ii) a full solve in contact mode (scalar kernels, forward-difference
    Jacobians as the reference takes them) reproduces the oracle's cost trace, accepted step sizes and gains on a
    schedule with swing phases, under physical gravity (the feet carry the robot)."""
    B = 3
    prob, x0, ui = make(B, seed=8, gravity=[0.0, 0.0, -9.81], walking=True)
    s = _solver(B); s.set_problem(prob); s.set_contact_mode(1)
    o = oracle_for(prob); o.set_contact_mode(1)
    rng = np.random.default_rng(5)
    xs = x0.copy(); xs[:, 26:] += rng.uniform(-0.2, 0.2, (B, 25)); xs[:, 7 + 3] += 0.3; xs[:, 7 + 8] += 0.3
    us = ui[:, 0, :]
    free = s.step_stance(xs, us, 0, 0)
    for sl, sr in ((1, 1), (1, 0), (0, 1), (0, 0)):
        got = s.step_stance(xs, us, sl, sr)
        for b in range(B):
            want = o.step_stance(xs[b], us[b], [sl, sr])
            assert np.abs(got[b] - want).max() < 1e-9 * max(1.0, np.abs(want).max()), (sl, sr, b, np.abs(got[b] - want).max())
        if sl or sr:
            assert np.abs(got - free).max() > 1e-3
    s.set_options(jacobian_mode=1, fd_eps=1e-5, early_exit=False); s.set_max_iterations(3)
    s.initialize(x0, ui)
    cost = s.solve(x0)
    tc, ta, tl = s.trace()
    for b in range(B):
        ob = oracle_for(prob, jac_mode=1, fd_eps=1e-5, early_exit=0, max_iter=3); ob.set_contact_mode(1)
        ob.initialize(x0[b], ui[b]); ok, c = ob.solve(x0[b])
        n, oc, oa, ol_ = ob.trace()
        assert n == 3 and np.allclose(tc[b], oc, rtol=1e-5) and np.array_equal(ta[b], oa), (tc[b], oc, ta[b], oa)
        assert abs(cost[b] - c) <= 1e-5 * abs(c)
        assert rel(s.gains_K()[b], ob.get("K")) < 1e-4
        assert rel(s.xbar()[b], ob.get("xbar")) < 1e-5
    s.close()


def test_unilateral_contact_mode_releases_pulled_feet_and_matches_oracle():
    """Contact mode 2 (ILQR_CONTACT_UNILATERAL_STANCE): the floor pushes, it does not pull.  (i) step parity GPU (two-lane
    kernels) <-> oracle for states where a scheduled stance foot is pulled off the ground (that foot is released) and where it
    is not (identical to mode 1); (ii) a full solve in mode 2 against the oracle; (iii) the nominal re-rollout reproduces the
    accepted candidates bit for bit in contact mode too."""
    B = 4
    prob, x0, ui = make(B, seed=18, gravity=[0.0, 0.0, -9.81], walking=True)
    s = _solver(B); s.set_problem(prob)
    o = oracle_for(prob)
    rng = np.random.default_rng(7)
    xs = x0.copy(); xs[:, 26:] += rng.uniform(-0.2, 0.2, (B, 25)); xs[:, 7 + 3] += 0.3; xs[:, 7 + 8] += 0.3
    xs[0, 28] = 3.0                               # pelvis moving up fast: both feet would have to be pulled along
    xs[1, 26 + 6 + 3] = -6.0                      # left knee flexing fast: the left foot lifts
    us = ui[:, 0, :]
    released = 0
    for sl, sr in ((1, 1), (1, 0), (0, 1)):
        s.set_contact_mode(1); got1 = s.step_stance(xs, us, sl, sr)
        s.set_contact_mode(2); got2 = s.step_stance(xs, us, sl, sr)
        for b in range(B):
            o.set_contact_mode(2); want2 = o.step_stance(xs[b], us[b], [sl, sr])
            o.set_contact_mode(1); want1 = o.step_stance(xs[b], us[b], [sl, sr])
            assert np.abs(got2[b] - want2).max() < 1e-9 * max(1.0, np.abs(want2).max()), (sl, sr, b, np.abs(got2[b] - want2).max())
            assert np.abs(got1[b] - want1).max() < 1e-9 * max(1.0, np.abs(want1).max())
            released += int(np.abs(want2 - want1).max() > 1e-6)
    assert released >= 2                           # the unilateral rule actually fired
    assert np.abs(got2[3] - got1[3]).max() < 1e-12   # an unperturbed standing rollout keeps both feet
    s.set_contact_mode(2)
    s.set_options(jacobian_mode=1, fd_eps=1e-5, early_exit=False); s.set_max_iterations(3)
    s.initialize(x0, ui)
    cost = s.solve(x0)
    tc, ta, tl = s.trace()
    assert s.adopt_mismatches() == 0
    for b in range(B):
        ob = oracle_for(prob, jac_mode=1, fd_eps=1e-5, early_exit=0, max_iter=3); ob.set_contact_mode(2)
        ob.initialize(x0[b], ui[b]); ok, c = ob.solve(x0[b])
        n, oc, oa, ol_ = ob.trace()
        assert n == 3 and np.allclose(tc[b], oc, rtol=1e-5) and np.array_equal(ta[b], oa), (tc[b], oc, ta[b], oa)
        assert abs(cost[b] - c) <= 1e-5 * abs(c) and rel(s.gains_K()[b], ob.get("K")) < 1e-4 and rel(s.xbar()[b], ob.get("xbar")) < 1e-5
    s.close()


def test_friction_limited_contact_mode_slides_feet_outside_the_cone_and_matches_oracle_and_golden():
    """Contact mode 3 (ILQR_CONTACT_FRICTION_STANCE): unilateral stance + Coulomb limit.  (i) the step on the two-lane kernels
    against the committed vectors of the independent NumPy KKT formulation (tests/golden/friction_golden.npz: no foot, one foot,
    both feet sliding) and against the oracle, 1e-9; (ii) with the cone inactive the step IS mode 2, bit for bit; (iv) a full solve with the reference's forward-difference Jacobians on a walking
    schedule (mu = 0.7) reproduces the oracle's trace, and the nominal re-rollout reproduces the accepted
    candidates bit for bit."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "friction_golden.npz"))
    n = len(g["x"])
    B = n
    prob, x0, ui = make(B, seed=28, gravity=list(g["gravity"]), walking=True)
    s = _solver(B); s.set_problem(prob)
    o = oracle_for(prob)
    slid = 0
    for i in range(n):
        mu = float(g["mu"][i])
        xs = np.tile(g["x"][i], (B, 1)); us = np.tile(g["u"][i], (B, 1))
        s.set_contact_mode(3, float(g["soft"])); s.set_friction(mu)
        got3 = s.step_stance(xs, us, int(g["stance"][i][0]), int(g["stance"][i][1]))
        s.set_contact_mode(2, float(g["soft"]))
        got2 = s.step_stance(xs, us, int(g["stance"][i][0]), int(g["stance"][i][1]))
        o.set_contact_mode(3, float(g["soft"])); o.set_friction(mu)
        want = o.step_stance(g["x"][i], g["u"][i], g["stance"][i])
        assert np.abs(got3 - got3[0]).max() == 0.0                                    # every lane pair alike
        assert np.abs(got3[0] - g["x_next"][i]).max() < 1e-9 * max(1.0, np.abs(g["x_next"][i]).max()), (i, np.abs(got3[0] - g["x_next"][i]).max())
        assert np.abs(got3[0] - want).max() < 1e-9 * max(1.0, np.abs(want).max())
        assert np.abs(got2[0] - g["x_next_mode2"][i]).max() < 1e-9 * max(1.0, np.abs(g["x_next_mode2"][i]).max())
        # mode 4 (kinetic friction on the sliding feet): golden of the unsymmetric KKT formulation, oracle
        s.set_contact_mode(4, float(g["soft"]))
        got4 = s.step_stance(xs, us, int(g["stance"][i][0]), int(g["stance"][i][1]))
        o.set_contact_mode(4, float(g["soft"]))
        want4 = o.step_stance(g["x"][i], g["u"][i], g["stance"][i])
        assert np.abs(got4 - got4[0]).max() == 0.0
        assert np.abs(got4[0] - g["x_next_mode4"][i]).max() < 1e-9 * max(1.0, np.abs(g["x_next_mode4"][i]).max()), (i, np.abs(got4[0] - g["x_next_mode4"][i]).max())
        assert np.abs(got4[0] - want4).max() < 1e-9 * max(1.0, np.abs(want4).max())
        if g["slide"][i].any():
            slid += 1
            assert np.abs(got3[0] - got2[0]).max() > 1e-3 and np.abs(got4[0] - got3[0]).max() > 1e-3
        else:
            assert np.array_equal(got3[0], got2[0]) and np.array_equal(got4[0], got2[0])
    assert slid >= 5
    # (iii) analytic Jacobians in these modes: test_sliding_contact_analytic_jacobians_match_oracle_ad
    # (iv) the solve, forward differences on both sides
    Bs = 4
    s.close()
    prob, x0, ui = make(Bs, seed=28, gravity=list(g["gravity"]), walking=True)
    for cmode in (3, 4):
        s = _solver(Bs); s.set_problem(prob); s.set_contact_mode(cmode); s.set_friction(0.7)   # (the cold-started gait loads its feet sideways)
        s.set_options(jacobian_mode=1, fd_eps=1e-5, early_exit=False); s.set_max_iterations(3)
        s.initialize(x0, ui)
        cost = s.solve(x0)
        tc, ta, tl = s.trace()
        assert s.adopt_mismatches() == 0
        differs = 0
        for b in range(Bs):
            ob = oracle_for(prob, jac_mode=1, fd_eps=1e-5, early_exit=0, max_iter=3); ob.set_contact_mode(cmode); ob.set_friction(0.7)
            ob.initialize(x0[b], ui[b]); ok, c = ob.solve(x0[b])
            nn, oc, oa, ol_ = ob.trace()
            assert nn == 3 and np.allclose(tc[b], oc, rtol=1e-5) and np.array_equal(ta[b], oa), (cmode, tc[b], oc, ta[b], oa)
            assert abs(cost[b] - c) <= 1e-5 * abs(c) and rel(s.gains_K()[b], ob.get("K")) < 1e-4 and rel(s.xbar()[b], ob.get("xbar")) < 1e-5
            o2 = oracle_for(prob, jac_mode=1, fd_eps=1e-5, early_exit=0, max_iter=3); o2.set_contact_mode(2)
            o2.initialize(x0[b], ui[b]); _, c2 = o2.solve(x0[b])
            differs += int(abs(c2 - c) > 1e-6 * abs(c))
        assert differs >= 1                              # the cone was active somewhere along these solves
        s.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_contact_analytic_jacobians_match_oracle_ad(mode):
    """Contact row f4: analytic Jacobians of the stance-constrained step (k_lin_tangent_c: KKT differentiation with the contact
    wrench as an external force in the tangent sweeps, unit-wrench columns G = Mhat^-1 J^T, constraint-row tangents) against
    the oracle's forward-mode AD through its constrained step -- exact derivatives on both sides, not finite differences --
    on a schedule with double support, single support and flight knots, under physical gravity."""
    B = 3
    prob, x0, ui = make(B, seed=23, gravity=[0.0, 0.0, -9.81], walking=True)
    prob["stance"] = prob["stance"].copy(); prob["stance"][0, 10:12, :] = 0          # two flight knots
    s = _solver(B); s.set_problem(prob); s.set_contact_mode(mode); s.set_options(jacobian_mode=0)
    if mode == 2:
        x0 = x0.copy(); x0[1, 28] = 2.5                                               # pelvis moving up: feet released at the first knots
    s.initialize(x0, ui)
    s.stage_linearize()
    A, Bm = s.linearization()
    xb, ub = s.xbar(), s.ubar()
    for b in range(B):
        o = oracle_for(prob, jac_mode=0); o.set_contact_mode(mode)
        o.set_trajectory(xb[b], ub[b]); o.linearize()
        Ao, Bo = o.get("A"), o.get("B")
        assert np.abs(A[b] - Ao).max() < 1e-8 * max(1.0, np.abs(Ao).max()), (b, np.abs(A[b] - Ao).max(), np.unravel_index(np.abs(A[b] - Ao).argmax(), Ao.shape))
        assert np.abs(Bm[b] - Bo).max() < 1e-8 * max(1.0, np.abs(Bo).max()), (b, np.abs(Bm[b] - Bo).max())
    # and a full solve with these Jacobians against the oracle (AD Jacobians, same contact mode)
    s.set_options(jacobian_mode=0, early_exit=False); s.set_max_iterations(3)
    s.initialize(x0, ui); cost = s.solve(x0)
    tc, ta, tl = s.trace()
    assert s.adopt_mismatches() == 0
    for b in range(B):
        ob = oracle_for(prob, jac_mode=0, early_exit=0, max_iter=3); ob.set_contact_mode(mode)
        ob.initialize(x0[b], ui[b]); ok, c = ob.solve(x0[b])
        n, oc, oa, ol_ = ob.trace()
        assert n == 3 and np.allclose(tc[b], oc, rtol=1e-5) and np.array_equal(ta[b], oa), (tc[b], oc, ta[b], oa)
        assert abs(cost[b] - c) <= 1e-5 * abs(c) and rel(s.gains_K()[b], ob.get("K")) < 1e-5 and rel(s.xbar()[b], ob.get("xbar")) < 1e-5
    s.close()


@pytest.mark.parametrize("mode", [3, 4])
def test_sliding_contact_analytic_jacobians_match_oracle_ad(mode):
    """Contact modes 3 / 4 (Coulomb limit, kinetic friction): analytic Jacobians of the step with sliding feet -- the normal row and
    the normal force turn with the foot (tangent of the up axis in link coordinates down the leg), mode 4's friction direction follows
    the tangent of the sticking solve -- against the oracle's forward-mode AD through the same branches, on the committed friction
    states (tests/golden/friction_golden.npz: no foot, the left, the right, both feet sliding), each with its own mu.  Then a full
    solve on a walking schedule against the oracle with AD Jacobians."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "friction_golden.npz"))
    n, N = len(g["x"]), 25
    prob, x0, ui = make(n, seed=28, gravity=list(g["gravity"]), walking=True)
    prob["stance"] = np.ones_like(prob["stance"])
    X = np.repeat(g["x"][:, None, :], N + 1, axis=1); U = np.repeat(g["u"][:, None, :], N, axis=1)
    slid = 0
    for mu in sorted(set(float(m) for m in g["mu"])):
        s = _solver(n); s.set_problem(prob); s.set_contact_mode(mode, float(g["soft"])); s.set_friction(mu); s.set_options(jacobian_mode=0)
        s.initialize(x0, ui); s.set_trajectory(X, U); s.stage_linearize()
        A, Bm = s.linearization()
        for i in [i for i in range(n) if float(g["mu"][i]) == mu]:
            o = oracle_for(prob, jac_mode=0); o.set_contact_mode(mode, float(g["soft"])); o.set_friction(mu)
            o.set_trajectory(X[i], U[i]); o.linearize()
            Ao, Bo = o.get("A")[0], o.get("B")[0]
            assert np.abs(A[i][0] - Ao).max() < 1e-8 * max(1.0, np.abs(Ao).max()), (mode, i, np.abs(A[i][0] - Ao).max(), np.unravel_index(np.abs(A[i][0] - Ao).argmax(), Ao.shape))
            assert np.abs(Bm[i][0] - Bo).max() < 1e-8 * max(1.0, np.abs(Bo).max()), (mode, i, np.abs(Bm[i][0] - Bo).max())
            if g["slide"][i].any():
                # the sliding branch is a different derivative: the mode-2 Jacobians of the same state are far away
                o2 = oracle_for(prob, jac_mode=0); o2.set_contact_mode(2, float(g["soft"])); o2.set_trajectory(X[i], U[i]); o2.linearize()
                assert np.abs(o2.get("A")[0] - Ao).max() > 1e-3
                slid += 1
        s.close()
    assert slid >= 5
    Bs = 3
    prob, x0, ui = make(Bs, seed=28, gravity=list(g["gravity"]), walking=True)
    s = _solver(Bs); s.set_problem(prob); s.set_contact_mode(mode); s.set_friction(0.7)
    s.set_options(jacobian_mode=0, early_exit=False); s.set_max_iterations(3)
    s.initialize(x0, ui); cost = s.solve(x0)
    tc, ta, tl = s.trace()
    assert s.adopt_mismatches() == 0
    differs = 0
    for b in range(Bs):
        ob = oracle_for(prob, jac_mode=0, early_exit=0, max_iter=3); ob.set_contact_mode(mode); ob.set_friction(0.7)
        ob.initialize(x0[b], ui[b]); ok, c = ob.solve(x0[b])
        nn, oc, oa, ol_ = ob.trace()
        assert nn == 3 and np.allclose(tc[b], oc, rtol=1e-5) and np.array_equal(ta[b], oa), (mode, tc[b], oc, ta[b], oa)
        assert abs(cost[b] - c) <= 1e-5 * abs(c) and rel(s.gains_K()[b], ob.get("K")) < 1e-5 and rel(s.xbar()[b], ob.get("xbar")) < 1e-5
        o2 = oracle_for(prob, jac_mode=0, early_exit=0, max_iter=3); o2.set_contact_mode(2)
        o2.initialize(x0[b], ui[b]); _, c2 = o2.solve(x0[b])
        differs += int(abs(c2 - c) > 1e-6 * abs(c))
    assert differs >= 1                              # the cone was active somewhere along these solves
    s.close()


def test_joint_limit_rows_step_and_solve_match_golden_and_oracle():
    """Joint-limit rows of the plant (include/ilqr_hip.h ilqr_hip_set_joint_limits): a hinge
    past its range that the step would still move outward is stopped.  On the GPU a stopped hinge is a hinge with armature 2^1000 in a
    second pass of the articulated-body recursion (exactly an acceleration-prescribed joint); the committed vectors come from the dense
    NumPy KKT system with explicit rows (tests/golden/joint_limit_golden.npz), the oracle from its own hybrid recursion.  (i) the step,
    1e-9, with and without unilateral stance, none / one / several hinges stopped; switched off it IS the unlimited step; (ii) a solve
    with the reference's forward-difference Jacobians from a state with a knee and an elbow past their ranges, against the oracle;
    (iii) the same solve with analytic Jacobians; (iv) the analytic Jacobians against the oracle's forward-mode AD."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "joint_limit_golden.npz"))
    n = len(g["x"])
    prob, x0, ui = make(2, seed=31, gravity=list(g["gravity"]), walking=True)
    s = _solver(2); s.set_problem(prob)
    o = oracle_for(prob)
    stopped = 0
    for i in range(n):
        cm = int(g["contact"][i])
        xs = np.tile(g["x"][i], (2, 1)); us = np.tile(g["u"][i], (2, 1))
        s.set_contact_mode(cm, float(g["soft"])); s.set_joint_limits(True)
        got = s.step_stance(xs, us, int(g["stance"][i][0]), int(g["stance"][i][1]))
        s.set_joint_limits(False)
        got0 = s.step_stance(xs, us, int(g["stance"][i][0]), int(g["stance"][i][1]))
        o.set_contact_mode(cm, float(g["soft"])); o.set_joint_limits(True)
        want = o.step_stance(g["x"][i], g["u"][i], g["stance"][i])
        assert np.array_equal(got[0], got[1])
        assert np.abs(got[0] - g["x_next"][i]).max() < 1e-9 * max(1.0, np.abs(g["x_next"][i]).max()), (i, np.abs(got[0] - g["x_next"][i]).max())
        assert np.abs(got[0] - want).max() < 1e-9 * max(1.0, np.abs(want).max())
        assert np.abs(got0[0] - g["x_next_unlimited"][i]).max() < 1e-9 * max(1.0, np.abs(g["x_next_unlimited"][i]).max())
        lock = np.flatnonzero(g["lock"][i])
        for j in lock:
            assert abs(got[0][32 + j]) < 1e-12
        stopped += int(len(lock) > 0)
        if len(lock) == 0:
            assert np.abs(got[0] - got0[0]).max() < 1e-12
    assert stopped >= 8
    s.close()
    # (ii) a solve: left knee (hinge 3: range -0.26 .. 2.05) and right elbow (hinge 18: -1.25 .. 2.61) past their ranges, moving outward
    Bs = 3
    for cm in (0, 2):
        prob, x0, ui = make(Bs, seed=31, gravity=list(g["gravity"]) if cm else None, walking=True)
        x0 = x0.copy()
        x0[:, 7 + 3] = 2.09; x0[:, 32 + 3] = 1.5
        x0[1:, 7 + 18] = -1.28; x0[1:, 32 + 18] = -2.0
        s = _solver(Bs); s.set_problem(prob); s.set_contact_mode(cm); s.set_joint_limits(True)
        s.set_options(jacobian_mode=1, fd_eps=1e-5, early_exit=False); s.set_max_iterations(3)
        s.initialize(x0, ui); cost = s.solve(x0)
        tc, ta, tl = s.trace()
        assert s.adopt_mismatches() == 0
        differs = 0
        for b in range(Bs):
            ob = oracle_for(prob, jac_mode=1, fd_eps=1e-5, early_exit=0, max_iter=3); ob.set_contact_mode(cm); ob.set_joint_limits(True)
            ob.initialize(x0[b], ui[b]); ok, c = ob.solve(x0[b])
            nn, oc, oa, ol_ = ob.trace()
            assert nn == 3 and np.allclose(tc[b], oc, rtol=1e-5) and np.array_equal(ta[b], oa), (cm, tc[b], oc, ta[b], oa)
            assert abs(cost[b] - c) <= 1e-5 * abs(c) and rel(s.gains_K()[b], ob.get("K")) < 1e-4 and rel(s.xbar()[b], ob.get("xbar")) < 1e-5
            o2 = oracle_for(prob, jac_mode=1, fd_eps=1e-5, early_exit=0, max_iter=3); o2.set_contact_mode(cm)
            o2.initialize(x0[b], ui[b]); _, c2 = o2.solve(x0[b])
            differs += int(abs(c2 - c) > 1e-6 * abs(c))
        assert differs >= 2                              # the rows mattered
        # (iii) the same solve with the analytic Jacobians (operand-layout Riccati kernel: the position rows of a stopped hinge still are
        # e + h * its velocity row), against the oracle with AD Jacobians
        s.set_options(jacobian_mode=0, early_exit=False); s.set_max_iterations(3)
        s.set_regularization(1e-6)        # (lambda is a member that outlives a solve, ilqr.cpp:16,620,646: the fresh oracles below start from 1e-6)
        s.initialize(x0, ui); cost = s.solve(x0)
        tc, ta, tl = s.trace()
        assert s.adopt_mismatches() == 0
        for b in range(Bs):
            ob = oracle_for(prob, jac_mode=0, early_exit=0, max_iter=3); ob.set_contact_mode(cm); ob.set_joint_limits(True)
            ob.initialize(x0[b], ui[b]); ok, c = ob.solve(x0[b])
            nn, oc, oa, ol_ = ob.trace()
            assert nn == 3 and np.allclose(tc[b], oc, rtol=1e-5) and np.array_equal(ta[b], oa), (cm, tc[b], oc, ta[b], oa)
            assert abs(cost[b] - c) <= 1e-5 * abs(c) and rel(s.gains_K()[b], ob.get("K")) < 1e-5 and rel(s.xbar()[b], ob.get("xbar")) < 1e-5
        s.close()
    # (iv) analytic Jacobians on the committed states against the oracle's forward-mode AD through the same rows: a stopped hinge is an
    # acceleration-prescribed joint of the dumped recursion, d qacc_i = -1 / h rides the direction of its own rate
    N = 25
    checked = 0
    for cm in (0, 2):
        for stance in ((1, 1), (1, 0)):
            ids = [i for i in range(n) if int(g["contact"][i]) == cm and tuple(int(v) for v in g["stance"][i]) == stance]
            if not ids:
                continue
            prob, x0, ui = make(len(ids), seed=28, gravity=list(g["gravity"]), walking=True)
            prob["stance"] = np.ones_like(prob["stance"]) * np.array(stance, dtype=prob["stance"].dtype)
            s = _solver(len(ids)); s.set_problem(prob); s.set_contact_mode(cm, float(g["soft"])); s.set_joint_limits(True); s.set_options(jacobian_mode=0)
            s.initialize(x0, ui)
            X = np.repeat(g["x"][ids][:, None, :], N + 1, axis=1); U = np.repeat(g["u"][ids][:, None, :], N, axis=1)
            s.set_trajectory(X, U); s.stage_linearize()
            A, Bm = s.linearization()
            for k, i in enumerate(ids):
                o = oracle_for(prob, jac_mode=0); o.set_contact_mode(cm, float(g["soft"])); o.set_joint_limits(True)
                o.set_trajectory(X[k], U[k]); o.linearize()
                Ao, Bo = o.get("A")[0], o.get("B")[0]
                assert np.abs(A[k][0] - Ao).max() < 1e-8 * max(1.0, np.abs(Ao).max()), (cm, i, np.abs(A[k][0] - Ao).max())
                assert np.abs(Bm[k][0] - Bo).max() < 1e-8 * max(1.0, np.abs(Bo).max())
                for j in np.flatnonzero(g["lock"][i]):
                    assert np.abs(A[k][0][32 + j]).max() < 1e-12 and np.abs(Bm[k][0][32 + j]).max() < 1e-12     # v_j+ = 0 whatever moves
                    e = np.zeros(51); e[7 + j] = 1.0
                    assert np.abs(A[k][0][7 + j] - e).max() < 1e-12                                                # q_j+ = q_j
                checked += 1
            s.close()
    assert checked == n


def test_joint_limit_rows_with_restoring_stiffness_match_golden_and_oracle():
    """Round 6: ilqr_hip_set_joint_limit_stiffness (include/ilqr_hip.h) -- the joint-limit rows prescribe MuJoCo's constraint reference
    acceleration in its hard limit, qacc_i = -v_i / h - k r_i, so a hinge outside its range is pushed back (v_i+ = -h k r_i: mj_step pushes
    it back too, robot_utils.cpp:113-114; the pure stop of round 5 does not).  (i) the step against the dense NumPy KKT system with explicit
    rows (tests/golden/joint_limit_stiffness_golden.npz, k = 1 / (2 h)^2 = 625) and against the oracle, constraint-free plant and unilateral
    stance, none / one / several hinges constrained, hinges that drift back in too slowly among them; k = 0 afterwards is the round-5 step
    bit for bit; (ii) solves with forward-difference and with analytic Jacobians from states with a knee and an elbow past their ranges
    against the oracle; the nominal re-rollout reproduces the accepted candidate (adopt_mismatches == 0); (iii) the analytic Jacobians
    against the oracle's forward-mode AD: the constrained hinge's velocity row is -h k e_theta (its own angle) and nothing else."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "joint_limit_stiffness_golden.npz"))
    g0 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "joint_limit_golden.npz"))
    n, k, h = len(g["x"]), float(g["stiffness"]), float(g["h"])
    prob, x0, ui = make(2, seed=31, gravity=list(g["gravity"]), walking=True)
    s = _solver(2); s.set_problem(prob)
    o = oracle_for(prob)
    constrained = 0
    for i in range(n):
        cm = int(g["contact"][i])
        xs = np.tile(g["x"][i], (2, 1)); us = np.tile(g["u"][i], (2, 1))
        s.set_contact_mode(cm, float(g["soft"])); s.set_joint_limits(True); s.set_joint_limit_stiffness(k)
        got = s.step_stance(xs, us, int(g["stance"][i][0]), int(g["stance"][i][1]))
        o.set_contact_mode(cm, float(g["soft"])); o.set_joint_limits(True); o.set_joint_limit_stiffness(k)
        want = o.step_stance(g["x"][i], g["u"][i], g["stance"][i])
        assert np.array_equal(got[0], got[1])
        assert np.abs(got[0] - g["x_next"][i]).max() < 1e-9 * max(1.0, np.abs(g["x_next"][i]).max()), (i, np.abs(got[0] - g["x_next"][i]).max())
        assert np.abs(got[0] - want).max() < 1e-9 * max(1.0, np.abs(want).max())
        for j in np.flatnonzero(g["lock"][i]):
            r = g["x"][i][7 + j] - (g["jrange"][j, 1] if g["x"][i][7 + j] > g["jrange"][j, 1] else g["jrange"][j, 0])
            assert abs(got[0][32 + j] + h * k * r) < 1e-10
        constrained += int(g["lock"][i].any())
    assert constrained >= 8
    # k = 0 again: the round-5 step, bit for bit, and its golden
    for i in range(len(g0["x"])):
        cm = int(g0["contact"][i])
        xs = np.tile(g0["x"][i], (2, 1)); us = np.tile(g0["u"][i], (2, 1))
        s.set_contact_mode(cm, float(g0["soft"])); s.set_joint_limits(True); s.set_joint_limit_stiffness(0.0)
        got = s.step_stance(xs, us, int(g0["stance"][i][0]), int(g0["stance"][i][1]))
        assert np.abs(got[0] - g0["x_next"][i]).max() < 1e-9 * max(1.0, np.abs(g0["x_next"][i]).max())
        for j in np.flatnonzero(g0["lock"][i]):
            assert abs(got[0][32 + j]) < 1e-12
    s.close()
    # (ii) solves: left knee (hinge 3) past its upper limit and moving out, right elbow (hinge 18) past its lower limit moving back in slowly
    Bs = 3
    for cm in (0, 2):
        prob, x0, ui = make(Bs, seed=33, gravity=list(g["gravity"]) if cm else None, walking=True)
        x0 = x0.copy()
        x0[:, 7 + 3] = 2.09; x0[:, 32 + 3] = 1.5
        x0[1:, 7 + 18] = -1.29; x0[1:, 32 + 18] = 0.1
        for jm, ktol in ((1, 3e-4), (0, 1e-5)):      # (forward differences: round-off / eps of two independent FD evaluations through three iterations, see test_fixed_iteration_mode_and_fd_mode_parity)
            s = _solver(Bs); s.set_problem(prob); s.set_contact_mode(cm); s.set_joint_limits(True); s.set_joint_limit_stiffness(k)
            s.set_options(jacobian_mode=jm, fd_eps=1e-5, early_exit=False); s.set_max_iterations(3)
            s.initialize(x0, ui); cost = s.solve(x0)
            tc, ta, tl = s.trace()
            assert s.adopt_mismatches() == 0
            differs = 0
            for b in range(Bs):
                ob = oracle_for(prob, jac_mode=jm, fd_eps=1e-5, early_exit=0, max_iter=3); ob.set_contact_mode(cm); ob.set_joint_limits(True); ob.set_joint_limit_stiffness(k)
                ob.initialize(x0[b], ui[b]); ok, c = ob.solve(x0[b])
                nn, oc, oa, ol_ = ob.trace()
                assert nn == 3 and np.allclose(tc[b], oc, rtol=1e-5) and np.array_equal(ta[b], oa), (cm, jm, tc[b], oc, ta[b], oa)
                assert abs(cost[b] - c) <= 1e-5 * abs(c) and rel(s.gains_K()[b], ob.get("K")) < ktol and rel(s.xbar()[b], ob.get("xbar")) < 1e-5
                o2 = oracle_for(prob, jac_mode=jm, fd_eps=1e-5, early_exit=0, max_iter=3); o2.set_contact_mode(cm); o2.set_joint_limits(True)
                o2.initialize(x0[b], ui[b]); _, c2 = o2.solve(x0[b])
                differs += int(abs(c2 - c) > 1e-6 * abs(c))
            assert differs >= 2                              # the restoring term mattered (against the pure stop)
            s.close()
    # (iii) analytic Jacobians on the committed states against the oracle's forward-mode AD
    N = 25
    checked = 0
    for cm in (0, 2):
        for stance in ((1, 1), (1, 0)):
            ids = [i for i in range(n) if int(g["contact"][i]) == cm and tuple(int(v) for v in g["stance"][i]) == stance]
            if not ids:
                continue
            prob, x0, ui = make(len(ids), seed=28, gravity=list(g["gravity"]), walking=True)
            prob["stance"] = np.ones_like(prob["stance"]) * np.array(stance, dtype=prob["stance"].dtype)
            s = _solver(len(ids)); s.set_problem(prob); s.set_contact_mode(cm, float(g["soft"])); s.set_joint_limits(True); s.set_joint_limit_stiffness(k); s.set_options(jacobian_mode=0)
            s.initialize(x0, ui)
            X = np.repeat(g["x"][ids][:, None, :], N + 1, axis=1); U = np.repeat(g["u"][ids][:, None, :], N, axis=1)
            s.set_trajectory(X, U); s.stage_linearize()
            A, Bm = s.linearization()
            for kk, i in enumerate(ids):
                o = oracle_for(prob, jac_mode=0); o.set_contact_mode(cm, float(g["soft"])); o.set_joint_limits(True); o.set_joint_limit_stiffness(k)
                o.set_trajectory(X[kk], U[kk]); o.linearize()
                Ao, Bo = o.get("A")[0], o.get("B")[0]
                assert np.abs(A[kk][0] - Ao).max() < 1e-8 * max(1.0, np.abs(Ao).max()), (cm, i, np.abs(A[kk][0] - Ao).max())
                assert np.abs(Bm[kk][0] - Bo).max() < 1e-8 * max(1.0, np.abs(Bo).max())
                for j in np.flatnonzero(g["lock"][i]):
                    e = np.zeros(51); e[7 + j] = -h * k                 # v_j+ = -h k (theta_j - limit): its own angle and nothing else
                    assert np.abs(A[kk][0][32 + j] - e).max() < 1e-10 and np.abs(Bm[kk][0][32 + j]).max() < 1e-12
                    e2 = np.zeros(51); e2[7 + j] = 1.0 - h * h * k        # q_j+ = q_j + h v_j+
                    assert np.abs(A[kk][0][7 + j] - e2).max() < 1e-10
                checked += 1
            s.close()
    assert checked == n


def test_forward_difference_jacobians_two_lane_vs_scalar_kernels():
    """The forward-difference Jacobians (the reference's scheme, robot_utils.cpp:120-160) on the two-lane step kernels equal the
    scalar kernels' (ILQR_DYN=s) to rounding / eps, with and without stance constraints."""
    B = 3
    for contact in (0, 2):
        prob, x0, ui = make(B, seed=19, gravity=[0.0, 0.0, -9.81] if contact else None, walking=True)
        out = []
        for dyn in (None, "s"):
            old = os.environ.get("ILQR_DYN")
            if dyn: os.environ["ILQR_DYN"] = dyn
            try:
                s = _solver(B); s.set_problem(prob); s.set_contact_mode(contact); s.set_options(jacobian_mode=1, fd_eps=1e-5)
                s.initialize(x0, ui); s.stage_linearize()
                out.append(s.linearization()); s.close()
            finally:
                if dyn:
                    if old is None: os.environ.pop("ILQR_DYN", None)
                    else: os.environ["ILQR_DYN"] = old
        assert np.abs(out[0][0] - out[1][0]).max() < 2e-5 and np.abs(out[0][1] - out[1][1]).max() < 2e-5


def test_warm_start_mpc_step_and_control_law():
    from mpc_ilqr_mujoco_amd import solver as sv
    B = 2
    prob, x0, ui = make(B, seed=7)
    s = _solver(B); s.set_problem(prob); s.set_max_iterations(3)
    mpc = sv.BatchedMPC(s, lambda t: (prob["x_ref"], prob["u_ref"], prob["com_ref"]))
    os_ = [oracle_for(prob, max_iter=3) for _ in range(B)]
    x = x0.copy()
    prev = [None] * B
    for step in range(3):
        u = mpc.step_once(x)
        xb, ub = s.xbar(), s.ubar()
        for b in range(B):
            o = os_[b]
            if prev[b] is None:
                o.initialize(x[b])       # gravity-compensation cold start
            else:
                o.initialize(x[b], None, prev[b][0], prev[b][1])
            o.solve(x[b])
            prev[b] = (o.get("xbar"), o.get("ubar"))
            assert rel(u[b], o.compute_control(x[b])) < 1e-5
            assert rel(xb[b], prev[b][0]) < 1e-5 and rel(ub[b], prev[b][1]) < 1e-5
        x = s.step(x, u)                 # plant = same smooth dynamics
    xm = x + 1e-3
    uc = s.compute_control(xm)
    Kk, xb, ub = s.gains_K(), s.xbar(), s.ubar()
    for b in range(B):
        assert np.allclose(uc[b], ub[b, 0] + Kk[b, 0] @ (xm[b] - xb[b, 0]), rtol=1e-12, atol=1e-12)
    s.close()


def test_mpc_on_loaded_walking_references_matches_oracle():
    """references + contact schedule from the reference's data files (excerpt in tests/golden),
    reference windows with the horizon-local contact index, warm-started MPC steps, GPU vs oracle."""
    from mpc_ilqr_mujoco_amd import references as rf
    from mpc_ilqr_mujoco_amd import solver as sv
    r = np.load(os.path.join(G, "refdata_golden.npz"))
    rd = rf.ReferenceData(sv.reference_kinematics, sv.reference_com_velocity)
    rd.set_states(np.concatenate([r["q_ref2_mj"], r["v_ref2"]], axis=1))
    rd.contact = r["contact_walking"].astype(np.int32)
    B, N = 2, 25
    cfg = dict(sc.SHIPPED_CONFIG); cfg["W_com_vel"] = 2.0
    base = sc.make_problem(sv.reference_kinematics, N=N, cfg=cfg, gravity=(0.0, 0.0, -2.0))
    s = _solver(B); s.set_max_iterations(3)
    rng = np.random.default_rng(11)
    x = np.repeat(rd.x_ref[:1], B, axis=0); x[:, 7:26] += rng.uniform(-0.01, 0.01, (B, 19))
    prev = [None] * B
    os_ = [ol.Oracle(N, base["dt"]) for _ in range(B)]
    for step in range(2):
        prob = rd.problem_at(step, N, base)
        assert prob["stance"].min() == 0          # the excerpt contains swing phases
        s.set_problem(prob)
        if step == 0:
            ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
            s.initialize(x, np.repeat(np.tile(ug, (N, 1))[None], B, axis=0))
        else:
            s.initialize_warm_resident(x)
        s.solve(x)
        xb, ub, K = s.xbar(), s.ubar(), s.gains_K()
        tr_cost = s.trace()[0]
        for b in range(B):
            o = os_[b]; o.set_problem(prob); o.set_options(max_iter=3)
            if prev[b] is None:
                o.initialize(x[b], np.tile(ug, (N, 1)))      # same cold-start controls as the GPU handle
            else:
                o.initialize(x[b], None, prev[b][0], prev[b][1])
            o.solve(x[b])
            prev[b] = (o.get("xbar"), o.get("ubar"))
            assert rel(xb[b], prev[b][0]) < 1e-5 and rel(ub[b], prev[b][1]) < 1e-5 and rel(K[b], o.get("K")) < 1e-5
            it, ocost, _, _ = o.trace()
            assert np.allclose(tr_cost[b][:it + 1], ocost[:it + 1], rtol=1e-5)
        x = s.step(x, s.compute_control(x))
    s.close()


def test_single_rollout_and_long_horizon_configs():
    """BASELINE.json configs[0] (one trajectory, shipped config, N = 25) and the N = 50 contact-scheduled walking
    window of configs[4] (references and contact schedule through the loader), each against the oracle."""
    from mpc_ilqr_mujoco_amd import references as rf
    from mpc_ilqr_mujoco_amd import solver as sv
    # --- configs[0]: B = 1
    prob, x0, ui = make(1, seed=21)
    s = _solver(1); s.set_problem(prob); s.initialize(x0, ui); cost = s.solve(x0)
    o = oracle_for(prob); o.initialize(x0[0], ui[0]); ok, c = o.solve(x0[0])
    n, oc, oa, _ = o.trace()
    assert n == s.iterations()[0] and np.allclose(s.trace()[0][0, : n + 1], oc[: n + 1], rtol=1e-5) and abs(cost[0] - c) <= 1e-5 * abs(c)
    assert rel(s.gains_K()[0], o.get("K")) < 1e-5
    s.close()
    # --- configs[4]: N = 50, walking references + contact schedule
    r = np.load(os.path.join(G, "refdata_golden.npz"))
    rd = rf.ReferenceData(sv.reference_kinematics, sv.reference_com_velocity)
    rd.set_states(np.concatenate([r["q_ref2_mj"], r["v_ref2"]], axis=1)); rd.contact = r["contact_walking"].astype(np.int32)
    B, N = 3, 50
    base = sc.make_problem(sv.reference_kinematics, N=N, gravity=(0.0, 0.0, -2.0))
    prob = rd.problem_at(4, N, base)
    assert prob["x_ref"].shape == (1, N + 1, 51) and prob["stance"].min() == 0
    rng = np.random.default_rng(5)
    x0 = np.repeat(rd.x_ref[4:5], B, axis=0); x0[:, 7:26] += rng.uniform(-0.02, 0.02, (B, 19)); x0[:, 26:] *= 0.2
    ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
    ui = np.repeat(np.tile(ug, (N, 1))[None], B, axis=0) + rng.uniform(-0.5, 0.5, (B, N, 19))
    s = _solver(B, N=N); s.set_problem(prob); s.set_max_iterations(4); s.initialize(x0, ui); cost = s.solve(x0)
    tc = s.trace()[0]
    for b in range(B):
        o = oracle_for(prob, max_iter=4); o.initialize(x0[b], ui[b]); ok, c = o.solve(x0[b])
        n, oc, oa, _ = o.trace()
        assert n == s.iterations()[b] and np.allclose(tc[b, : n + 1], oc[: n + 1], rtol=1e-5), (b, tc[b], oc)
        assert rel(s.gains_K()[b], o.get("K")) < 1e-5 and rel(s.xbar()[b], o.get("xbar")) < 1e-5
    s.close()


def test_closed_loop_runner_with_logs(tmp_path):
    """closed-loop batched MPC on loaded references; the logged first knot / control equal the solver state."""
    from mpc_ilqr_mujoco_amd import mpc_loop as ml
    from mpc_ilqr_mujoco_amd import references as rf
    from mpc_ilqr_mujoco_amd import solver as sv
    r = np.load(os.path.join(G, "refdata_golden.npz"))
    rd = rf.ReferenceData(sv.reference_kinematics, sv.reference_com_velocity)
    rd.set_states(np.concatenate([r["q_ref2_mj"], r["v_ref2"]], axis=1)); rd.contact = r["contact_walking"].astype(np.int32)
    B, N = 3, 25
    base = sc.make_problem(sv.reference_kinematics, N=N, gravity=(0.0, 0.0, -2.0))
    s = _solver(B); s.set_max_iterations(2)
    run = ml.MPCRunner(s, rd, base, log_dir=str(tmp_path), log_rollouts=(0, 2), profile_stages=True)
    x0 = np.repeat(rd.x_ref[:1], B, axis=0); x0[:, 26:] *= 0.1
    ug = sv.gravity_compensation(sc.standing_state(), base["gravity"])
    xs, us = run.run(x0, 3, u_init=np.repeat(np.tile(ug, (N, 1))[None], B, axis=0))
    run.close()
    table = run.profiling_table()      # the reference's profiler keys and table layout (main/humanoid_mpc.cpp:195-226)
    for key in ("MPC_stepOnce", "MPC_extractReference", "MPC_warmStart", "MPC_iLQR_solve", "MPC_computeControl", "iLQR_backwardPass", "iLQR_lineSearch", "iLQR_linearization"):
        row = [l for l in table.splitlines() if l.startswith(key)]
        assert len(row) == 1 and int(row[0].split()[1]) == 3, key
    assert xs.shape == (4, B, 51) and us.shape == (3, B, 19) and np.all(np.isfinite(xs))
    assert np.allclose(xs[1], s.step(xs[0], us[0]), rtol=0, atol=1e-12)        # the plant is the model's own step
    for b in (0, 2):
        rows = np.loadtxt(os.path.join(str(tmp_path), "rollout_%d" % b, "mpc_log.csv"), delimiter=",", skiprows=1)
        assert rows.shape == (3, 4 + 2 * 70) and np.array_equal(rows[:, 0], [1, 2, 3])
        assert np.allclose(rows[:, 4:55], xs[:3, b], rtol=1e-5, atol=1e-6) and np.allclose(rows[:, 55:74], us[:, b], rtol=1e-5, atol=1e-6)
        assert np.allclose(rows[2, 74:125], rd.x_ref[2], rtol=1e-5, atol=1e-6)
        q = np.loadtxt(os.path.join(str(tmp_path), "rollout_%d" % b, "q_optimal.csv"), delimiter=",", skiprows=1)
        assert q.shape == (3, 28) and np.allclose(q[:, 2:], xs[:3, b, :26], rtol=1e-5, atol=1e-6)   # xbar[0] == x_measured after a solve
    s.close()


def test_closed_loop_standing_under_physical_gravity_needs_the_contact_row(tmp_path):
    """with the constraint-free plant a robot under -9.81 gravity is in free fall (the balance term turns
    NaN within the horizon); with the scheduled feet held in stance the same closed loop stands."""
    from mpc_ilqr_mujoco_amd import mpc_loop as ml
    from mpc_ilqr_mujoco_amd import references as rf
    from mpc_ilqr_mujoco_amd import solver as sv
    B, N = 2, 25
    base = sc.make_problem(sv.reference_kinematics, N=N, gravity=(0.0, 0.0, -9.81))
    rd = rf.ReferenceData(sv.reference_kinematics, sv.reference_com_velocity)
    rd.set_states(np.tile(sc.standing_state(), (60, 1))); rd.contact = np.ones((60, 2), dtype=np.int32)
    ug = sv.gravity_compensation(sc.standing_state(), base["gravity"])
    x0, ui = sc.synthetic_batch(B, N, 0, ug)
    s = _solver(B); s.set_max_iterations(3); s.set_contact_mode(1)
    run = ml.MPCRunner(s, rd, base)
    xs, us = run.run(x0, 8, u_init=ui)
    assert np.all(np.isfinite(xs)) and np.all(np.isfinite(us))
    assert xs[:, :, 2].min() > 0.97 and np.abs(xs[-1, :, 26:]).max() < 5.0      # pelvis height kept, no runaway velocity
    com0, ee0 = sv.reference_kinematics(xs[0, 0]); com8, ee8 = sv.reference_kinematics(xs[-1, 0])
    assert np.abs(ee8 - ee0).max() < 5e-3                                        # the stance feet stayed where they were
    s.close()
    s = _solver(B); s.set_max_iterations(3)
    free = ml.MPCRunner(s, rd, base).run(x0, 8, u_init=ui)[0]
    assert (not np.all(np.isfinite(free))) or free[-1, :, 2].max() < xs[-1, :, 2].min() - 0.1
    s.close()


def test_closed_loop_with_the_coulomb_limit_stands_on_grip_and_slides_on_ice(tmp_path):
    """Contact mode 3 in the closed loop (plant and solver on the friction-limited stance row, the reference's forward-difference
    Jacobians).  A robot standing still under physical gravity loads its feet almost along the normal: with MuJoCo's default friction
    the plant's feet stay where they are, as in the unilateral mode.  Pushed sideways at 0.6 m/s, the rigid rows of mode 2 hold the
    feet whatever it takes (infinite friction); on mu = 0.02 the same push makes them slide -- along the floor: the normal row still
    holds them down.  (tools/probes/friction_loop_probe.py prints the figures.)"""
    from mpc_ilqr_mujoco_amd import mpc_loop as ml
    from mpc_ilqr_mujoco_amd import references as rf
    from mpc_ilqr_mujoco_amd import solver as sv
    B, N = 2, 25
    base = sc.make_problem(sv.reference_kinematics, N=N, gravity=(0.0, 0.0, -9.81))
    rd = rf.ReferenceData(sv.reference_kinematics, sv.reference_com_velocity)
    rd.set_states(np.tile(sc.standing_state(), (60, 1))); rd.contact = np.ones((60, 2), dtype=np.int32)
    ug = sv.gravity_compensation(sc.standing_state(), base["gravity"])
    x_still = np.tile(sc.standing_state(), (B, 1)); u_still = np.tile(ug, (B, N, 1))
    out = {}
    for mode, mu, push in ((2, 1.0, 0.0), (3, 1.0, 0.0), (2, 1.0, 0.6), (3, 0.02, 0.6), (4, 1.0, 0.6), (4, 0.02, 0.6)):
        s = _solver(B); s.set_max_iterations(3); s.set_contact_mode(mode); s.set_friction(mu); s.set_options(jacobian_mode=1, fd_eps=1e-5)
        xp = x_still.copy(); xp[:, 27] += push                                  # a sideways velocity of the pelvis
        xs, us = ml.MPCRunner(s, rd, base).run(xp, 6, u_init=u_still)
        assert np.all(np.isfinite(xs)) and np.all(np.isfinite(us))
        ee0 = sv.reference_kinematics(xs[0, 0])[1]; ee1 = sv.reference_kinematics(xs[-1, 0])[1]
        out[(mode, push) if mode != 4 else (mode, mu)] = (xs, np.abs(ee1[:, :2] - ee0[:, :2]).max(), np.abs(ee1[:, 2] - ee0[:, 2]).max())
        s.close()
    # standing still the plant's feet carry f_t / f_n = 0.06: no foot slides, in either mode (the solver's line-search candidates do
    # leave the cone, so the two closed loops are not the same trajectory)
    for m in (2, 3):
        assert out[(m, 0.0)][0][:, :, 2].min() > 0.97 and out[(m, 0.0)][1] < 5e-3 and out[(m, 0.0)][2] < 5e-3, (m, out[(m, 0.0)][1:])
    assert out[(2, 0.6)][1] < 5e-3                                                               # rigid rows: the push does not move the feet
    assert out[(3, 0.6)][1] > 2e-2 and out[(3, 0.6)][2] < 3e-2, out[(3, 0.6)][1:]                # ice: they slide, along the floor
    # mode 4 keeps kinetic friction on a sliding foot: the friction coefficient now matters -- less slip on grip than on ice
    assert np.all(np.isfinite(out[(4, 1.0)][0])) and out[(4, 1.0)][1] < 0.7 * out[(4, 0.02)][1], (out[(4, 1.0)][1:], out[(4, 0.02)][1:])


def test_launch_orchestration_variants_are_bitwise_equivalent():
    """The solve may be enqueued as contiguous batch slices on separate streams (ILQR_SLICES) and the nominal re-rollout
    of iterations >= 1 may run beside the linearisation into a shadow buffer (ILQR_OVERLAP_ROLLOUT): the kernels and the
    data they see are the same, so costs, traces and gains must not change by a single bit -- except that with the
    sequential rollout the linearisation reads the re-rolled trajectory instead of the accepted candidate (1e-12)."""
    B = 200        # not a multiple of the slice count
    prob, x0, ui = make(B, seed=12)
    out = {}
    for key, env in (("base", {}), ("slices", {"ILQR_SLICES": "3"}), ("seq", {"ILQR_OVERLAP_ROLLOUT": "0"})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            s = _solver(B); s.set_problem(prob); s.set_max_iterations(4)
            s.initialize(x0, ui)
            cost = s.solve(x0)
            out[key] = (cost, s.trace()[0], s.gains_K(), s.iterations())
            s.close()
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
    for a, b in zip(out["base"], out["slices"]):
        assert np.array_equal(a, b, equal_nan=True)
    assert np.array_equal(out["base"][3], out["seq"][3])
    assert np.allclose(out["base"][0], out["seq"][0], rtol=1e-9) and rel(out["base"][2], out["seq"][2]) < 1e-6
    # Round 3: after a COLD start (the nominal trajectory is itself a rollout by the same kernel) iteration 0 re-rolls it beside the
    # linearisation too.  solve(x0) hands over a new x0 and therefore rolls out first; solve() without one takes the concurrent
    # path -- the re-rollout reproduces the cold start bit for bit, so nothing may change; a warm start always rolls out first.
    s = _solver(B); s.set_problem(prob); s.set_max_iterations(4)
    s.initialize(x0, ui); cost = s.solve()
    got = (cost, s.trace()[0], s.gains_K(), s.iterations())
    assert s.adopt_mismatches() == 0
    for a, b in zip(out["base"], got):
        assert np.array_equal(a, b, equal_nan=True)
    s.set_regularization(1e-6); s.initialize_warm_resident(x0); c_async = s.solve()
    K_async = s.gains_K()
    s2 = _solver(B); s2.set_problem(prob); s2.set_max_iterations(4)
    s2.initialize(x0, ui); s2.solve(x0); s2.set_regularization(1e-6); s2.initialize_warm_resident(x0); c_seq = s2.solve(x0)
    assert np.array_equal(c_async, c_seq) and np.array_equal(K_async, s2.gains_K())
    s.close(); s2.close()


def test_per_rollout_reference_sets():
    B, N = 3, 25
    prob, x0, ui = make(B, seed=8)
    # give every rollout its own reference window / contact schedule
    rng = np.random.default_rng(0)
    for k in ("x_ref", "u_ref", "com_ref", "ee_ref", "com_vel_ref", "stance"):
        prob[k] = np.repeat(prob[k], B, axis=0)
    prob["x_ref"][:, :, 0] += rng.uniform(-0.02, 0.02, (B, 1))
    prob["com_ref"][:, :, 0] += rng.uniform(-0.02, 0.02, (B, 1))
    prob["stance"][1, 5:10, 0] = 0
    s = _solver(B); s.set_problem(prob); s.set_max_iterations(2)
    s.initialize(x0, ui); cost = s.solve(x0)
    for b in range(B):
        o = oracle_for(prob, b=b, max_iter=2)
        o.initialize(x0[b], ui[b]); ok, c = o.solve(x0[b])
        assert abs(cost[b] - c) <= 1e-5 * abs(c) and rel(s.gains_K()[b], o.get("K")) < 1e-5
    s.close()


def test_bench_size_properties():
    """B = 4096 (BASELINE.json configs[2]): size-independent properties instead of a full oracle run."""
    B = 4096
    prob, x0, ui = make(B, seed=0)
    s = _solver(B); s.set_problem(prob)
    s.initialize(x0, ui)
    cost = s.solve(x0)
    tc, ta, tl = s.trace()
    it = s.iterations()
    assert np.all(np.isfinite(cost)) and np.all(it >= 1) and np.all(it <= 10)
    # the line search only ever accepts strict decreases
    for b in range(0, B, 97):
        c = tc[b, : it[b] + 1]
        assert np.all(np.diff(c) <= 0)
    xb = s.xbar()
    assert np.abs(np.linalg.norm(xb[:, :, 3:7], axis=2) - 1).max() < 1e-12   # quaternions stay normalised
    assert np.array_equal(xb[:, 0], x0)
    # batch invariance: any rollout solved alone gives bit-identical results (no cross-rollout coupling)
    pick = [0, 1, 777, 2048, 4095]
    Kfull = s.gains_K()
    K_all = Kfull[pick]; K0_all = Kfull[:, 0].copy(); ub_all = s.ubar()
    del Kfull
    s.close()
    s2 = _solver(len(pick)); s2.set_problem(prob)
    s2.initialize(x0[pick], ui[pick]); c2 = s2.solve(x0[pick])
    assert np.array_equal(c2, cost[pick]) and np.array_equal(s2.gains_K(), K_all)
    s2.close()
    # spot-check against the oracle: 32 rollouts spread over the batch through its OpenMP batch (final cost, iteration count,
    # first control, first-knot gain)
    spot = [int(i) for i in np.linspace(0, B - 1, 32)]
    o = oracle_for(prob)
    _, oc, oit, ou0, oK0 = o.batch_solve(x0[spot], ui[spot], nthreads=0, want_gains=True)
    assert np.array_equal(oit, it[spot]) and np.allclose(oc, cost[spot], rtol=1e-5, atol=0)
    assert rel(ub_all[spot, 0], ou0) < 1e-5 and rel(K0_all[spot], oK0) < 1e-5


def test_bench_contract_json_line():
    """bench.py prints ONE JSON line with the fields the driver and the judge read (task statement, section 4)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--batch", "64", "--steps", "1", "--warmup", "0", "--cpu-seconds", "1"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 0 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["value"] > 0 and d["value"] > 0
