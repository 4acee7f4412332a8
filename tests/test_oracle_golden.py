"""Pin the CPU oracle (oracle/) against the committed independent goldens (tests/golden/).
No GPU needed.  See tests/golden/gen_golden.py for how each fixture was produced."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from conftest import load_package

G = os.path.join(os.path.dirname(__file__), "golden")
sc = load_package().scenario


def test_dynamics_step_vs_kane_golden():
    d = np.load(os.path.join(G, "dynamics_golden.npz"))
    o = ol.Oracle(25, float(d["h"]))
    for x, u, g, xn in zip(d["x"], d["u"], d["gravity"], d["x_next"]):
        prob = sc.make_problem(ol.reference_kinematics, gravity=g)
        o.set_problem(prob)
        got = o.step(x, u)
        assert np.allclose(got, xn, rtol=0, atol=1e-10), np.abs(got - xn).max()


def test_aba_rnea_identity_and_mass_matrix():
    rng = np.random.default_rng(0)
    g = np.array([0, 0, -9.81])
    for _ in range(5):
        x = sc.standing_state()
        x[7:26] = rng.uniform(-0.5, 0.5, 19)
        q = rng.normal(size=4)
        x[3:7] = q / np.linalg.norm(q)
        x[26:] = rng.uniform(-2, 2, 25)
        tau = rng.uniform(-30, 30, 19)
        qacc = ol.forward_dynamics(x, tau, 0.12, g)
        back = ol.inverse_dynamics(x, qacc, 0.12, g)
        assert np.abs(back[:6]).max() < 1e-9
        assert np.abs(back[6:] - tau).max() < 1e-9
    x0 = x.copy()
    x0[26:] = 0
    bias = ol.inverse_dynamics(x0, np.zeros(25), 0.0, g)
    M = np.array([ol.inverse_dynamics(x0, np.eye(25)[i], 0.0, g) - bias for i in range(25)]).T
    assert np.abs(M - M.T).max() < 1e-10
    assert np.linalg.eigvalsh(0.5 * (M + M.T)).min() > 0
    assert abs(M[0, 0] - 51.649896) < 1e-6 and abs(M[2, 2] - 51.649896) < 1e-6  # total MJCF mass


def test_energy_conservation_without_damping_and_gravity_momentum():
    # free flight, zero gravity, zero torque: v'_lin of the CoM stays constant (momentum conservation)
    o = ol.Oracle(25, 0.002)
    prob = sc.make_problem(ol.reference_kinematics, gravity=(0.0, 0.0, 0.0))
    o.set_problem(prob)
    rng = np.random.default_rng(1)
    x = sc.standing_state()
    x[26:] = rng.uniform(-0.5, 0.5, 25)
    # damping torque is internal -> linear momentum of the whole robot is conserved exactly in continuous time
    com0, _ = ol.reference_kinematics(x)
    xs = [x]
    for _ in range(50):
        xs.append(o.step(xs[-1], np.zeros(19)))
    coms = np.array([ol.reference_kinematics(s)[0] for s in xs])
    vel = np.diff(coms, axis=0) / 0.002
    assert np.abs(vel - vel[0]).max() < 5e-3 * max(1.0, np.abs(vel[0]).max())


def test_state_conventions_against_reference_data_files():
    """Known-answer test on the reference's own data files:
    data/v_ref2.csv == differentiatePos(data/q_ref2_mj.csv, 0.02) with world-frame linear velocity,
    BODY-frame angular velocity (log of q_t^-1 (x) q_{t+1}) and plain joint differences; and
    q_ref2_mj is q_ref2_pin with the quaternion reordered xyzw -> wxyz."""
    d = np.load(os.path.join(G, "refdata_golden.npz"))
    q, v, dt = d["q_ref2_mj"], d["v_ref2"], float(d["dt"])
    assert np.array_equal(d["q_ref2_pin"][:, [6, 3, 4, 5]], q[:, 3:7])
    for t in range(q.shape[0] - 1):
        qa, qb = q[t, 3:7], q[t + 1, 3:7]
        ca = qa * np.array([1, -1, -1, -1])
        dq = np.concatenate([[ca[0] * qb[0] - ca[1:] @ qb[1:]], ca[0] * qb[1:] + qb[0] * ca[1:] + np.cross(ca[1:], qb[1:])])
        s = np.linalg.norm(dq[1:])
        speed = 2 * np.arctan2(s, dq[0])
        if speed > np.pi:
            speed -= 2 * np.pi
        w = (dq[1:] / s if s > 1e-15 else np.zeros(3)) * speed / dt
        assert np.abs((q[t + 1, :3] - q[t, :3]) / dt - v[t, :3]).max() < 1e-9
        assert np.abs((q[t + 1, 7:] - q[t, 7:]) / dt - v[t, 6:]).max() < 1e-9
        assert np.abs(w - v[t, 3:6]).max() < 1e-9
        # and the oracle's integrator inverts it: quat_t (x) exp(dt * w) == quat_{t+1} (up to the CSV's 6 digits)
        x = np.zeros(51); x[:26] = q[t]; x[26:] = v[t]
        ang = np.linalg.norm(w) * dt
        e = np.array([1.0, 0, 0, 0]) if ang < 1e-14 else np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * w / np.linalg.norm(w)])
        a = qa / np.linalg.norm(qa)
        r = np.concatenate([[a[0] * e[0] - a[1:] @ e[1:]], a[0] * e[1:] + e[0] * a[1:] + np.cross(a[1:], e[1:])])
        assert np.abs(r - qb / np.linalg.norm(qb)).max() < 1e-9


def _golden_problem(c, **weights):
    cfg = dict(sc.SHIPPED_CONFIG)
    cfg.update(W_com_pos=0.0, W_com_vel=0.0, W_foot=0.0, W_foot_vel=0.0, W_upright=0.0, w_balance=0.0)
    cfg.update(weights)
    prob = sc.make_problem(ol.reference_kinematics, cfg=cfg)
    # zero out tracking so only the task term remains
    prob["Q"] = np.zeros(51); prob["Qf"] = np.zeros(51); prob["R"] = np.zeros(19)
    prob["w_joint"] = 0.0; prob["w_ctrl"] = 0.0
    return prob


@pytest.mark.parametrize("mode", [0, 1])
def test_cost_terms_vs_torch_autograd_golden(mode):
    c = np.load(os.path.join(G, "cost_golden.npz"))
    o = ol.Oracle(25, 0.02)
    t = 3

    def check(prob, names):
        o.set_problem(prob)
        for i, x in enumerate(c["x"]):
            lx, lu, lxx, luu = o.knot_quadratics(t, x, np.zeros(19), mode)
            g = sum(c["grad_" + n][i] for n in names)
            H = sum(c["hess_" + n][i] for n in names)
            assert np.abs(lx - g).max() <= 1e-9 * max(1.0, np.abs(g).max()), (names, np.abs(lx - g).max())
            assert np.abs(lxx - H).max() <= 1e-9 * max(1.0, np.abs(H).max()), (names, np.abs(lxx - H).max())

    p = _golden_problem(c, W_com_pos=float(c["w_com"]))
    p["com_ref"][:] = c["ref_com"]
    check(p, ["com"])
    p = _golden_problem(c, W_com_vel=float(c["w_comvel"]))
    p["com_vel_ref"][:] = c["ref_comvel"]
    check(p, ["comvel"])
    p = _golden_problem(c, W_foot=float(c["w_eepos"]))
    p["stance"][:] = 0
    p["ee_ref"][0, :, 0] = c["ref_ee"]
    p["ee_ref"][0, :, 1] = c["ref_ee"] * np.array([1, -1, 1])
    check(p, ["eepos_L", "eepos_R"])
    p = _golden_problem(c, W_foot_vel=float(c["w_eevel"]))
    check(p, ["eevel_L", "eevel_R"])
    p["stance"][0, :, 1] = 0
    check(p, ["eevel_L"])
    p = _golden_problem(c, W_upright=float(c["w_upright"]))
    check(p, ["upright"])
    p = _golden_problem(c, w_balance=float(c["w_balance"]))
    # support point = mean of the foot refs (both stance)
    p["ee_ref"][0, :, 0, :2] = c["ref_ps"] + np.array([0.0, 0.1])
    p["ee_ref"][0, :, 1, :2] = c["ref_ps"] - np.array([0.0, 0.1])
    check(p, ["balance"])


def test_total_cost_and_reference_kinematics_vs_golden():
    c = np.load(os.path.join(G, "cost_golden.npz"))
    for i, x in enumerate(c["x"]):
        com, ee = ol.reference_kinematics(x)
        assert np.abs(com - c["com_mj"][i]).max() < 1e-12
        assert np.abs(ee - c["ee_mj"][i]).max() < 1e-12
    # computeTotalCost restated independently in numpy (ilqr.cpp:363-518) on a short random trajectory
    rng = np.random.default_rng(5)
    N = 4
    o = ol.Oracle(N, 0.02)
    prob = sc.make_problem(ol.reference_kinematics, N=N)
    o.set_problem(prob)
    xs = np.tile(sc.standing_state(), (N + 1, 1)) + 0.05 * rng.normal(size=(N + 1, 51))
    xs[:, 3:7] /= np.linalg.norm(xs[:, 3:7], axis=1, keepdims=True)
    xs[2, 7 + 3] = 2.0   # knee beyond its soft limit
    us = rng.uniform(-50, 50, size=(N, 19))
    us[1, 4] = 39.0      # ankle torque beyond the 10 % margin
    o.set_trajectory(xs, us)
    Q, R, Qf = prob["Q"], prob["R"], prob["Qf"]
    wu, wb = prob["task_weights"][4], prob["task_weights"][5]
    jr, cr = c["jrange"], c["ctrlrange"]

    def pen(val, rng_, w):
        lo = rng_[:, 0] + 0.1 * (rng_[:, 1] - rng_[:, 0]); hi = rng_[:, 1] - 0.1 * (rng_[:, 1] - rng_[:, 0])
        return w * (np.maximum(val - hi, 0) ** 2 + np.maximum(lo - val, 0) ** 2).sum()

    tot = 0.0
    for t in range(N + 1):
        x = xs[t]; e = x - prob["x_ref"][0, t]
        tot += 0.5 * e @ ((Qf if t == N else Q) * e)
        if t < N:
            eu = us[t] - prob["u_ref"][0, t]; tot += 0.5 * eu @ (R * eu)
        qw, qx, qy, qz = x[3:7]
        r = np.array([2 * (qx * qz + qw * qy), 2 * (qy * qz - qw * qx), 1 - 2 * (qx * qx + qy * qy) - 1.0])
        tot += 0.5 * wu * r @ r
        com, _ = ol.reference_kinematics(x)
        ps = 0.5 * (prob["ee_ref"][0, t, 0, :2] + prob["ee_ref"][0, t, 1, :2])
        rb = com[:2] + x[26:28] * np.sqrt(com[2] / 9.81) - ps
        tot += 0.5 * wb * rb @ rb
        tot += pen(x[7:26], jr, prob["w_joint"]) + pen(us[t] if t < N else np.zeros(19), cr, prob["w_ctrl"])
    assert abs(o.total_cost() - tot) < 1e-9 * abs(tot)


@pytest.mark.parametrize("case", ["spd", "bump"])
def test_backward_pass_vs_numpy_golden(case):
    r = np.load(os.path.join(G, "riccati_golden.npz"))
    A, B = r[case + "_A"], r[case + "_B"]
    N = A.shape[0]
    o = ol.Oracle(N, 0.02)
    o.set_options(lam=float(r["lam"]))
    o.set_linearization(A, B)
    o.set_quadratics(r[case + "_lx"], r[case + "_lu"], r[case + "_lxx"], r[case + "_luu"])
    o.backward_pass()
    for name, key in (("K", "K"), ("kff", "k"), ("Vx", "Vx"), ("Vxx", "Vxx")):
        got, want = o.get(name), r[case + "_" + key]
        tol = 1e-9 if case == "spd" else 1e-6
        assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max()), (name, np.abs(got - want).max())


def test_analytic_jacobians_vs_finite_differences():
    o = ol.Oracle(2, 0.02)
    prob = sc.make_problem(ol.reference_kinematics, N=2)
    o.set_problem(prob)
    rng = np.random.default_rng(3)
    x = sc.standing_state()
    x[7:26] = rng.uniform(-0.3, 0.3, 19)
    x[3:7] = sc._axis_angle_quat(rng.uniform(-0.5, 0.5, 3))
    x[26:] = rng.uniform(-1, 1, 25)
    u = rng.uniform(-5, 5, 19)
    o.set_trajectory(np.tile(x, (3, 1)), np.tile(u, (2, 1)))
    o.set_options(jac_mode=0)
    o.linearize()
    A0, B0 = o.get("A")[0], o.get("B")[0]
    eps = 1e-6
    Ac = np.stack([(o.step(x + eps * np.eye(51)[j], u) - o.step(x - eps * np.eye(51)[j], u)) / (2 * eps) for j in range(51)], axis=1)
    Bc = np.stack([(o.step(x, u + eps * np.eye(19)[j]) - o.step(x, u - eps * np.eye(19)[j])) / (2 * eps) for j in range(19)], axis=1)
    assert np.abs(A0 - Ac).max() < 1e-7 and np.abs(B0 - Bc).max() < 1e-7
    # the quaternion block annihilates the radial direction (normalisation inside the step)
    assert np.abs(A0[:, 3:7] @ x[3:7]).max() < 1e-10
    o.set_options(jac_mode=1, fd_eps=1e-5)  # reference-style forward differences (robot_utils.cpp:120-160)
    o.linearize()
    assert np.abs(o.get("A")[0] - A0).max() < 5e-5 and np.abs(o.get("B")[0] - B0).max() < 1e-7


def test_structure_of_the_step_jacobians_the_tangent_and_riccati_kernels_rely_on():
    """Two structural facts the round-4 tangent kernel (k_lin_tangent2) and the folded Riccati kernel build on, checked on the oracle's
    forward-mode AD Jacobians of the constraint-free step and on a complex-step-free, plain central difference of the step itself:
    (i) Galilean invariance -- the step does not depend on the base's linear velocity beyond the integrator: d f / d v_lin =
    [h I; 0; I; 0] (so the kernel does not sweep those three directions); (ii) the semi-implicit Euler integrator makes every
    hinge-position row a copy of its velocity row: A[7 + j] = e_(7+j) + h A[32 + j], B[7 + j] = h B[32 + j] (rows 8..23 are the ones the
    folded kernel never reads and k_lin_tangent2<true> does not write)."""
    h = 0.02
    o = ol.Oracle(2, h)
    rng = np.random.default_rng(11)
    for gravity in ((0.0, 0.0, -1.0), (0.0, 0.0, -9.81)):
        prob = sc.make_problem(ol.reference_kinematics, N=2, gravity=gravity)
        o.set_problem(prob)
        x = sc.standing_state()
        x[7:26] = rng.uniform(-0.4, 0.4, 19); x[3:7] = sc._axis_angle_quat(rng.uniform(-0.8, 0.8, 3)); x[26:] = rng.uniform(-1.5, 1.5, 25)
        u = rng.uniform(-20, 20, 19)
        o.set_trajectory(np.tile(x, (3, 1)), np.tile(u, (2, 1)))
        o.set_options(jac_mode=0)
        o.linearize()
        A, B = o.get("A")[0], o.get("B")[0]
        C = np.zeros((51, 3)); C[0:3] = h * np.eye(3); C[26:29] = np.eye(3)
        assert np.abs(A[:, 26:29] - C).max() < 1e-15
        # the same from the step itself (no derivative code involved): shifting v_lin shifts p' by h dv and v_lin' by dv, nothing else
        dv = np.array([0.3, -0.2, 0.5])
        x2 = x.copy(); x2[26:29] += dv
        d = o.step(x2, u) - o.step(x, u)
        want = np.zeros(51); want[0:3] = h * dv; want[26:29] = dv
        assert np.abs(d - want).max() < 1e-13
        E = np.zeros((19, 51)); E[np.arange(19), 7 + np.arange(19)] = 1.0
        assert np.abs(A[7:26] - (E + h * A[32:51])).max() < 1e-15 and np.abs(B[7:26] - h * B[32:51]).max() < 1e-15


def test_solve_control_flow_and_trace():
    o = ol.Oracle(25, 0.02)
    prob = sc.make_problem(ol.reference_kinematics)
    o.set_problem(prob)
    ug = o.grav_comp(sc.standing_state())
    x0, ui = sc.synthetic_batch(2, 25, 0, ug)
    o.set_options()
    o.initialize(x0[0], ui[0])
    ok, cost = o.solve(x0[0])
    it, tc, al, lam = o.trace()
    assert ok and 1 <= it <= 10
    assert np.all(np.diff(tc[: it + 1]) <= 1e-12)        # monotone: line search only accepts decreases
    assert abs(tc[it] - cost) == 0.0
    assert np.all(np.isin(al[:it], [0.0, 1.0, 0.8, 0.6, 0.4, 0.2, 0.1, 0.05, 0.01]))
    # closed-form quadratics and AD quadratics give the same solve
    o2 = ol.Oracle(25, 0.02)
    o2.set_problem(prob)
    o2.set_options(quad_mode=1, max_iter=2)
    o2.initialize(x0[0], ui[0])
    o2.solve(x0[0])
    o.set_options(max_iter=2)
    o.initialize(x0[0], ui[0])
    o.solve(x0[0])
    assert np.abs(o.get("K") - o2.get("K")).max() < 1e-7 * np.abs(o.get("K")).max()
    # warm start shift (ilqr.cpp:68-80)
    xb, ub = o.get("xbar"), o.get("ubar")
    o.initialize(xb[1], None, xb, ub)
    xs, us = o.get("xbar"), o.get("ubar")
    assert np.array_equal(us[:-1], ub[1:]) and np.array_equal(us[-1], ub[-1])
    assert np.array_equal(xs[1:-1], xb[2:]) and np.array_equal(xs[0], xb[1])
    assert np.allclose(xs[-1], o.step(xs[-2], us[-1]), atol=0)
    # control law (mpc.cpp:97-101)
    o.solve(xb[1])
    xm = o.get("xbar")[0] + 1e-3
    assert np.allclose(o.compute_control(xm), o.get("ubar")[0] + o.get("K")[0] @ (xm - o.get("xbar")[0]))


def test_stance_constrained_step_identities():
    """Contact row: rigid stance constraints on the scheduled feet.  No reference fixture exists (the
    reference's contacts live inside MuJoCo), so the restatement is pinned by what defines it: (i) no stance flag ->
    the constraint-free step bit for bit; (ii) a stance foot does not move over the step (velocity-level constraint:
    its origin travels O(h^2), orders of magnitude less than unconstrained); (iii) the constraint acts through the
    stance legs only: M_hat (qacc_c - qacc_free) has no component on the pelvis-fixed torso / arm hinges beyond what the
    leg wrenches transmit, i.e. it is zero on every hinge outside the stance legs; (iv) left / right symmetry of which
    leg carries it."""
    pkg = load_package()
    sc = pkg.scenario
    g = np.array([0.0, 0.0, -9.81])
    prob = sc.make_problem(ol.reference_kinematics, N=5, gravity=list(g))
    o = ol.Oracle(5, prob["dt"]); o.set_problem(prob)
    h = prob["dt"]
    rng = np.random.default_rng(3)
    x = sc.standing_state()
    x[7:26] += rng.uniform(-0.1, 0.1, 19); x[7 + 3] += 0.4; x[7 + 8] += 0.4     # bent knees
    x[3:7] = np.array([0.995, 0.05, -0.06, 0.03]); x[3:7] /= np.linalg.norm(x[3:7])
    x[26:] = rng.uniform(-0.3, 0.3, 25)
    u = rng.uniform(-20, 20, 19)
    free = o.step(x, u)
    o.set_contact_mode(1)
    assert np.array_equal(o.step_stance(x, u, [0, 0]), free)
    arm_eff = 0.1 + h * 1.0        # h1.xml: armature 0.1, damping 1 (implicit in the step)
    ee0 = ol.reference_kinematics(x)[1]
    move_free = np.linalg.norm(ol.reference_kinematics(free)[1] - ee0, axis=1)
    qacc_free = (free[26:] - x[26:]) / h
    leg = {0: list(range(6, 11)), 1: list(range(11, 16))}      # qvel indices of the left / right leg hinges
    for st in ([1, 1], [1, 0], [0, 1]):
        xn = o.step_stance(x, u, st)
        move = np.linalg.norm(ol.reference_kinematics(xn)[1] - ee0, axis=1)
        for f in range(2):
            if st[f]:
                assert move[f] < 5e-4 and move[f] < 0.1 * move_free[f], (st, f, move, move_free)
        qacc_c = (xn[26:] - x[26:]) / h
        dtau = ol.inverse_dynamics(x, qacc_c, arm_eff, g) - ol.inverse_dynamics(x, qacc_free, arm_eff, g)
        carried = sorted(i for f in range(2) if st[f] for i in leg[f])
        others = [i for i in range(6, 25) if i not in carried]
        assert np.abs(dtau[others]).max() < 1e-7 * max(1.0, np.abs(dtau).max()), (st, dtau)
        assert np.abs(dtau[carried]).max() > 1.0
    # the step stays differentiable: forward-mode AD Jacobians of the constrained step agree with central differences
    o2 = ol.Oracle(5, h); o2.set_problem(prob); o2.set_contact_mode(1)
    xb = np.tile(x, (6, 1)); ub = np.tile(u, (5, 1))
    o2.set_trajectory(xb, ub); o2.linearize()
    A = o2.get("A")[0]
    eps = 1e-6
    for j in (2, 10, 30, 45):
        xp, xm = x.copy(), x.copy(); xp[j] += eps; xm[j] -= eps
        col = (o2.step_stance(xp, u, [1, 1]) - o2.step_stance(xm, u, [1, 1])) / (2 * eps)
        assert np.abs(A[:, j] - col).max() < 1e-5 * max(1.0, np.abs(col).max()), (j, np.abs(A[:, j] - col).max())


def test_friction_limited_stance_step_against_the_independent_kkt_formulation():
    """Contact mode 3 (unilateral + Coulomb limit, round 4).  The oracle works through articulated-body unit-wrench responses and a
    projected constraint-space system; tests/golden/friction_golden.npz was produced by the dense NumPy KKT formulation of
    gen_golden.py (mass matrix by inverse dynamics, explicit constraint Jacobian, rows selected by S = [I3 0; 0 up^T] on a sliding
    foot) -- two derivations of the same rule.  Covers: cone inactive (== mode 2), one foot sliding (left / right), both sliding;
    and the rule's defining properties on the oracle alone: a sliding foot receives no tangential force (its horizontal
    acceleration equals that of the foot with the tangential rows absent) while its normal velocity is still held."""
    g = np.load(os.path.join(G, "friction_golden.npz"))
    prob = sc.make_problem(ol.reference_kinematics, N=5, gravity=list(g["gravity"]))
    o = ol.Oracle(5, float(g["h"])); o.set_problem(prob)
    pat = set()
    for i in range(len(g["x"])):
        o.set_contact_mode(3, float(g["soft"])); o.set_friction(float(g["mu"][i]))
        xn = o.step_stance(g["x"][i], g["u"][i], g["stance"][i])
        o.set_contact_mode(2, float(g["soft"]))
        x2 = o.step_stance(g["x"][i], g["u"][i], g["stance"][i])
        assert np.abs(xn - g["x_next"][i]).max() < 1e-9 * max(1.0, np.abs(g["x_next"][i]).max()), (i, np.abs(xn - g["x_next"][i]).max())
        assert np.abs(x2 - g["x_next_mode2"][i]).max() < 1e-9 * max(1.0, np.abs(g["x_next_mode2"][i]).max())
        if g["slide"][i].any():
            assert np.abs(xn - x2).max() > 1e-3
        else:
            assert np.array_equal(xn, x2)
        # mode 4: the same decision, kinetic friction mu f_n on the sliding feet (unsymmetric KKT in the generator, unit-wrench responses +
        # Gaussian elimination in the oracle)
        o.set_contact_mode(4, float(g["soft"]))
        x4 = o.step_stance(g["x"][i], g["u"][i], g["stance"][i])
        assert np.abs(x4 - g["x_next_mode4"][i]).max() < 1e-9 * max(1.0, np.abs(g["x_next_mode4"][i]).max()), (i, np.abs(x4 - g["x_next_mode4"][i]).max())
        if g["slide"][i].any():
            assert np.abs(x4 - xn).max() > 1e-3 and np.abs(x4 - x2).max() > 1e-3            # neither frictionless sliding nor sticking
        else:
            assert np.array_equal(x4, x2)
        pat.add(tuple(int(v) for v in g["slide"][i]))
        # normal velocity of every foot that carries load is held either way: the foot's height moves O(h^2)
        ee0 = ol.reference_kinematics(g["x"][i])[1]; ee1 = ol.reference_kinematics(xn)[1]
        for f in range(2):
            if g["act"][i][f]:
                assert abs(ee1[f][2] - ee0[f][2]) < 2e-3, (i, f, ee1[f][2] - ee0[f][2])
    assert pat == {(0, 0), (1, 0), (0, 1), (1, 1)}
    # mu = 0: every loaded foot slides; a huge mu: none does (== mode 2)
    i = 0
    o.set_contact_mode(3, float(g["soft"])); o.set_friction(1e9)
    o.set_contact_mode(3, float(g["soft"])); a = o.step_stance(g["x"][i], g["u"][i], g["stance"][i])
    o.set_contact_mode(2, float(g["soft"])); b2 = o.step_stance(g["x"][i], g["u"][i], g["stance"][i])
    assert np.array_equal(a, b2)


def test_joint_limit_rows_against_the_independent_kkt_formulation():
    """Joint-limit rows of the plant.  The oracle treats a limited hinge as an
    acceleration-prescribed joint of the articulated-body recursion (hybrid dynamics: no reduction of the articulated inertia, bias
    carries IA S qacc_i; the stance rows are solved on that system); tests/golden/joint_limit_golden.npz is the dense NumPy KKT system
    with the rows E qacc = -v_L / h and multipliers of their own (gen_golden.py kane_step_lim) -- two derivations of the same rule, on the
    constraint-free plant and with unilateral stance, with none / one / several hinges stopped.  And the rule's defining properties."""
    g = np.load(os.path.join(G, "joint_limit_golden.npz"))
    prob = sc.make_problem(ol.reference_kinematics, N=5, gravity=list(g["gravity"]))
    o = ol.Oracle(5, float(g["h"])); o.set_problem(prob)
    seen = set()
    for i in range(len(g["x"])):
        cm = int(g["contact"][i])
        o.set_contact_mode(cm, float(g["soft"]))
        o.set_joint_limits(True)
        xn = o.step_stance(g["x"][i], g["u"][i], g["stance"][i])
        o.set_joint_limits(False)
        x0 = o.step_stance(g["x"][i], g["u"][i], g["stance"][i])
        assert np.abs(xn - g["x_next"][i]).max() < 1e-9 * max(1.0, np.abs(g["x_next"][i]).max()), (i, np.abs(xn - g["x_next"][i]).max())
        assert np.abs(x0 - g["x_next_unlimited"][i]).max() < 1e-9 * max(1.0, np.abs(g["x_next_unlimited"][i]).max())
        lock = np.flatnonzero(g["lock"][i])
        if len(lock) == 0:
            assert np.array_equal(xn, x0)               # no hinge stopped: the step IS the unlimited one
        for j in lock:
            assert abs(xn[32 + j]) < 1e-12 and abs(xn[7 + j] - g["x"][i][7 + j]) < 1e-12      # the hinge stops, where it is
            assert abs(x0[32 + j]) > 0.05                                                       # ... and would not have
        # a hinge past its range that moves back in is left alone: outside the range, not locked, velocity unchanged in sign
        for j in range(19):
            out_hi, out_lo = g["x"][i][7 + j] > g["jrange"][j, 1], g["x"][i][7 + j] < g["jrange"][j, 0]
            if (out_hi or out_lo) and j not in lock:
                assert (xn[32 + j] < 0) if out_hi else (xn[32 + j] > 0)
        seen.add((cm, min(len(lock), 2)))
    assert seen == {(0, 0), (0, 1), (0, 2), (2, 0), (2, 1), (2, 2)}
    assert np.allclose(g["jrange"], ol.joint_ranges())


def test_joint_limit_rows_with_the_restoring_stiffness_against_the_independent_kkt_formulation():
    """Round 6 (VERDICT r5 "what's missing" 2: mj_step pushes a hinge back into its range, the pure stop does not).  With a stiffness k the
    row prescribes MuJoCo's constraint reference acceleration in its hard limit, qacc_i = -v_i / h - k r_i (r_i the violation; k = 1 / (2 h)^2
    = 625 is solref's default time constant, clamped to 2 h): v_i+ = -h k r_i, and the row is active whenever the unlimited step falls short of
    that on the outward side -- a hinge that drifts back in too slowly is constrained too.  tests/golden/joint_limit_stiffness_golden.npz: the
    dense NumPy KKT system with the rows E qacc = -v_L / h - k r (gen_golden.py kane_step_lim(stiffness=...)); the oracle: its hybrid
    recursion.  k = 0 is the round-5 rule, bit for bit."""
    g = np.load(os.path.join(G, "joint_limit_stiffness_golden.npz"))
    k, h = float(g["stiffness"]), float(g["h"])
    assert abs(k - 1.0 / (2 * h) ** 2) < 1e-9
    prob = sc.make_problem(ol.reference_kinematics, N=5, gravity=list(g["gravity"]))
    o = ol.Oracle(5, h); o.set_problem(prob)
    seen, slow_in = set(), 0
    for i in range(len(g["x"])):
        cm = int(g["contact"][i])
        o.set_contact_mode(cm, float(g["soft"]))
        o.set_joint_limits(True); o.set_joint_limit_stiffness(k)
        xn = o.step_stance(g["x"][i], g["u"][i], g["stance"][i])
        o.set_joint_limits(False)
        x0 = o.step_stance(g["x"][i], g["u"][i], g["stance"][i])
        assert np.abs(xn - g["x_next"][i]).max() < 1e-9 * max(1.0, np.abs(g["x_next"][i]).max()), (i, np.abs(xn - g["x_next"][i]).max())
        assert np.abs(x0 - g["x_next_unlimited"][i]).max() < 1e-9 * max(1.0, np.abs(g["x_next_unlimited"][i]).max())
        lock = np.flatnonzero(g["lock"][i])
        if len(lock) == 0:
            assert np.array_equal(xn, x0)
        for j in lock:
            r = g["x"][i][7 + j] - (g["jrange"][j, 1] if g["x"][i][7 + j] > g["jrange"][j, 1] else g["jrange"][j, 0])
            assert abs(xn[32 + j] + h * k * r) < 1e-10                    # pushed back: v+ = -h k r, a quarter of the violation per step
            assert abs((xn[7 + j] - g["x"][i][7 + j]) + 0.25 * r) < 1e-10
            slow_in += int(g["x"][i][32 + j] * r < 0)                      # ... also when it was already on its way back, too slowly
        seen.add((cm, min(len(lock), 2)))
    assert seen == {(0, 0), (0, 1), (0, 2), (2, 0), (2, 1), (2, 2)} and slow_in >= 1
    # k = 0: the round-5 rule exactly (same golden as before, same bits whether the stiffness was ever set or not)
    g0 = np.load(os.path.join(G, "joint_limit_golden.npz"))
    o.set_joint_limits(True)
    for i in range(len(g0["x"])):
        o.set_contact_mode(int(g0["contact"][i]), float(g0["soft"]))
        o.set_joint_limit_stiffness(k); a = o.step_stance(g0["x"][i], g0["u"][i], g0["stance"][i])
        o.set_joint_limit_stiffness(0.0); b = o.step_stance(g0["x"][i], g0["u"][i], g0["stance"][i])
        assert np.abs(b - g0["x_next"][i]).max() < 1e-9 * max(1.0, np.abs(g0["x_next"][i]).max())
        if np.flatnonzero(g0["lock"][i]).size:
            assert np.abs(a - b).max() > 1e-6
    # closed loop on one hinge: the left knee 0.05 rad past its upper limit, at rest.  The first step takes a quarter of the violation
    # back (v+ = -h k r); from then on the hinge is on its way in faster than the row asks for and the row -- which only pushes -- lets it
    # go: the violation shrinks every step and is gone within a few; with the pure stop (k = 0) the knee stays where it is until
    # something else moves it
    o.set_contact_mode(0, 0.0)
    hi = ol.joint_ranges()[3, 1]
    u = np.zeros(19)
    for kk in (k, 0.0):
        o.set_joint_limit_stiffness(kk)
        x = sc.standing_state().copy(); x[7 + 3] = hi + 0.05
        viol = []
        for _ in range(8):
            viol.append(x[7 + 3] - hi)
            x = o.step_stance(x, u, np.array([0, 0]))
        if kk > 0.0:
            assert abs(viol[1] - 0.75 * viol[0]) < 1e-9 and all(viol[n + 1] < viol[n] for n in range(7)) and viol[-1] < 0.0
        else:
            assert viol[1] > 0.049


def test_op_counter_pins_the_algorithmic_flop_figures_of_the_bench():
    """'flops of dynamics + analytic Jacobians + cost quadratics + one line-search alpha: to be taken from the CPU
    restatement's op counter'.  The counter (oracle/opcount.cpp) runs the oracle's own code on a counting scalar; bench.py's
    per-stage constants are these numbers, at knot 7 of rollout 0 of the seed-0 standing batch under the shipped weights.  The
    Jacobian scheme that is counted is first checked to BE a correct algorithm: on plain doubles it reproduces the oracle's
    forward-mode AD Jacobians."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod3", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    N = 25
    prob = sc.make_problem(ol.reference_kinematics, N=N)
    o = ol.Oracle(N, prob["dt"]); o.set_problem(prob)
    ug = o.grav_comp(sc.standing_state())
    x0, ui = sc.synthetic_batch(2, N, 0, ug)
    o.initialize(x0[0], ui[0]); o.linearize()
    xb, ub = o.get("xbar"), o.get("ubar")
    for t in (0, 7, 24):
        A, B = ol.tangent_scheme_jacobians(xb[t], ub[t], prob["dt"], prob["gravity"])
        assert np.abs(A - o.get("A")[t]).max() < 1e-12 and np.abs(B - o.get("B")[t]).max() < 1e-12
    flops, trig = o.op_counts(7, xb[7], ub[7])
    assert (flops["step"], flops["alpha_trial_knot"], flops["quadratics_knot"]) == (int(bench.STEP_FLOPS), int(bench.ALPHA_TRIAL_FLOPS_PER_KNOT), int(bench.QUAD_FLOPS_PER_KNOT)), flops
    # (the zero-aware part of the Jacobian count sees the occasional exact zero of the data: a few hundred flops in 200 k)
    assert abs(flops["jacobians"] - bench.JACOBIAN_FLOPS_PER_KNOT) < 1e-3 * bench.JACOBIAN_FLOPS_PER_KNOT, flops
    # the counts are properties of the algorithm, not of the knot: another state of the trajectory gives the same step / trial
    # counts, the quadratics of the terminal knot lose the control terms, the Jacobian count moves only with exact zeros in the data
    f2, _ = o.op_counts(N, xb[N], ub[N - 1])
    assert f2["step"] == flops["step"] and abs(f2["jacobians"] - flops["jacobians"]) < 0.01 * flops["jacobians"]
    assert 0 < flops["quadratics_knot"] - f2["quadratics_knot"] < 200
    assert trig["step"] == 40          # 19 hinges + the quaternion's half angle, sine and cosine each
    # the whole-iteration total bench.py reports beside SURVEY's planning budget of 27.8 MFLOP
    assert abs(bench.ITER_FLOPS_COUNTED_N25 - 32952292.0) < 1.0
