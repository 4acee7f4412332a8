"""GPU tests of the BASELINE.json configurations at their true per-GPU shapes, of every selectable kernel variant, and of the
multi-rank shard / gather path as far as one GPU can execute it (two ranks on device 0)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol
from conftest import load_package

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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


class env:
    """Set environment variables for a block (the library re-reads its kernel-variant switches at every call)."""

    def __init__(self, **kv):
        self.kv = {k: v for k, v in kv.items() if v is not None}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def standing(B, N=25, seed=0, gravity=None):
    from mpc_ilqr_mujoco_amd import solver as sv
    prob = sc.make_problem(sv.reference_kinematics, N=N, gravity=gravity)
    ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
    x0, ui = sc.synthetic_batch(B, N, seed, ug)
    return prob, x0, ui


# ---------------------------------------------------------------------------------------------------------------------
# kernel variants: every family that ships is exercised against the oracle / the NumPy golden
VARIANTS = [
    dict(ILQR_BACKWARD="wave", ILQR_LS="s", ILQR_ROLLOUT="s"),     # defaults
    dict(ILQR_BACKWARD="wg", ILQR_LS="r", ILQR_ROLLOUT="r"),       # four-wave MFMA Riccati, one lane per trajectory
    dict(ILQR_BACKWARD="valu", ILQR_LS="s", ILQR_ROLLOUT="r"),     # VALU Riccati; mixed line search / rollout -> sequential re-rollout
    dict(ILQR_BACKWARD="wave", ILQR_DYN="s"),                      # scratch-resident scalar dynamics everywhere
    dict(ILQR_BACKWARD="wave-fold", ILQR_LS="s", ILQR_ROLLOUT="s"),  # the folded one-wave Riccati kernel on the standard layout
]


@pytest.mark.parametrize("var", VARIANTS, ids=lambda v: ",".join("%s=%s" % (k[5:].lower(), x) for k, x in v.items()))
def test_full_solve_parity_for_every_kernel_variant(var):
    B = 4
    prob, x0, ui = standing(B, seed=31)
    with env(**var):
        s = _solver(B); s.set_problem(prob); s.set_max_iterations(5)
        s.initialize(x0, ui)
        cost = s.solve(x0)
        tc, ta, tl = s.trace()
        K, it = s.gains_K(), s.iterations()
        mism = s.adopt_mismatches()
        s.close()
    # wherever the re-rollout ran beside the linearisation it reproduced the accepted candidate bit for bit
    assert mism == 0
    for b in range(B):
        o = ol.Oracle(prob["N"], prob["dt"]); o.set_problem(prob); o.set_options(max_iter=5)
        o.initialize(x0[b], ui[b]); ok, c = o.solve(x0[b])
        n, oc, oa, _ = o.trace()
        assert n == it[b] and np.allclose(tc[b, : n + 1], oc[: n + 1], rtol=1e-5, atol=0) and np.array_equal(ta[b, :n], oa[:n])
        assert abs(cost[b] - c) <= 1e-5 * abs(c) and rel(K[b], o.get("K")) < 1e-5


@pytest.mark.parametrize("backward", ["wave", "wg", "valu"])
def test_backward_kernels_vs_numpy_golden_and_indefinite_fallback(backward):
    r = np.load(os.path.join(G, "riccati_golden.npz"))
    with env(ILQR_BACKWARD=backward):
        for case, tol in (("spd", 1e-9), ("bump", 1e-6)):
            A, Bm = r[case + "_A"], r[case + "_B"]
            s = _solver(2, N=A.shape[0]); s.set_regularization(float(r["lam"]))
            rep = lambda a: np.stack([a, a])
            s.set_linearization(rep(A), rep(Bm))
            s.set_quadratics(rep(r[case + "_lx"]), rep(r[case + "_lu"]), rep(r[case + "_lxx"]), rep(r[case + "_luu"]))
            s.stage_backward_pass()
            Vx, Vxx = s.value_function()
            for got, key in ((s.gains_K(), "K"), (s.gains_kff(), "k"), (Vx, "Vx"), (Vxx, "Vxx")):
                want = r[case + "_" + key]
                assert np.abs(got[1] - want).max() <= tol * max(1.0, np.abs(want).max()), (backward, case, key)
            s.close()
        # indefinite Quu even after the +1e-4 bump (ilqr.cpp:278-285): NumPy restatement of the long form
        rng = np.random.default_rng(11)
        N, n, m, lam = 4, 51, 19, 1e-6
        A = np.eye(n)[None] + 0.05 * rng.standard_normal((N, n, n)); Bm = 0.1 * rng.standard_normal((N, n, m))
        lx = rng.standard_normal((N + 1, n)); lu = rng.standard_normal((N, m))
        lxx = np.stack([np.diag(rng.uniform(1.0, 3.0, n)) for _ in range(N + 1)]); luu = rng.uniform(0.5, 1.5, (N, m))
        luu[2, 3] = -40.0; luu[1, 7] = -25.0
        Vx, Vxx = lx[N].copy(), lxx[N].copy()
        Kw = np.zeros((N, m, n))
        for t in range(N - 1, -1, -1):
            Qx = lx[t] + A[t].T @ Vx; Qu = lu[t] + Bm[t].T @ Vx
            Qxx = lxx[t] + A[t].T @ Vxx @ A[t]; Quu = np.diag(luu[t]) + Bm[t].T @ Vxx @ Bm[t] + lam * np.eye(m); Qxu = A[t].T @ Vxx @ Bm[t]
            if np.linalg.eigvalsh(Quu).min() <= 0:
                Quu = Quu + 1e-4 * np.eye(m)
            K = -np.linalg.solve(Quu, Qxu.T); k = -np.linalg.solve(Quu, Qu)
            Vx = Qx + K.T @ Quu @ k + K.T @ Qu + Qxu @ k
            Vxx = Qxx + K.T @ Quu @ K + K.T @ Qxu.T + Qxu @ K; Vxx = 0.5 * (Vxx + Vxx.T)
            Kw[t] = K
        s = _solver(2, N=N); s.set_regularization(lam)
        rep = lambda a: np.stack([a] * 2)
        s.set_linearization(rep(A), rep(Bm)); s.set_quadratics(rep(lx), rep(lu), rep(lxx), rep(luu))
        s.stage_backward_pass()
        gVx, gVxx = s.value_function()
        assert rel(s.gains_K()[1], Kw) < 1e-7 and rel(gVx[1], Vx) < 1e-7 and rel(gVxx[0], Vxx) < 1e-7
        s.close()


def _riccati_numpy(A, Bm, lx, lu, lxx, luu, lam):
    """ilqr.cpp:250-309 for one rollout (positive definite Quu: no bump)."""
    N, n, m = A.shape[0], A.shape[1], Bm.shape[2]
    Vx, Vxx = lx[N].copy(), lxx[N].copy()
    K, k = np.zeros((N, m, n)), np.zeros((N, m))
    for t in range(N - 1, -1, -1):
        Qx = lx[t] + A[t].T @ Vx; Qu = lu[t] + Bm[t].T @ Vx
        Qxx = lxx[t] + A[t].T @ Vxx @ A[t]; Quu = np.diag(luu[t]) + Bm[t].T @ Vxx @ Bm[t] + lam * np.eye(m); Qxu = A[t].T @ Vxx @ Bm[t]
        if np.linalg.eigvalsh(Quu).min() <= 0:
            Quu = Quu + 1e-4 * np.eye(m)
        K[t] = -np.linalg.solve(Quu, Qxu.T); k[t] = -np.linalg.solve(Quu, Qu)
        Vx = Qx + K[t].T @ Quu @ k[t] + K[t].T @ Qu + Qxu @ k[t]
        Vxx = Qxx + K[t].T @ Quu @ K[t] + K[t].T @ Qxu.T + Qxu @ K[t]; Vxx = 0.5 * (Vxx + Vxx.T)
    return K, k, Vx, Vxx


@pytest.mark.parametrize("contact", [0, 2])
def test_folded_backward_pass_equals_generic_kernel_and_numpy(contact):
    """The one-wave Riccati kernels drop the k-steps of the position rows when the Jacobians come from the analytic linearisation
    (A[p] = e_p + h A[v(p)], B[p] = h B[v(p)] for the nineteen hinge angles and the three base positions): riccati_pack.hip
    (ILQR_BACKWARD=wave, the default: operand layout, all 22 rows) and riccati_wave.hip fold_rows (ILQR_BACKWARD=wave-fold: standard
    layout, the sixteen rows that fill whole k-steps).  Same Jacobians and quadratics through both, the generic kernel
    (ILQR_BACKWARD=wave-generic) and NumPy; then with an indefinite Quu, where the folded products feed the Gauss-Jordan fallback."""
    B = 3
    prob, x0, ui = standing(B, seed=17, gravity=[0.0, 0.0, -9.81] if contact else None)
    s = _solver(B, legacy=True); s.set_problem(prob); s.set_contact_mode(contact); s.set_options(jacobian_mode=0); s.set_regularization(1e-6)
    s.initialize(x0, ui)
    s.stage_linearize(); s.stage_cost_quadratics()
    A, Bm = s.linearization()
    # the structure the fold relies on: exact off the diagonal, one rounding of 1 + h a on it (the kernel may fuse it)
    h = prob["dt"]
    for p_, v_ in [(7 + j, 32 + j) for j in range(19)] + [(i, 26 + i) for i in range(3)]:
        e = np.zeros(51); e[p_] = 1.0
        assert np.abs(A[:, :, p_, :] - (e + h * A[:, :, v_, :])).max() <= 2.3e-16 and np.array_equal(Bm[:, :, p_, :], h * Bm[:, :, v_, :])
    lx, lu, lxx, luu = s.quadratics()
    for indefinite in (False, True):
        if indefinite:
            luu = luu.copy(); luu[:, 7, 3] = -4e4; luu[:, 20, 11] = -2.5e4
            s.set_quadratics(lx, lu, lxx, luu)
        out = {}
        for kind in ("wave", "wave-fold", "wave-generic"):
            with env(ILQR_BACKWARD=kind):
                s.stage_backward_pass()
                out[kind] = (s.gains_K(), s.gains_kff()) + tuple(s.value_function())
        tol = 1e-6 if indefinite else 1e-9
        for kind in ("wave", "wave-fold"):
            for got, want in zip(out[kind], out["wave-generic"]):
                assert rel(got, want) < tol, (kind, contact, indefinite, rel(got, want))
            assert any(not np.array_equal(g, w) for g, w in zip(out[kind], out["wave-generic"]))     # different kernels did run
        assert any(not np.array_equal(g, w) for g, w in zip(out["wave"], out["wave-fold"]))
        # the conversions to the operand layout and back leave the standard layout as it was (lxx: symmetrised from its stored half)
        A_, B_ = s.linearization(); q_ = s.quadratics()
        assert np.array_equal(A_, A) and np.array_equal(B_, Bm) and np.array_equal(q_[0], lx) and rel(q_[2], lxx) < 1e-12
        for b in range(B):
            ref = _riccati_numpy(A[b], Bm[b], lx[b], lu[b], lxx[b], luu[b], 1e-6)
            for kind in ("wave", "wave-fold"):
                for got, want in zip(out[kind], ref):
                    assert rel(got[b], want) < 10 * tol, (kind, contact, indefinite, b, rel(got[b], want))
    s.close()


def _slot_tile():
    """slot tile (0..3) of every state in the operand layout (csrc/riccati_pack.h pk_state_slot)"""
    def slot(st):
        return 19 + st if st < 3 else 51 + st if st < 7 else st - 7 if st < 26 else st + 25 if st < 29 else st + 29 if st < 31 else 63 if st == 31 else st
    return np.array([slot(st) >> 4 for st in range(51)])


@pytest.mark.parametrize("kind", ["wave", "wave-generic"])
def test_stage_backward_pass_on_another_kernel_family_after_a_solve_sees_the_whole_lxx(kind):
    """Inside a solve the cost quadratics store only half of the symmetric lxx_t: the tiles on and below the diagonal of the operand
    layout (default kernel, riccati_pack.h: every knot, in slot order, with lx in the vector slot) or of the state order (generic
    one-wave kernel: the knots t < N).  The stage API afterwards -- the getter, and the backward pass on a kernel family that reads
    the full matrix in the standard layout -- must see the symmetric whole: mirrored by the conversion kernel / on the host (getter)
    or on the device (stage call)."""
    B = 3
    prob, x0, ui = standing(B, seed=29)
    with env(ILQR_BACKWARD=kind):
        s = _solver(B, legacy=True); s.set_problem(prob); s.set_max_iterations(2); s.set_options(early_exit=False)      # (the VALU family below: test library)
        s.initialize(x0, ui); s.solve()
        Ksolve = s.gains_K()
        lx, lu, lxx, luu = s.quadratics()
    # entries whose two indices lie in different tiles exist once and are mirrored bit for bit; inside a diagonal tile the two halves
    # are computed independently (symmetric to rounding)
    tile = _slot_tile() if kind == "wave" else np.arange(51) >> 4
    knots = slice(None) if kind == "wave" else slice(None, -1)
    off = tile[:, None] != tile[None, :]
    H = lxx[:, knots]
    assert np.array_equal(H[..., off], np.swapaxes(H, -1, -2)[..., off])
    assert np.abs(H[..., off]).max() > 0 and np.abs(lxx - np.swapaxes(lxx, -1, -2)).max() < 1e-9 * np.abs(lxx).max()
    s.set_regularization(1e-6)
    with env(ILQR_BACKWARD=kind):
        s.stage_backward_pass(); Kw = s.gains_K()
    with env(ILQR_BACKWARD="valu"):
        s.stage_backward_pass(); Kv = s.gains_K()
    assert rel(Kv, Kw) < 1e-9
    A, Bm = s.linearization()
    for b in range(B):
        ref = _riccati_numpy(A[b], Bm[b], lx[b], lu[b], lxx[b], luu[b], 1e-6)
        assert rel(Kv[b], ref[0]) < 1e-8
    s.close()


def test_position_rows_of_the_jacobians_have_no_slot_in_the_operand_layout_and_are_rebuilt_on_demand():
    """Inside a solve whose backward pass is the operand-layout Riccati kernel the two-knot tangent kernel writes A_t / B_t straight into
    that layout (riccati_pack.h), which has no slot for the 22 position rows (= e_p + h x their velocity row).  The getter converts
    back and rebuilds them: the result equals, to the last bits, what the one-knot kernel (ILQR_LINT=1: standard layout, converted by
    k_pack_ab inside the solve; it also sweeps the three base-linear-velocity directions the two-knot kernel drops as analytically
    zero) and the standard-layout solve (ILQR_BACKWARD=wave-fold) leave behind, and the structure holds exactly."""
    B = 5
    prob, x0, ui = standing(B, seed=37)
    out = {}
    for name, var in (("pack", {}), ("lint", dict(ILQR_LINT="1")), ("fold", dict(ILQR_BACKWARD="wave-fold"))):
        with env(**var):
            s = _solver(B); s.set_problem(prob); s.set_max_iterations(3); s.set_options(early_exit=False)
            s.initialize(x0, ui); s.solve()
            out[name] = s.linearization() + (s.gains_K(), s.cost())
            out[name + "2"] = s.linearization()          # (a second call finds the standard layout)
            s.close()
    A2, B2, K2, c2 = out["pack"]
    h = prob["dt"]
    assert np.all(np.isfinite(A2)) and np.all(np.isfinite(B2))
    assert np.array_equal(out["pack2"][0], A2) and np.array_equal(out["pack2"][1], B2)
    for other in ("lint", "fold"):
        A1, B1, K1, c1 = out[other]
        assert rel(K2, K1) < 1e-9 and rel(c2, c1) < 1e-12, other
        assert np.abs(A2 - A1).max() < 1e-12 and np.abs(B2 - B1).max() < 1e-12, other
    # structure of the rebuilt rows
    pos = np.r_[0:3, 7:26]; vel = np.r_[26:29, 32:51]
    E = np.zeros((22, 51)); E[np.arange(22), pos] = 1.0
    assert np.abs(A2[:, :, pos, :] - (E + h * A2[:, :, vel, :])).max() < 1e-15
    assert np.abs(B2[:, :, pos, :] - h * B2[:, :, vel, :]).max() < 1e-15
    # base linear velocity columns: d f / d v_lin = [h I; 0; I; 0] exactly in the two-knot kernel
    C = np.zeros((51, 3)); C[0:3] = h * np.eye(3); C[26:29] = np.eye(3)
    assert np.array_equal(A2[:, :, :, 26:29], np.broadcast_to(C, A2[:, :, :, 26:29].shape))


def test_early_exit_gate_stops_launching_and_changes_nothing():
    """With the convergence exit on, the host follows the device-side count of active rollouts and stops enqueuing iterations
    once the batch is done (ilqr_capi.hip enqueue_solve); the compacted work lists (DevState::order) feed the Riccati and
    line-search launches.  Same traces, iteration counts, gains and trajectories with the gate off, and against the oracle."""
    B = 6
    prob, x0, ui = standing(B, seed=41)
    out = {}
    for gate in ("1", "0"):
        with env(ILQR_EE_GATE=gate):
            s = _solver(B); s.set_problem(prob); s.set_options(early_exit=True); s.set_max_iterations(10)
            s.initialize(x0, ui); cost = s.solve(x0)
            # warm re-solve from the converged solution: every rollout leaves after a few iterations
            s.initialize_warm_resident(x0); cost2 = s.solve(x0)
            out[gate] = (cost, cost2, s.iterations(), s.gains_K(), s.xbar(), s.trace()[0], s.iterations_enqueued())
            s.close()
    # the per-handle switch (ilqr_hip_set_early_exit_gate) does what the process-wide environment variable does
    s = _solver(B); s.set_problem(prob); s.set_options(early_exit=True); s.set_max_iterations(10); s.set_early_exit_gate(False)
    s.initialize(x0, ui); cost = s.solve(x0)
    s.initialize_warm_resident(x0); cost2 = s.solve(x0)
    out["handle-off"] = (cost, cost2, s.iterations(), s.gains_K(), s.xbar(), s.trace()[0], s.iterations_enqueued())
    s.close()
    assert out["handle-off"][6] == 10
    for a, b_ in zip(out["1"][:6], out["0"][:6]):
        assert np.array_equal(a, b_, equal_nan=True)
    for a, b_ in zip(out["1"][:6], out["handle-off"][:6]):
        assert np.array_equal(a, b_, equal_nan=True)
    its = out["1"][2]
    assert out["0"][6] == 10 and out["1"][6] < 10 and out["1"][6] >= its.max()      # launched no more than one or two idle iterations
    for b in range(B):
        o = ol.Oracle(prob["N"], prob["dt"]); o.set_problem(prob); o.set_options(max_iter=10, early_exit=1)
        o.initialize(x0[b], ui[b]); ok, c = o.solve(x0[b])
        assert abs(out["1"][0][b] - c) <= 1e-5 * abs(c)


@pytest.mark.parametrize("mode", ["fixed", "early_exit", "contact", "wg-r"])
def test_speculative_lambda_retry_is_the_sequential_retry_bit_for_bit(mode):
    """(i) Small passes run the lambda retry of ilqr.cpp:619-644 beside the first pass (k_control_spec): Riccati and line search for
    lambda and for min(10 lambda, 1e-3) on two streams, the bookkeeping played once with both outcomes.  Every observable of the
    solve -- cost trace, accepted step sizes, lambda schedule, iteration counts, gains, feed-forward, value function, trajectory,
    final lambda -- equals the sequential order's (ILQR_SPEC=0) bit for bit, on cold starts that walk through accept, fail -> retry
    -> accept and fail -> retry -> fail (-> continue / break), with and without the convergence exit, in contact mode and on the
    other kernel families; a pass above the threshold (ILQR_SPEC_MAX) stays sequential.  (ii) Where the retry stays sequential, the
    rollouts whose first line search accepted start the next iteration's linearisation / cost quadratics / re-rollout beside the
    retry of the others (early continuation, two groups per concurrent region): bit for bit the one-group order (ILQR_SPLIT=0)."""
    B = 12
    gravity = [0.0, 0.0, -9.81] if mode == "contact" else None
    prob, x0, ui = standing(B, seed=53, gravity=gravity)
    var = dict(ILQR_BACKWARD="wg", ILQR_LS="r", ILQR_ROLLOUT="r") if mode == "wg-r" else {}
    out = {}
    # ("max": threshold below the batch -- with the convergence exit the device then chooses between the two enqueued orders while the host's
    #  count is within four times the threshold, launch_spec_gate; "max5": a threshold the pass only falls below near the end; "max-nodual":
    #  the host's one-iteration-old count alone decides)
    for spec in ("1", "0", "max", "0-nosplit") + (("max5", "max-nodual") if mode == "early_exit" else ()):
        kv = dict(var); kv.update(ILQR_SPEC="1" if spec.startswith("max") else spec[0])
        if spec.startswith("max"):
            kv["ILQR_SPEC_MAX"] = "5" if spec == "max5" else str(B - 1)
        if spec == "max-nodual":
            kv["ILQR_SPEC_DUAL"] = "0"
        kv["ILQR_SPLIT"] = "0" if spec == "0-nosplit" else "1"       # (the default: on with the convergence exit only)
        with env(**kv):
            s = _solver(B); s.set_problem(prob); s.set_options(early_exit=(mode == "early_exit")); s.set_max_iterations(10)
            if mode == "contact":
                s.set_contact_mode(2)
            s.initialize(x0, ui); cost = s.solve(x0)
            tc, ta, tl = s.trace()
            Vx, Vxx = s.value_function()
            out[spec] = (cost, tc, ta, tl, s.iterations(), s.gains_K(), s.gains_kff(), Vx, Vxx, s.xbar(), s.ubar(), s.lambdas(), s.speculative_iterations(), s.adopt_mismatches(), s.split_iterations())
            s.close()
    assert out["1"][12] >= 3 and out["0"][12] == 0 and out["1"][13] == 0
    # early continuation (two groups per concurrent region): on wherever the sequential retry runs beside a concurrent re-rollout
    assert out["0-nosplit"][14] == 0 and out["0-nosplit"][12] == 0 and out["0"][13] == 0 and out["max"][13] == 0
    if mode != "wg-r":
        assert out["0"][14] >= 3 and out["1"][14] == (0 if mode != "early_exit" else out["1"][14])
    for k in range(12):
        assert np.array_equal(out["0"][k], out["0-nosplit"][k], equal_nan=True), k
    if mode != "early_exit":
        assert out["1"][12] == 10 and out["max"][12] == 0         # (early exit: the pass shrinks below the threshold on its way)
    else:
        assert 0 < out["max-nodual"][12] < out["1"][12] and out["max"][12] >= out["max-nodual"][12] and out["max5"][12] >= 3
    for k in range(12):
        for v in out:
            assert np.array_equal(out[v][k], out["0"][k], equal_nan=True), (v, k)
    ta = out["1"][2]
    it = out["1"][4]
    fails = sum(int((ta[b, : it[b]] == 0.0).sum()) for b in range(B))
    accepts = sum(int((ta[b, : it[b]] > 0.0).sum()) for b in range(B))
    assert fails >= 3 and accepts >= 2 * B                            # both outcomes of the retry occur
    tl = out["1"][3]
    assert np.nanmax(tl) > 1e-6                                        # ... and accepted steps that needed the bumped lambda
    for b in (0, B - 1):
        o = ol.Oracle(prob["N"], prob["dt"]); o.set_problem(prob); o.set_options(max_iter=10, early_exit=int(mode == "early_exit"))   # (lambda lives in the solver object)
        if mode == "contact":
            o.set_contact_mode(2)
        o.initialize(x0[b], ui[b]); ok, c = o.solve(x0[b])
        n, oc, oa, olam = o.trace()
        assert n == it[b] and np.allclose(out["1"][1][b, : n + 1], oc[: n + 1], rtol=1e-5) and np.array_equal(ta[b, :n], oa[:n])
        assert np.allclose(tl[b, :n], olam[:n], rtol=1e-12) and rel(out["1"][5][b], o.get("K")) < 1e-5


@pytest.mark.parametrize("early_exit", [False, True])
def test_joint_limit_rows_under_every_launch_order(early_exit):
    """The joint-limit rows (ilqr_hip_set_joint_limits) run in kernels of their own (rollout, line search, dump, tangent):
    with hinges past their ranges, unilateral stance and analytic Jacobians, every launch order of the solve -- speculative lambda retry,
    sequential retry with and without the early continuation -- gives the same bits, the nominal re-rollout reproduces the accepted
    candidates, and the solve is the oracle's; the plant step of the closed loop (ilqr_hip_step) carries the rows as well."""
    B = 12
    prob, x0, ui = standing(B, seed=57, gravity=[0.0, 0.0, -9.81])
    x0 = x0.copy()
    x0[:, 7 + 3] = 2.08; x0[:, 32 + 3] = 1.2                     # left knee past its range, opening
    x0[B // 2:, 7 + 14] = 4.47; x0[B // 2:, 32 + 14] = 0.8       # left elbow past its range, opening
    out = {}
    for spec in ("1", "0", "0-nosplit"):
        with env(ILQR_SPEC=spec[0], ILQR_SPLIT="0" if spec == "0-nosplit" else "1"):
            s = _solver(B); s.set_problem(prob); s.set_contact_mode(2); s.set_joint_limits(True)
            s.set_options(early_exit=early_exit); s.set_max_iterations(6)
            s.initialize(x0, ui); cost = s.solve(x0)
            tc, ta, tl = s.trace()
            out[spec] = (cost, tc, ta, tl, s.iterations(), s.gains_K(), s.gains_kff(), s.xbar(), s.ubar(), s.lambdas(), s.adopt_mismatches())
            if spec == "1":
                x1 = s.step(x0, s.ubar()[:, 0])                   # the plant of the closed loop carries the rows too
                warm = (s.xbar().copy(), s.ubar().copy(), x1)
            s.close()
    for v in out:
        assert out[v][10] == 0
        for k in range(10):
            assert np.array_equal(out[v][k], out["0"][k], equal_nan=True), (v, k)
    stopped = 0
    for b in (0, B - 1):
        o = ol.Oracle(prob["N"], prob["dt"]); o.set_problem(prob); o.set_options(max_iter=6, early_exit=int(early_exit)); o.set_contact_mode(2); o.set_joint_limits(True)
        o.initialize(x0[b], ui[b]); ok, c = o.solve(x0[b])
        n, oc, oa, olam = o.trace()
        it = out["1"][4]
        assert n == it[b] and np.allclose(out["1"][1][b, : n + 1], oc[: n + 1], rtol=1e-5) and np.array_equal(out["1"][2][b, :n], oa[:n])
        assert rel(out["1"][5][b], o.get("K")) < 1e-5 and rel(out["1"][7][b], o.get("xbar")) < 1e-6
        assert rel(warm[2][b], o.step_stance(x0[b], o.get("ubar")[0], [1, 1])) < 1e-9
        stopped += int(abs(warm[2][b][32 + 3]) < 1e-12)
    assert stopped >= 1                                            # the plant stopped the knee on its first step


@pytest.mark.parametrize("cm", [0, 4])
def test_joint_limit_rows_are_batch_invariant_across_the_wide_kernels(cm):
    """A batch above 1024 rollouts takes the four-rollouts-per-wave line search (k_line_search_s<3 | 4, 4>), a small one the
    one-rollout-per-wave kernels: with the joint-limit rows on (constraint-free plant, and contact mode 4 with a low friction
    coefficient) the same rollouts give the same bits in either, wherever they sit in the batch."""
    Bs, B = 6, 1030
    gravity = [0.0, 0.0, -9.81] if cm else None
    prob, x0, ui = standing(Bs, seed=61, gravity=gravity)
    x0 = x0.copy()
    x0[:, 7 + 3] = 2.08; x0[:, 32 + 3] = 1.2
    x0[Bs // 2:, 7 + 14] = 4.47; x0[Bs // 2:, 32 + 14] = 0.8
    reps = (B + Bs - 1) // Bs
    xb, ub = np.tile(x0, (reps, 1))[:B], np.tile(ui, (reps, 1, 1))[:B]
    res = []
    for xx, uu in ((x0, ui), (xb, ub)):
        s = _solver(len(xx)); s.set_problem(prob); s.set_contact_mode(cm); s.set_friction(0.3); s.set_joint_limits(True)
        s.set_options(early_exit=False); s.set_max_iterations(4)
        s.initialize(xx, uu); cost = s.solve(xx)
        res.append((cost, s.trace()[0], s.gains_K(), s.xbar(), s.adopt_mismatches())); s.close()
    assert res[0][4] == 0 and res[1][4] == 0
    for k in range(4):
        assert np.array_equal(res[0][k], res[1][k][:Bs], equal_nan=True), k
        assert np.array_equal(res[0][k], res[1][k][B - B % Bs - Bs: B - B % Bs] if B % Bs else res[1][k][-Bs:], equal_nan=True), k
    o = ol.Oracle(prob["N"], prob["dt"]); o.set_problem(prob); o.set_options(max_iter=4, early_exit=0); o.set_contact_mode(cm); o.set_friction(0.3); o.set_joint_limits(True)
    o.initialize(x0[Bs - 1], ui[Bs - 1]); ok, c = o.solve(x0[Bs - 1])
    assert abs(res[0][0][Bs - 1] - c) <= 1e-5 * abs(c)


def test_cross_check_families_refuse_what_they_do_not_carry():
    """The joint-limit rows and the sliding-foot Jacobians exist on the two-lane / two-knot kernels only: the scalar dynamics family
    refuses the option, the one-knot tangent kernels refuse an analytic linearisation with it (and in contact modes 3 / 4) -- loudly,
    with the reason -- while forward differences stay available there."""
    from mpc_ilqr_mujoco_amd import solver as sv
    B = 2
    prob, x0, ui = standing(B, seed=3, gravity=[0.0, 0.0, -9.81])
    with env(ILQR_DYN="s"):
        s = sv.BatchedILQR(B, lib_path=sv.LEGACY_LIB_PATH); s.set_problem(prob)
        with pytest.raises(sv.ILQRError, match="joint-limit rows exist on the two-lane kernels only"):
            s.set_joint_limits(True)
        s.close()
    with env(ILQR_LINT="1"):
        s = sv.BatchedILQR(B, lib_path=sv.LEGACY_LIB_PATH); s.set_problem(prob); s.set_contact_mode(2); s.set_joint_limits(True)
        s.set_options(jacobian_mode=0); s.initialize(x0, ui)
        with pytest.raises(sv.ILQRError, match="joint-limit rows"):
            s.stage_linearize()
        s.set_joint_limits(False); s.set_contact_mode(3)
        with pytest.raises(sv.ILQRError, match="Coulomb limit"):
            s.stage_linearize()
        s.set_joint_limits(True)
        s.set_options(jacobian_mode=1, fd_eps=1e-5); s.stage_linearize()          # the reference's scheme runs
        A, Bm = s.linearization()
        assert np.all(np.isfinite(A)) and np.all(np.isfinite(Bm))
        s.close()


def test_forward_difference_jacobians_never_take_the_early_continuation():
    """the forward-difference launchers select rollouts by S.active, not by a group's work list, and k_fd_finish
    rewrites S.A / S.Bm in place -- a second group's pass would rewrite the Jacobians the retry's backward pass is reading.  The early
    continuation is therefore analytic-only (ilqr_capi.hip can_split): with the reference's Jacobian scheme, the convergence exit,
    ILQR_SPEC=0 and ILQR_SPLIT=1 no iteration is split, and the solve equals the ILQR_SPLIT=0 one and its own repetition bit for bit."""
    B = 12
    prob, x0, ui = standing(B, seed=53)
    out = {}
    for name, split in (("split", "1"), ("nosplit", "0"), ("again", "1")):
        with env(ILQR_SPEC="0", ILQR_SPLIT=split):
            s = _solver(B); s.set_problem(prob); s.set_options(jacobian_mode=1, fd_eps=1e-5, early_exit=True); s.set_max_iterations(10)
            s.initialize(x0, ui); cost = s.solve(x0)
            tc, ta, tl = s.trace()
            out[name] = (cost, tc, ta, tl, s.iterations(), s.gains_K(), s.gains_kff(), s.xbar(), s.ubar(), s.split_iterations())
            s.close()
    assert out["split"][9] == 0 and out["nosplit"][9] == 0
    for k in range(9):
        assert np.array_equal(out["split"][k], out["nosplit"][k], equal_nan=True) and np.array_equal(out["split"][k], out["again"][k], equal_nan=True), k
    o = ol.Oracle(prob["N"], prob["dt"]); o.set_problem(prob); o.set_options(max_iter=10, early_exit=1, jac_mode=1, fd_eps=1e-5)
    o.initialize(x0[0], ui[0]); ok, c = o.solve(x0[0])
    assert abs(out["split"][0][0] - c) <= 1e-5 * abs(c)


@pytest.mark.parametrize("early_exit", [False, True])
def test_skipping_saturated_lambda_retries_changes_no_observable(early_exit):
    """ilqr_hip_set_dedup_saturated_retry: once lambda sits at its cap (min(10 lambda, 1e-3) == lambda) the retry of ilqr.cpp:619-644
    repeats the failed pass on identical inputs; with the option on it is not executed and its bookkeeping is played at once.  Cost
    trace, step sizes, lambda schedule, iteration counts, gains, value function, trajectories: bit for bit the default solve's, on a
    batch where such retries do occur (ILQR_SPEC=0: the sequential orders are the ones that can skip)."""
    B = 12
    prob, x0, ui = standing(B, seed=53)
    out = {}
    for on in (0, 1):
        with env(ILQR_SPEC="0"):
            s = _solver(B); s.set_problem(prob); s.set_options(early_exit=early_exit); s.set_max_iterations(10)
            s.set_dedup_saturated_retry(bool(on))
            s.initialize(x0, ui); cost = s.solve(x0)
            tc, ta, tl = s.trace()
            Vx, Vxx = s.value_function()
            out[on] = (cost, tc, ta, tl, s.iterations(), s.gains_K(), s.gains_kff(), Vx, Vxx, s.xbar(), s.ubar(), s.lambdas())
            s.close()
    for k in range(12):
        assert np.array_equal(out[0][k], out[1][k], equal_nan=True), k
    # the batch does contain failed iterations entered with a saturated lambda (the ones whose retry is skipped)
    ta, tl, it = out[1][2], out[1][3], out[1][4]
    skipped = sum(1 for b in range(B) for i in range(1, it[b]) if ta[b, i] == 0.0 and tl[b, i] == 1e-3 and ta[b, i - 1] == 0.0 and tl[b, i - 1] == 1e-3)
    assert early_exit or skipped >= 3, skipped      # (with the convergence exit a failed iteration > 1 ends the rollout: ilqr.cpp:653-655)


def test_profiled_stage_mask_times_only_the_chosen_stages():
    """bench.py keeps event pairs only around the roofline kernel group inside its timed region (ilqr_hip_set_profiled_stages)."""
    B = 4
    prob, x0, ui = standing(B, seed=3)
    s = _solver(B); s.set_problem(prob); s.set_max_iterations(3); s.set_options(early_exit=False)
    s.enable_profiling(True)
    s.initialize(x0, ui); s.solve(x0)
    ms_all, n_all = s.stage_ms()
    assert all(n_all[k] > 0 and ms_all[k] > 0 for k in ("iLQR_backwardPass", "iLQR_lineSearch", "iLQR_linearization", "iLQR_costQuadratics"))
    s.set_profiled_stages(["iLQR_backwardPass", "iLQR_backwardPass_retry"])
    s.set_regularization(1e-6); s.initialize(x0, ui); cost = s.solve(x0)       # (lambda is state that survives a solve, as in the reference)
    ms, n = s.stage_ms()
    assert n["iLQR_backwardPass"] == 3 and n["iLQR_backwardPass_retry"] == 3 and ms["iLQR_backwardPass"] > 0
    assert all(n[k] == 0 and ms[k] == 0 for k in ms if not k.startswith("iLQR_backwardPass"))
    s.set_profiled_stages(None)
    s.set_regularization(1e-6); s.initialize(x0, ui); cost2 = s.solve(x0)
    assert np.array_equal(cost, cost2) and s.stage_ms()[1]["iLQR_lineSearch"] == 3
    s.close()


@pytest.mark.parametrize("ls", ["s", "r"])
def test_line_search_variants_match_oracle(ls):
    prob, x0, ui = standing(3, seed=4)
    with env(ILQR_LS=ls, ILQR_ROLLOUT=ls):
        s = _solver(3); s.set_problem(prob)
        s.initialize(x0, ui)
        s.stage_linearize(); s.stage_cost_quadratics(); s.stage_backward_pass()
        xb, ub = s.xbar(), s.ubar()
        A, Bm = s.linearization(); lx, lu, lxx, luu = s.quadratics()
        imp, cost, alpha = s.stage_line_search()
        xn = s.xbar()
        s.close()
    for b in range(3):
        o = ol.Oracle(prob["N"], prob["dt"]); o.set_problem(prob)
        o.set_trajectory(xb[b], ub[b]); o.set_linearization(A[b], Bm[b]); o.set_quadratics(lx[b], lu[b], lxx[b], luu[b])
        o.backward_pass()
        ok, c, a = o.line_search(x0[b])
        assert ok == bool(imp[b]) and a == alpha[b] and abs(c - cost[b]) < 1e-8 * abs(c) and rel(xn[b], o.get("xbar")) < 1e-8


def test_diverged_neighbour_does_not_poison_a_healthy_rollout():
    """Rollouts are independent (no cross-rollout term anywhere in ilqr.cpp).  The MFMA feedback of the two-lane line search pads
    K_t to 52 columns; the A operand of the phantom column is the first entry of the NEXT row of K -- at row 18 of the last knot
    that is the next ROLLOUT's K_0[0][0] -- so both operands are zeroed (0 * NaN would be NaN).  One rollout of the batch gets
    NaN gains (through NaN quadratics); every other rollout's line search must be bit-identical to the all-healthy run."""
    B = 8
    prob, x0, ui = standing(B, seed=23)
    s = _solver(B); s.set_problem(prob)
    s.initialize(x0, ui)
    s.stage_linearize(); s.stage_cost_quadratics()
    xb, ub = s.xbar(), s.ubar()
    lx, lu, lxx, luu = s.quadratics()
    s.stage_backward_pass()
    imp0, cost0, alpha0 = s.stage_line_search()
    xn0, un0 = s.xbar(), s.ubar()
    for bad in (1, 5):                      # bad - 1 is the rollout whose last-knot feedback reads past its own gains
        s.set_trajectory(xb, ub)
        lxx_bad = lxx.copy(); lxx_bad[bad, -1] = np.nan
        s.set_quadratics(lx, lu, lxx_bad, luu)
        s.stage_backward_pass()
        assert np.all(np.isnan(s.gains_K()[bad]))
        imp, cost, alpha = s.stage_line_search()
        xn, un = s.xbar(), s.ubar()
        ok = [b for b in range(B) if b != bad]
        assert not imp[bad]
        assert np.array_equal(imp[ok], imp0[ok]) and np.array_equal(cost[ok], cost0[ok]) and np.array_equal(alpha[ok], alpha0[ok])
        assert np.array_equal(xn[ok], xn0[ok]) and np.array_equal(un[ok], un0[ok])
    s.close()


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[1]: B = 1024 standing rollouts, N = 25, forward rollout + Jacobians only
def test_config1_rollout_and_jacobians_at_1024():
    B, N = 1024, 25
    prob, x0, ui = standing(B, N, seed=1)
    s = _solver(B, N=N); s.set_problem(prob)
    s.initialize(x0, ui)
    s.stage_linearize()
    A, Bm = s.linearization()
    xb, ub = s.xbar(), s.ubar()
    assert np.all(np.isfinite(A)) and np.all(np.isfinite(Bm)) and np.array_equal(xb[:, 0], x0)
    assert np.abs(np.linalg.norm(xb[:, :, 3:7], axis=2) - 1).max() < 1e-12
    # structure of the semi-implicit Euler step: hinge positions q' = q + h v'  =>  dq'/du = h dv'/du, dq'/dq = I + h dv'/dq
    h = prob["dt"]
    assert np.abs(Bm[:, :, 7:26, :] - h * Bm[:, :, 32:51, :]).max() < 1e-12
    assert np.abs(A[:, :, 0:3, 0:3] - np.eye(3) - h * A[:, :, 26:29, 0:3]).max() < 1e-12
    # the trajectory satisfies the dynamics: x_{t+1} = f(x_t, u_t) (same step through the stage API)
    for b in (0, 511, 1023):
        assert rel(s.step(xb[b, :N], ub[b]), xb[b, 1:]) < 1e-12        # (one-lane step kernel vs two-lane rollout kernel: rounding only)
    # oracle spot checks (forward-mode AD Jacobians) and a central-difference check through the GPU step itself
    for b in (0, 333, 1023):
        o = ol.Oracle(N, prob["dt"]); o.set_problem(prob); o.set_trajectory(xb[b], ub[b]); o.linearize()
        assert np.abs(A[b] - o.get("A")).max() < 1e-9 and np.abs(Bm[b] - o.get("B")).max() < 1e-9
        o2 = ol.Oracle(N, prob["dt"]); o2.set_problem(prob); o2.initialize(x0[b], ui[b])      # cold start = N rollout steps
        assert rel(xb[b], o2.get("xbar")) < 1e-11
    b, t, eps = 700, 12, 1e-6
    xp = np.repeat(xb[b, t][None], 2 * 51, axis=0); up = np.repeat(ub[b, t][None], 2 * 51, axis=0)
    for i in range(51):
        xp[2 * i, i] += eps; xp[2 * i + 1, i] -= eps
    f = s.step(xp, up)
    Afd = ((f[0::2] - f[1::2]) / (2 * eps)).T
    assert np.abs(Afd - A[b, t]).max() < 5e-7 * max(1.0, np.abs(A[b, t]).max())
    s.close()


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[4] at its per-GPU shape: B = 1024, N = 50, walking windows cut from the reference's walking file with
# per-rollout start rows t0, contact-scheduled cost terms
def walking_problem(B, N, seed):
    """scenario.walking_batch (shared with bench.py --workload config4) + the check that the contact-schedule tool reproduces the
    clearance signs stored with the fixture."""
    from mpc_ilqr_mujoco_amd import references as rf
    from mpc_ilqr_mujoco_amd import solver as sv
    r = np.load(os.path.join(G, "refdata_golden.npz"))
    flags = rf.contact_schedule(rf.pinocchio_to_mujoco(r["walking_pin_rows"]), sv.foot_clearance)
    assert np.array_equal(flags, (r["walking_pin_clearance"] < 0).astype(np.int32))
    return sc.walking_batch(B, N, seed, os.path.join(G, "refdata_golden.npz"), sv, rf)


def test_config4_walking_windows_at_per_gpu_shape():
    B, N = 1024, 50
    prob, x0, ui, t0 = walking_problem(B, N, seed=3)
    assert prob["stance"].shape == (B, N + 1, 2) and prob["stance"].min() == 0 and len(set(t0.tolist())) > 50
    s = _solver(B, N=N); s.set_problem(prob); s.set_max_iterations(4)
    s.initialize(x0, ui)
    cost = s.solve(x0)
    tc = s.trace()[0]; it = s.iterations(); xb = s.xbar(); K = s.gains_K()
    assert np.all(np.isfinite(cost)) and np.all(it >= 1) and np.all(it <= 4) and s.adopt_mismatches() == 0
    for b in range(B):
        assert np.all(np.diff(tc[b, : it[b] + 1]) <= 0)                   # accepted steps only ever decrease the cost
    # (knot 0 is the measured state: rows of the 6-digit CSV, normalised by the step, not before it)
    assert np.abs(np.linalg.norm(xb[:, 1:, 3:7], axis=2) - 1).max() < 1e-12 and np.array_equal(xb[:, 0], x0)
    s.close()
    # batch invariance: rollouts solved alone (their own reference sets) give bit-identical results
    pick = [0, 500, 1023]
    sub = dict(prob)
    for k in ("x_ref", "u_ref", "com_ref", "stance", "ee_ref", "com_vel_ref"):
        sub[k] = prob[k][pick]
    s2 = _solver(len(pick), N=N); s2.set_problem(sub); s2.set_max_iterations(4)
    s2.initialize(x0[pick], ui[pick]); c2 = s2.solve(x0[pick])
    assert np.array_equal(c2, cost[pick]) and np.array_equal(s2.gains_K(), K[pick])
    s2.close()
    # oracle spot checks on the per-rollout reference sets
    for b in (0, 777):
        o = ol.Oracle(N, prob["dt"]); o.set_problem(prob, b); o.set_options(max_iter=4)
        o.initialize(x0[b], ui[b]); ok, c = o.solve(x0[b])
        n, oc, oa, _ = o.trace()
        assert n == it[b] and np.allclose(tc[b, : n + 1], oc[: n + 1], rtol=1e-5) and rel(K[b], o.get("K")) < 1e-5


# ---------------------------------------------------------------------------------------------------------------------
# The GLOBAL sizes of the two 8-GPU configurations on ONE device (288 GB of HBM hold them: DESIGN section 2): what needs eight
# physical GPUs is only the placement of the shards -- a rollout's results do not depend on it (next section).
def _first_knot_rows(s, with_gains=True):
    """[u0 | cost | K0] rows through the product's own pack (the gather payload), without pulling K for the whole batch."""
    import torch
    from mpc_ilqr_mujoco_amd import sharding as sh
    rows = torch.zeros(s.B, sh.payload_width(with_gains), dtype=torch.float64, device="cuda:0")
    s.comm_init(1, 0)
    s.gather_first_knot(rows.data_ptr(), root=0, with_gains=with_gains); s.synchronize()
    s.comm_destroy()
    return rows.cpu().numpy()


def test_config3_global_batch_of_32768_on_one_device():
    """BASELINE.json configs[3]: 32768 rollouts, N = 25, full iLQR -- its whole global batch on one GPU (57 GB).  Properties
    at full size (finite, monotone cost traces, unit quaternions, iteration counts), batch invariance against the same rollouts
    solved in a small handle (bit for bit, first-knot payload rows), oracle spot checks."""
    B, N, iters = 32768, 25, 4
    prob, x0, ui = standing(B, N, seed=0)
    s = _solver(B, N=N); s.set_problem(prob); s.set_max_iterations(iters)
    s.initialize(x0, ui)
    cost = s.solve(x0)
    tc, ta, _ = s.trace(); it = s.iterations()
    assert cost.shape == (B,) and np.all(np.isfinite(cost)) and np.all(it >= 1) and np.all(it <= iters) and s.adopt_mismatches() == 0
    idx = np.arange(iters + 1)[None, :] <= it[:, None]
    d = np.diff(tc, axis=1)                                            # (NaN beyond the executed iterations: masked below)
    assert np.all(d[idx[:, 1:]] <= 0)                                 # accepted steps only ever decrease the cost
    stuck = np.where(~(cost < tc[:, 0]))[0]                             # a handful of cold starts never find an improving step:
    assert np.all(cost <= tc[:, 0]) and len(stuck) < B // 1000          # fail -> retry -> continue -> fail -> break (ilqr.cpp:619-644)
    assert len(stuck) >= 1 and np.all(it[stuck] == 3) and np.all(ta[stuck, :3] == 0.0)
    rows = _first_knot_rows(s)
    assert np.array_equal(rows[:, 19], cost) and np.all(np.isfinite(rows))
    xb_pick = None
    pick = [0, int(stuck[0]), 4095, 4096, 17000, 32767]                # a stuck rollout; first / last rollout of a rank's shard at 8 ranks; the batch's ends
    s.close()
    s2 = _solver(len(pick), N=N); s2.set_problem(prob); s2.set_max_iterations(iters)
    s2.initialize(x0[pick], ui[pick]); c2 = s2.solve(x0[pick])
    r2 = _first_knot_rows(s2)
    assert np.array_equal(c2, cost[pick]) and np.array_equal(r2, rows[pick])
    xb = s2.xbar(); K = s2.gains_K(); tc2 = s2.trace()[0]; it2 = s2.iterations()
    assert np.abs(np.linalg.norm(xb[:, :, 3:7], axis=2) - 1).max() < 1e-12
    s2.close()
    # 32 rollouts spread over the global batch on the oracle's OpenMP batch: final cost, iteration count, first control, first gain
    spot = [int(i) for i in np.linspace(0, B - 1, 32)]
    ob = ol.Oracle(N, prob["dt"]); ob.set_problem(prob); ob.set_options(max_iter=iters)
    _, oc32, oit32, ou0, oK0 = ob.batch_solve(x0[spot], ui[spot], nthreads=0, want_gains=True)
    assert np.array_equal(oit32, it[spot]) and np.allclose(oc32, cost[spot], rtol=1e-5, atol=0)
    assert rel(rows[spot, :19], ou0) < 1e-5 and rel(rows[spot, 20:].reshape(32, 19, 51), oK0) < 1e-5
    for j, b in enumerate(pick[:2]):
        o = ol.Oracle(N, prob["dt"]); o.set_problem(prob); o.set_options(max_iter=iters)
        o.initialize(x0[b], ui[b]); ok, c = o.solve(x0[b])
        n, oc, oa, _ = o.trace()
        assert n == it2[j] and np.allclose(tc2[j, : n + 1], oc[: n + 1], rtol=1e-5) and rel(K[j], o.get("K")) < 1e-5


def test_config4_global_batch_of_8192_walking_windows_contact_scheduled():
    """BASELINE.json configs[4]: 8192 rollouts, N = 50, reference windows cut from the reference's walking file with per-rollout
    start rows, contact-scheduled cost terms AND the schedule-driven unilateral stance constraints in the dynamics
    (set_contact_mode(2), row f4) -- the whole global batch on one GPU (28 GB)."""
    B, N, iters = 8192, 50, 3
    prob, x0, ui, t0 = walking_problem(B, N, seed=7)
    assert prob["stance"].shape == (B, N + 1, 2) and prob["stance"].min() == 0
    s = _solver(B, N=N); s.set_problem(prob); s.set_contact_mode(2); s.set_max_iterations(iters)
    s.initialize(x0, ui)
    cost = s.solve(x0)
    tc = s.trace()[0]; it = s.iterations()
    assert np.all(np.isfinite(cost)) and np.all(it >= 1) and np.all(it <= iters) and s.adopt_mismatches() == 0
    idx = np.arange(iters + 1)[None, :] <= it[:, None]
    d = np.diff(tc, axis=1)                                            # (NaN beyond the executed iterations: masked below)
    assert np.all(d[idx[:, 1:]] <= 0)
    rows = _first_knot_rows(s)
    assert np.array_equal(rows[:, 19], cost) and np.all(np.isfinite(rows))
    s.close()
    pick = [0, 1023, 1024, 8191]
    sub = dict(prob)
    for k in ("x_ref", "u_ref", "com_ref", "stance", "ee_ref", "com_vel_ref"):
        sub[k] = prob[k][pick]
    s2 = _solver(len(pick), N=N); s2.set_problem(sub); s2.set_contact_mode(2); s2.set_max_iterations(iters)
    s2.initialize(x0[pick], ui[pick]); c2 = s2.solve(x0[pick])
    r2 = _first_knot_rows(s2)
    assert np.array_equal(c2, cost[pick]) and np.array_equal(r2, rows[pick])
    K = s2.gains_K(); tc2 = s2.trace()[0]; it2 = s2.iterations(); xb = s2.xbar()
    assert np.abs(np.linalg.norm(xb[:, 1:, 3:7], axis=2) - 1).max() < 1e-12
    s2.close()
    b = pick[2]
    o = ol.Oracle(N, prob["dt"]); o.set_problem(prob, b); o.set_contact_mode(2); o.set_options(max_iter=iters)
    o.initialize(x0[b], ui[b]); ok, c = o.solve(x0[b])
    n, oc, oa, _ = o.trace()
    assert n == it2[2] and np.allclose(tc2[2, : n + 1], oc[: n + 1], rtol=1e-5) and rel(K[2], o.get("K")) < 1e-5


# ---------------------------------------------------------------------------------------------------------------------
# multi-rank: two ranks (both on device 0, payload staged through a gloo gather) vs one process solving the global batch
_WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from conftest import load_package
pkg = load_package()
from mpc_ilqr_mujoco_amd import sharding as sh, solver as sv
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
d = np.load(%(inp)r)
G = d["x0"].shape[0]
lo, hi = sh.shard_range(G, rank, world)
prob = {k[2:]: d[k] for k in d.files if k.startswith("p_")}
prob["task_weights"] = tuple(prob["task_weights"]); prob["gravity"] = tuple(prob["gravity"])
for k in ("N", "dt", "w_joint", "w_ctrl"):
    prob[k] = prob[k].item()
s = sv.BatchedILQR(hi - lo, N=prob["N"], dt=prob["dt"], device=0)
s.set_problem(prob); s.set_max_iterations(3)
s.initialize(d["x0"][lo:hi], d["ui"][lo:hi]); s.solve(d["x0"][lo:hi])
# the C ABI's own gather in its one-rank form packs the payload rows [u0 | cost | K0] on the device ...
s.comm_init(1, 0)
rows = torch.zeros(hi - lo, sh.payload_width(True), dtype=torch.float64, device="cuda:0")
s.gather_first_knot(rows.data_ptr(), root=0, with_gains=True); s.synchronize()
# ... and the exchange between the ranks is staged through the host here (two ranks share one GPU, which RCCL refuses)
g = sh.gather_first_knot(rows.cpu(), dst=0)
if rank == 0:
    np.save(%(out)r, g.numpy())
dist.barrier(); dist.destroy_process_group(); s.close()
"""


def test_two_rank_shard_gather_equals_single_process_bitwise(tmp_path):
    from mpc_ilqr_mujoco_amd import sharding as sh
    Gb = 16
    prob, x0, ui = standing(Gb, seed=9)
    inp, out = str(tmp_path / "in.npz"), str(tmp_path / "g.npy")
    np.savez(inp, x0=x0, ui=ui, **{"p_" + k: np.asarray(v) for k, v in prob.items()})
    script = str(tmp_path / "worker.py")
    open(script, "w").write(_WORKER % dict(root=ROOT, inp=inp, out=out))
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), script], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    g = np.load(out)
    import torch
    u0, cost, K0 = sh.unpack_payload(torch.from_numpy(g), True)
    s = _solver(Gb); s.set_problem(prob); s.set_max_iterations(3)
    s.initialize(x0, ui); c = s.solve(x0)
    assert np.array_equal(cost.numpy(), c) and np.array_equal(u0.numpy(), s.ubar()[:, 0]) and np.array_equal(K0.numpy(), s.gains_K()[:, 0])
    s.close()


@pytest.mark.parametrize("try_rccl", [False, True])
def test_bench_two_ranks_rehearsed_on_one_gpu(try_rccl):
    """The world > 1 branch of bench.py (global batch sliced by shard_range, barrier, max-over-ranks timing, gather) executed
    with two ranks on device 0 over gloo (--rehearse-single-gpu; on a multi-GPU node the same branch gathers over RCCL through
    the C ABI).  With --try-rccl the RCCL communicator of the C ABI is attempted for real: ilqr_hip_comm_get_unique_id, the
    128-byte id over the process group, ilqr_hip_comm_init(world = 2) on both ranks -- on ONE device RCCL refuses the second
    rank, every rank agrees on the fallback and the line says so (what would happen on a node whose RCCL does not come up)."""
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "64", "--steps", "1", "--warmup", "0",
                        "--rehearse-single-gpu"] + (["--try-rccl"] if try_rccl else []), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["value"] > 0 and "roofline" in d and "cpu_baseline" not in d
    assert d["config"]["gather_check"] == "rank 0 received 128 rows in global rollout order"
    assert d["config"]["workload"].startswith("custom:")
    if try_rccl:
        # either RCCL accepted two ranks on one device (then its gather carried the rows) or both ranks fell back together
        assert ("RCCL grouped send/recv" in d["config"]["collective"] and d["config"]["collective_error"] is None) or \
               ("gloo" in d["config"]["collective"] and d["config"]["collective_error"])
    else:
        assert "gloo" in d["config"]["collective"] and d["config"]["collective_error"] is None


def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2 --rehearse-single-gpu` started PLAINLY (no torch.distributed.run around it, the way the driver
    starts `--gpus 1`): the parent never touches the GPU, starts the two ranks as a child process and relays rank 0's line --
    `n_gpus: 2`, the global batch of both ranks gathered, the launcher named in the line (it used to time one rank)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "64", "--steps", "1", "--warmup", "0", "--rehearse-single-gpu"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["value"] > 0
    assert d["config"]["gather_check"] == "rank 0 received 128 rows in global rollout order"
    assert d["config"]["launcher"].startswith("self-launched") and "gloo" in d["config"]["barrier"]


# ---------------------------------------------------------------------------------------------------------------------
# C++ drop-in sharding without Python: the demo's one-rank form against the Python wrapper on the same rollouts
def _demo_rollout(g, N):
    M = (1 << 64) - 1
    st = [(0x9E3779B97F4A7C15 * (g + 1)) & M]

    def rnd():
        s = st[0]
        s ^= (s << 13) & M; s ^= s >> 7; s ^= (s << 17) & M
        st[0] = s
        return (s >> 11) / 9007199254740992.0 * 2.0 - 1.0
    x0 = np.zeros(51); x0[2] = 1.0432
    for i in range(3):
        x0[i] += 0.02 * rnd()
    w = np.array([0.05 * rnd(), 0.05 * rnd(), 0.05 * rnd()])
    ang = float(np.sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2])); shf = np.sin(0.5 * ang) / ang if ang > 1e-12 else 0.5
    x0[3] = np.cos(0.5 * ang); x0[4:7] = shf * w
    for i in range(7, 26):
        x0[i] = 0.05 * rnd()
    for i in range(26, 51):
        x0[i] = 0.1 * rnd()
    u = np.array([[rnd() for _ in range(19)] for _ in range(N)])
    return x0, u


def test_cpp_multi_gpu_demo_one_rank(tmp_path):
    from mpc_ilqr_mujoco_amd import solver as sv
    exe = os.path.join(ROOT, "tests", "cpp", "build", "cpp_multi_gpu_demo")
    assert os.path.exists(exe), "built by __graft_entry__.build()"
    B, N = 6, 25
    out = str(tmp_path / "rows.bin")
    r = subprocess.run([exe, "1", str(B), "1", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    rows = np.fromfile(out).reshape(B, 20 + 19 * 51)
    prob = sc.make_problem(sv.reference_kinematics, N=N)
    xs, us = zip(*[_demo_rollout(g, N) for g in range(B)])
    x0, ui = np.array(xs), np.array(us)
    s = _solver(B); s.set_problem(prob); s.set_max_iterations(3)
    s.initialize(x0, ui); c = s.solve(x0)
    assert rel(rows[:, 19], c) < 1e-9 and rel(rows[:, :19], s.ubar()[:, 0]) < 1e-8 and rel(rows[:, 20:].reshape(B, 19, 51), s.gains_K()[:, 0]) < 1e-7
    s.close()


def test_product_library_holds_the_default_kernel_family_only():
    """The cross-check families (scalar / one-lane dynamics, VALU and four-wave Riccati, the folded Riccati kernel on the standard layout,
    one-knot tangent kernels) live in lib/libilqr_hip_legacy.so (-DILQR_LEGACY_KERNELS, tests only).  The product library refuses a
    handle whose environment selects one of them -- loudly, with the reason -- instead of silently running something else."""
    from mpc_ilqr_mujoco_amd import solver as sv
    for var in (dict(ILQR_BACKWARD="valu"), dict(ILQR_BACKWARD="wg"), dict(ILQR_BACKWARD="wave-fold"), dict(ILQR_LS="r"), dict(ILQR_ROLLOUT="r"), dict(ILQR_DYN="s"), dict(ILQR_LINT="1")):
        with env(**var):
            with pytest.raises(sv.ILQRError, match="cross-check kernel family"):
                sv.BatchedILQR(2)
            s = sv.BatchedILQR(2, lib_path=sv.LEGACY_LIB_PATH); s.close()      # the test library takes it
    with env(ILQR_BACKWARD="wave-generic"):
        s = sv.BatchedILQR(2); s.close()                                       # (the generic one-wave kernel is product code: forward-difference Jacobians run on it)
    assert os.path.getsize(sv.LIB_PATH) < 0.8 * os.path.getsize(sv.LEGACY_LIB_PATH)      # (both carry the sliding-foot and joint-limit instantiations of round 5)


def _demo_and_fake_rccl():
    exe = os.path.join(ROOT, "tests", "cpp", "build", "cpp_multi_gpu_demo")
    fake = os.path.join(ROOT, "tests", "cpp", "build", "libfake_rccl.so")
    assert os.path.exists(exe) and os.path.exists(fake), "built by __graft_entry__.build()"
    return exe, fake


@pytest.mark.parametrize("root", [0, 1])
def test_world_2_gather_branch_of_the_c_abi_runs_on_one_device_through_a_stand_in_rccl(tmp_path, root):
    """ilqr_hip_gather_first_knot with world > 1 (ilqr_capi.hip: ncclGroupStart, the root's ncclRecv loop with receive offsets r * cnt,
    the peers' ncclSend, ncclGroupEnd) cannot run under real RCCL on a one-GPU box (it refuses a second rank on the same device).  A
    test-only stand-in library (tests/cpp/fake_rccl.cpp: the eight symbols the loader resolves, send / recv between two
    communicators of one process as stream-ordered device copies), selected with ILQR_RCCL_LIB, lets tests/cpp/cpp_multi_gpu_demo
    run it with two host threads and two handles on device 0: the root's [2 B][989] rows equal the single-handle solve of the whole
    batch bit for bit, for either root.  The consumer of these rows in the reference: MPC::stepOnce (src/ilqr/mpc.cpp:97-113)."""
    exe, fake = _demo_and_fake_rccl()
    B, W = 5, 20 + 19 * 51
    two, one = str(tmp_path / "two.bin"), str(tmp_path / "one.bin")
    r = subprocess.run([exe, "2", str(B), "1", two, "0", str(root)], capture_output=True, text=True, timeout=600, env=dict(os.environ, ILQR_RCCL_LIB=fake))
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "1", str(2 * B), "1", one, "0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    rows2, rows1 = np.fromfile(two).reshape(2 * B, W), np.fromfile(one).reshape(2 * B, W)
    assert np.all(np.isfinite(rows1)) and np.abs(rows1[:, 20:]).max() > 0
    assert np.array_equal(rows2, rows1)
    # (without gains: the 160-byte payload rows)
    r = subprocess.run([exe, "2", str(B), "0", two, "0", str(root)], capture_output=True, text=True, timeout=600, env=dict(os.environ, ILQR_RCCL_LIB=fake))
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.fromfile(two).reshape(2 * B, 20), rows1[:, :20])


@pytest.mark.parametrize("root,with_gains", [(0, 1), (5, 1), (5, 0)])
def test_world_8_gather_on_one_device_at_the_width_of_the_node(tmp_path, root, with_gains):
    """The machine has eight GPUs; code that has only run at world = 2 is not ready for 8.  Eight host threads, eight handles and eight
    stand-in communicators on device 0 (ONE process: the GPU box admits six processes on its card, so eight bench ranks cannot be
    rehearsed there): the root's receive offsets r * cnt for r = 0..7, a root in the middle of the ranks, with and without K0 -- the
    gathered [8 B][width] rows equal the single-handle solve of the whole batch bit for bit.  Consumer: mpc.cpp:97-113."""
    exe, fake = _demo_and_fake_rccl()
    B, W = 3, (20 + 19 * 51) if with_gains else 20
    eight, one = str(tmp_path / "eight.bin"), str(tmp_path / "one.bin")
    r = subprocess.run([exe, "8", str(B), str(with_gains), eight, "0", str(root)], capture_output=True, text=True, timeout=900, env=dict(os.environ, ILQR_RCCL_LIB=fake))
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "1", str(8 * B), str(with_gains), one, "0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    rows8, rows1 = np.fromfile(eight).reshape(8 * B, W), np.fromfile(one).reshape(8 * B, W)
    assert np.all(np.isfinite(rows1)) and np.abs(rows1[:, :19]).max() > 0
    assert np.array_equal(rows8, rows1)
    # every shard is somebody else's rollouts: no two blocks of B rows coincide (a wrong receive offset would repeat or drop one)
    assert len({rows8[k * B:(k + 1) * B].tobytes() for k in range(8)}) == 8


def test_bench_four_ranks_without_a_launcher_on_one_gpu():
    """bench.py's self-launcher and its world > 1 branch above world = 2: `python bench.py --gpus 4 --batch 64 --rehearse-single-gpu`
    started plainly -- four ranks on device 0 over gloo (with this pytest process that is five processes on the card; the box admits
    six, which is why the eight-rank case is not rehearsed here), shard_range for ranks 0..3, global batch 256 gathered in global
    rollout order, one line labelled n_gpus: 4."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--batch", "64", "--steps", "1", "--warmup", "0", "--rehearse-single-gpu"],
                       capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["config"]["global_batch"] == 256 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["gather_check"] == "rank 0 received 256 rows in global rollout order"
    assert d["config"]["launcher"].startswith("self-launched")


def test_an_rccl_error_inside_the_gather_group_closes_the_group_and_surfaces_through_last_error(tmp_path):
    """An ncclRecv that fails inside the group (injected by the stand-in library) still has the group closed (ncclGroupEnd is called,
    the peer's ncclGroupEnd returns instead of waiting forever) and comes back as ILQR_ERR_HIP with the RCCL error text in
    ilqr_hip_last_error; the demo prints it and exits non-zero."""
    exe, fake = _demo_and_fake_rccl()
    out = str(tmp_path / "never.bin")
    r = subprocess.run([exe, "2", "3", "0", out, "0"], capture_output=True, text=True, timeout=600, env=dict(os.environ, ILQR_RCCL_LIB=fake, FAKE_RCCL_FAIL_RECV="1"))
    assert r.returncode == 1 and not os.path.exists(out)
    assert "RCCL gather (grouped send/recv)" in r.stderr and "injected" in r.stderr, r.stderr
    assert "rank 0: gather" in r.stderr and "rank 1: gather" in r.stderr, r.stderr      # the peer's group end reported the remote failure
