"""The product's own environment rule under pytest: the diagnostic switches are read ONCE per handle, by ilqr_hip_create, and a handle
whose environment selects a kernel family the product library does not hold is refused -- at creation, by ilqr_hip_reload_environment,
and (ILQR_ENV_PER_CALL handles) by the next call that would launch kernels.  tests/conftest.py sets ILQR_ENV_PER_CALL=1 for the rest of the
suite; every handle here is created with the variable removed, i.e. the way smoke(), bench.py and an application create theirs."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from conftest import load_package

pytestmark = pytest.mark.gpu
pkg = load_package()
sc = pkg.scenario


class env:
    """set / remove (value None) environment variables for a block"""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _problem(B, seed):
    from mpc_ilqr_mujoco_amd import solver as sv
    prob = sc.make_problem(sv.reference_kinematics, N=25)
    ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
    x0, ui = sc.synthetic_batch(B, 25, seed, ug)
    return prob, x0, ui


def _solve(s, prob, x0, ui, iters=3):
    s.set_problem(prob); s.set_max_iterations(iters); s.set_options(early_exit=False)
    s.set_regularization(1e-6)      # (lambda persists from solve to solve, ilqr.cpp:16, 620, 646: every run here starts from the default)
    s.initialize(x0, ui)
    cost = s.solve(x0)
    return cost, s.trace()[0], s.gains_K()


def test_frozen_handle_ignores_later_environment_and_matches_oracle():
    from mpc_ilqr_mujoco_amd import solver as sv
    B = 3
    prob, x0, ui = _problem(B, 11)
    with env(ILQR_ENV_PER_CALL=None):
        s = sv.BatchedILQR(B)                       # the product library, environment frozen here
    c0, t0, K0 = _solve(s, prob, x0, ui)
    # a cross-check family selected AFTER creation: nothing is re-read, the default family keeps running, results bit-identical
    with env(ILQR_BACKWARD="valu", ILQR_LS="r", ILQR_DYN="s"):
        c1, t1, K1 = _solve(s, prob, x0, ui)
        assert np.array_equal(c0, c1) and np.array_equal(t0, t1) and np.array_equal(K0, K1)
        # ... and asking for the re-read is refused loudly; the handle keeps its family and keeps working
        with pytest.raises(sv.ILQRError, match="kernel family"):
            s.reload_environment()
        c2, t2, K2 = _solve(s, prob, x0, ui)
        assert np.array_equal(c0, c2) and np.array_equal(K0, K2)
        # a NEW handle under that environment is refused at creation
        with env(ILQR_ENV_PER_CALL=None):
            with pytest.raises(sv.ILQRError, match="cross-check kernel family"):
                sv.BatchedILQR(B)
    assert s.adopt_mismatches() == 0
    for b in range(B):
        o = ol.Oracle(25, prob["dt"]); o.set_problem(prob); o.set_options(max_iter=3, early_exit=0)
        o.initialize(x0[b], ui[b]); ok, c = o.solve(x0[b])
        n, oc, oa, olam = o.trace()
        assert n == 3 and np.allclose(t0[b], oc, rtol=1e-5, atol=0) and abs(c0[b] - c) <= 1e-5 * abs(c)
        assert np.abs(K0[b] - o.get("K")).max() <= 1e-5 * np.abs(o.get("K")).max()
    s.close()


def test_supported_switches_are_taken_at_creation_and_only_there():
    """ILQR_SPEC (side-by-side lambda retry, B <= 512) read at creation: a handle created with it off never speculates, whatever the
    environment says later; a handle created with the default does, even with ILQR_SPEC=0 set afterwards; reload_environment re-reads."""
    from mpc_ilqr_mujoco_amd import solver as sv
    B = 4
    prob, x0, ui = _problem(B, 12)
    with env(ILQR_ENV_PER_CALL=None, ILQR_SPEC="0"):
        s_off = sv.BatchedILQR(B)
    with env(ILQR_ENV_PER_CALL=None, ILQR_SPEC=None):
        s_on = sv.BatchedILQR(B)
    with env(ILQR_SPEC="1"):
        c_off, t_off, K_off = _solve(s_off, prob, x0, ui)
        assert s_off.speculative_iterations() == 0
    with env(ILQR_SPEC="0"):
        c_on, t_on, K_on = _solve(s_on, prob, x0, ui)
        assert s_on.speculative_iterations() == 3
        # (both launch orders: bit-identical observables)
        assert np.array_equal(c_on, c_off) and np.array_equal(t_on, t_off) and np.array_equal(K_on, K_off)
        s_on.reload_environment()
        _solve(s_on, prob, x0, ui)
        assert s_on.speculative_iterations() == 0
    s_off.close(); s_on.close()


def test_per_call_handle_refuses_a_family_the_product_library_does_not_hold():
    """ILQR_ENV_PER_CALL=1 (the test suite's mode) on the PRODUCT library: switching to a cross-check family around a call no longer passes
    vacuously on the default family -- the call that would launch kernels returns ILQR_ERR_UNSUPPORTED; back on a held family it runs."""
    from mpc_ilqr_mujoco_amd import solver as sv
    B = 2
    prob, x0, ui = _problem(B, 13)
    with env(ILQR_ENV_PER_CALL="1"):
        s = sv.BatchedILQR(B)
        c0, t0, K0 = _solve(s, prob, x0, ui, iters=2)
        with env(ILQR_BACKWARD="valu"):
            with pytest.raises(sv.ILQRError, match="kernel family"):
                s.solve(x0)
            with pytest.raises(sv.ILQRError, match="kernel family"):
                s.stage_backward_pass()
        c1, t1, K1 = _solve(s, prob, x0, ui, iters=2)
        assert np.array_equal(c0, c1) and np.array_equal(K0, K1)
        s.close()
