"""ctypes binding of the CPU oracle (oracle/build/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NX, NU, NQ, NV = 51, 19, 26, 25
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def build_oracle():
    so = os.path.join(ROOT, "oracle", "build", "liboracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("ilqr_oracle.cpp", "h1_costs.hpp", "h1_dynamics.hpp", "ad.hpp", "h1_model_data.h")]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return so


_lib = None
_lib_native = None
NATIVE_FLAGS = "-O3 -march=native"
PORTABLE_FLAGS = "-O3"


def _declare(L):
    L.orc_create.restype = C.c_void_p
    L.orc_create.argtypes = [C.c_int, C.c_double]
    L.orc_total_cost.restype = C.c_double
    L.orc_get_lambda.restype = C.c_double
    L.orc_batch_solve.restype = C.c_long
    return L


def lib():
    global _lib
    if _lib is None:
        _lib = _declare(C.CDLL(build_oracle()))
    return _lib


_SELFTEST = """
import ctypes as C, sys
import numpy as np
L = C.CDLL(sys.argv[1])
L.orc_create.restype = C.c_void_p; L.orc_create.argtypes = [C.c_int, C.c_double]; L.orc_batch_solve.restype = C.c_long
h = C.c_void_p(L.orc_create(6, C.c_double(0.02)))
dp = C.POINTER(C.c_double)
x0 = np.zeros((2, 51)); x0[:, 2] = 1.0432; x0[:, 3] = 1.0; x0[1, 7:26] = 0.05
cost = np.zeros(2); it = np.zeros(2, dtype=np.int32); u0 = np.zeros((2, 19))
n = L.orc_batch_solve(h, 2, x0.ctypes.data_as(dp), None, cost.ctypes.data_as(dp), it.ctypes.data_as(C.POINTER(C.c_int)), u0.ctypes.data_as(dp), None, 2)
assert n > 0 and np.all(np.isfinite(cost))
print("ok")
"""


def lib_native():
    """The TIMED copy of the restatement (bench.py's cpu_baseline leg): the same sources built host-tuned ON THIS HOST
    (oracle/Makefile; never shipped between machines).  Candidates in order: `-O3 -march=native`, `-O3 -march=x86-64-v3`; each
    is run once in a CHILD process on a tiny problem before it is loaded here (a build that crashes on this CPU -- seen with
    g++ 11.4 -march=native on an EPYC 9575F -- must not take the bench down).  Returns (library, flags string); falls back to
    the portable checker build, and says so, when no host-tuned build works."""
    global _lib_native
    if _lib_native is None:
        import sys
        notes = []
        for tag, flags in (("native", "-O3 -march=native"), ("v3", "-O3 -march=x86-64-v3")):
            so = os.path.join(ROOT, "oracle", "build", "liboracle_%s.so" % tag)
            try:
                subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "build/liboracle_%s.so" % tag], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                r = subprocess.run([sys.executable, "-c", _SELFTEST, so], capture_output=True, text=True, timeout=300)
                if r.returncode != 0 or "ok" not in r.stdout:
                    raise RuntimeError("self-test of the build exited with code %d" % r.returncode)
                _lib_native = (_declare(C.CDLL(so)), "g++ %s -fopenmp (built and self-tested on this host%s)" % (flags, "; " + "; ".join(notes) if notes else ""))
                break
            except Exception as e:  # noqa: BLE001
                notes.append("%s: %s" % (flags, e))
        if _lib_native is None:
            _lib_native = (lib(), "g++ " + PORTABLE_FLAGS + " -fopenmp (portable checker build; " + "; ".join(notes) + ")")
    return _lib_native


def c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    """One-rollout CPU restatement of iLQR (mirrors iLQR's API, ilqr.hpp:19-45)."""

    def __init__(self, N=25, dt=0.02, native=False):
        self.N = N
        self.L = lib_native()[0] if native else lib()
        self.h = C.c_void_p(self.L.orc_create(N, C.c_double(dt)))

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    # --- problem data ---
    def set_problem(self, prob, b=0):
        """prob: dict from scenario.make_problem(); b selects the reference set when per-rollout."""
        L, h = self.L, self.h
        L.orc_set_cost_weights(h, _p(c64(prob["Q"])), _p(c64(prob["R"])), _p(c64(prob["Qf"])))
        w = prob["task_weights"]
        L.orc_set_task_weights(h, *[C.c_double(float(v)) for v in w])
        L.orc_set_constraint_weights(h, C.c_double(prob["w_joint"]), C.c_double(prob["w_ctrl"]))
        g = prob["gravity"]
        L.orc_set_gravity(h, C.c_double(g[0]), C.c_double(g[1]), C.c_double(g[2]))

        def pick(a):
            a = np.asarray(a)
            return a[b] if a.shape[0] > 1 else a[0]

        self._keep = [c64(pick(prob["x_ref"])), c64(pick(prob["u_ref"])), c64(pick(prob["com_ref"])),
                      np.ascontiguousarray(pick(prob["stance"]), dtype=np.int32), c64(pick(prob["ee_ref"])), c64(pick(prob["com_vel_ref"]))]
        L.orc_set_references(h, _p(self._keep[0]), _p(self._keep[1]), _p(self._keep[2]))
        L.orc_set_contact_schedule(h, self._keep[3].ctypes.data_as(_ip))
        L.orc_set_ee_references(h, _p(self._keep[4]), _p(self._keep[5]))

    def set_options(self, lam=1e-6, max_iter=10, tol=1e-4, jac_mode=0, fd_eps=1e-5, quad_mode=0, early_exit=1):
        self.L.orc_set_options(self.h, C.c_double(lam), int(max_iter), C.c_double(tol), int(jac_mode), C.c_double(fd_eps), int(quad_mode), int(early_exit))
        self.max_iter = max_iter

    def set_contact_mode(self, mode, soft=0.0):
        """0: constraint-free step; 1: rigid stance constraints on the feet the contact schedule marks;
        2: unilateral; 3: unilateral + Coulomb limit (set_friction)."""
        self.L.orc_set_contact_mode(self.h, int(mode), C.c_double(soft))

    def set_friction(self, mu):
        """Sliding friction coefficient of contact mode 3 (unilateral + Coulomb release)."""
        self.L.orc_set_friction(self.h, C.c_double(mu))

    def set_joint_limit_stiffness(self, k):
        self.L.orc_set_joint_limit_stiffness(self.h, C.c_double(float(k)))

    def set_joint_limits(self, on):
        """Joint-limit rows of the plant: a hinge past its range that the step would still move outward is stopped (h1_step)."""
        self.L.orc_set_joint_limits(self.h, int(bool(on)))

    def step_stance(self, x, u, stance):
        x, u = c64(x), c64(u)
        st = np.ascontiguousarray(stance, dtype=np.int32)
        xn = np.zeros(NX)
        self.L.orc_step_stance(self.h, _p(x), _p(u), st.ctypes.data_as(_ip), _p(xn))
        return xn

    def get_lambda(self):
        return self.L.orc_get_lambda(self.h)

    # --- solver ---
    def initialize(self, x0, u_init=None, prev_xbar=None, prev_ubar=None):
        ks = [c64(x0), None if u_init is None else c64(u_init), None if prev_xbar is None else c64(prev_xbar), None if prev_ubar is None else c64(prev_ubar)]
        self.L.orc_initialize(self.h, _p(ks[0]), _p(ks[1]), _p(ks[2]), _p(ks[3]))

    def set_trajectory(self, xbar, ubar):
        a, b = c64(xbar), c64(ubar)
        self.L.orc_set_trajectory(self.h, _p(a), _p(b))

    def solve(self, x0):
        x0 = c64(x0)
        c = C.c_double(0)
        rc = self.L.orc_solve(self.h, _p(x0), C.byref(c))
        return rc == 0, c.value

    def trace(self):
        n = self.max_iter
        cost, alpha, lam = np.full(n + 1, np.nan), np.full(n, np.nan), np.full(n, np.nan)
        it = self.L.orc_get_trace(self.h, _p(cost), _p(alpha), _p(lam))
        return it, cost, alpha, lam

    # --- stages ---
    def step(self, x, u):
        x, u = c64(x), c64(u)
        xn = np.zeros(NX)
        self.L.orc_step(self.h, _p(x), _p(u), _p(xn))
        return xn

    def rollout(self):
        self.L.orc_rollout(self.h)

    def linearize(self):
        self.L.orc_linearize(self.h)

    def cost_quadratics(self):
        self.L.orc_cost_quadratics(self.h)

    def backward_pass(self):
        self.L.orc_backward_pass(self.h)

    def line_search(self, x0):
        x0 = c64(x0)
        c, a = C.c_double(0), C.c_double(0)
        ok = self.L.orc_line_search(self.h, _p(x0), C.byref(c), C.byref(a))
        return bool(ok), c.value, a.value

    def total_cost(self):
        return self.L.orc_total_cost(self.h)

    def grav_comp(self, x):
        x = c64(x)
        u = np.zeros(NU)
        self.L.orc_grav_comp(self.h, _p(x), _p(u))
        return u

    def set_linearization(self, A, B):
        a, b = c64(A), c64(B)
        self.L.orc_set_linearization(self.h, _p(a), _p(b))

    def set_quadratics(self, lx, lu, lxx, luu):
        ks = [c64(lx), c64(lu), c64(lxx), c64(luu)]
        self.L.orc_set_quadratics(self.h, *[_p(k) for k in ks])

    def knot_quadratics(self, t, x, u, mode):
        x = c64(x)
        u = c64(u if u is not None else np.zeros(NU))
        lx, lu, lxx, luu = np.zeros(NX), np.zeros(NU), np.zeros((NX, NX)), np.zeros(NU)
        self.L.orc_knot_quadratics(self.h, int(t), _p(x), _p(u), int(mode), _p(lx), _p(lu), _p(lxx), _p(luu))
        return lx, lu, lxx, luu

    def compute_control(self, x_meas):
        x = c64(x_meas)
        u = np.zeros(NU)
        self.L.orc_compute_control(self.h, _p(x), _p(u))
        return u

    def get(self, name):
        N = self.N
        shapes = dict(xbar=(N + 1, NX), ubar=(N, NU), K=(N, NU, NX), kff=(N, NU), A=(N, NX, NX), B=(N, NX, NU),
                      lx=(N + 1, NX), lu=(N, NU), lxx=(N + 1, NX, NX), luu=(N, NU), Vx=(NX,), Vxx=(NX, NX))
        out = np.zeros(shapes[name])
        getattr(self.L, "orc_get_" + name)(self.h, _p(out))
        return out

    def op_counts(self, t, x, u):
        """Algorithmic flop counts of SURVEY.md 8(d) on the oracle's own code (oracle/opcount.cpp): dict with the flops of one
        dynamics step, one knot of one line-search trial, the cost quadratics of knot t and one analytic (A_t, B_t), plus the
        sin / cos evaluations among them."""
        out = np.zeros(8)
        x, u = c64(x), c64(u)
        self.L.orc_op_counts(self.h, int(t), _p(x), _p(u), _p(out))
        keys = ("step", "alpha_trial_knot", "quadratics_knot", "jacobians")
        return {k: int(round(v)) for k, v in zip(keys, out[:4])}, {k: int(round(v)) for k, v in zip(keys, out[4:])}

    def batch_solve(self, x0, u_init, nthreads=0, want_gains=False):
        x0 = c64(x0)
        B = x0.shape[0]
        ui = None if u_init is None else c64(u_init)
        cost, iters = np.zeros(B), np.zeros(B, dtype=np.int32)
        u0 = np.zeros((B, NU))
        K0 = np.zeros((B, NU, NX)) if want_gains else None
        total = self.L.orc_batch_solve(self.h, int(B), _p(x0), _p(ui), _p(cost), iters.ctypes.data_as(_ip), _p(u0), _p(K0), int(nthreads))
        return total, cost, iters, u0, K0


def joint_ranges():
    """The model table's hinge ranges (h1.xml jnt_range), [19][2]."""
    out = np.zeros((19, 2))
    lib().orc_joint_ranges(_p(out))
    return out


def reference_kinematics(x):
    x = c64(x)
    com, ee = np.zeros(3), np.zeros((2, 3))
    lib().orc_reference_kinematics(_p(x), _p(com), _p(ee))
    return com, ee


def forward_dynamics(x, tau, arm_eff, grav):
    x, tau, grav = c64(x), c64(tau), c64(grav)
    qacc = np.zeros(NV)
    lib().orc_forward_dynamics(_p(x), _p(tau), C.c_double(arm_eff), _p(grav), _p(qacc))
    return qacc


def inverse_dynamics(x, qacc, arm, grav):
    x, qacc, grav = c64(x), c64(qacc), c64(grav)
    tau = np.zeros(NV)
    lib().orc_inverse_dynamics(_p(x), _p(qacc), C.c_double(arm), _p(grav), _p(tau))
    return tau


def tangent_scheme_jacobians(x, u, h, gravity):
    """(A, B) of the constraint-free step by the scheme whose flops opcount.cpp counts (same code on plain doubles)."""
    L = lib()
    x, u, g = c64(x), c64(u), c64(gravity)
    A, B = np.zeros((NX, NX)), np.zeros((NX, NU))
    L.orc_tangent_scheme_jacobians(_p(x), _p(u), C.c_double(h), _p(g), _p(A), _p(B))
    return A, B


def max_threads():
    return lib().orc_max_threads()
