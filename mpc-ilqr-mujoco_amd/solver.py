"""Host-side Python mirror of the reference's solver interface over the C-ABI HIP library.

`BatchedILQR` mirrors `iLQR` (reference include/ilqr/ilqr.hpp:17-45) for a batch of B independent
rollouts; `BatchedMPC` mirrors `MPC::stepOnce` (reference include/ilqr/mpc.hpp:18-47,
src/ilqr/mpc.cpp:40-127).  All compute happens in libilqr_hip.so (hand-written HIP kernels for
gfx950); there is no CPU fallback -- constructing a solver without the library or without a GPU
raises.
"""
import ctypes as C
import os

import numpy as np

NX, NU, NQ, NV = 51, 19, 26, 25
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ILQR_HIP_LIB", os.path.join(_HERE, "lib", "libilqr_hip.so"))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

STATUS = {0: "ILQR_OK", 1: "ILQR_ERR_ARG", 2: "ILQR_ERR_HIP", 3: "ILQR_ERR_NO_DEVICE", 4: "ILQR_ERR_STATE", 5: "ILQR_ERR_UNSUPPORTED"}
JAC_ANALYTIC, JAC_FD_FORWARD = 0, 1

# every symbol include/ilqr_hip.h declares (checked by the CPU test-suite against the built library)
EXPORTS = [
    "ilqr_hip_create", "ilqr_hip_destroy", "ilqr_hip_last_error", "ilqr_hip_batch", "ilqr_hip_horizon", "ilqr_hip_num_slices", "ilqr_hip_reload_environment", "ilqr_hip_set_dedup_saturated_retry",
    "ilqr_hip_set_cost_weights", "ilqr_hip_set_task_weights", "ilqr_hip_set_constraint_weights", "ilqr_hip_set_gravity",
    "ilqr_hip_set_contact_schedule", "ilqr_hip_set_ee_references", "ilqr_hip_set_references",
    "ilqr_hip_set_regularization", "ilqr_hip_set_max_iterations", "ilqr_hip_set_tolerance", "ilqr_hip_set_options", "ilqr_hip_set_early_exit_gate",
    "ilqr_hip_initialize", "ilqr_hip_initialize_warm_resident", "ilqr_hip_initialize_device",
    "ilqr_hip_solve", "ilqr_hip_solve_async", "ilqr_hip_synchronize",
    "ilqr_hip_get_xbar", "ilqr_hip_get_ubar", "ilqr_hip_get_gains_K", "ilqr_hip_get_gains_kff", "ilqr_hip_get_cost",
    "ilqr_hip_get_iterations", "ilqr_hip_get_lambda", "ilqr_hip_get_trace", "ilqr_hip_first_knot_device", "ilqr_hip_pack_first_knot_device",
    "ilqr_hip_compute_control",
    "ilqr_hip_set_trajectory", "ilqr_hip_stage_rollout", "ilqr_hip_stage_linearize", "ilqr_hip_stage_cost_quadratics",
    "ilqr_hip_stage_backward_pass", "ilqr_hip_stage_line_search", "ilqr_hip_stage_total_cost",
    "ilqr_hip_get_linearization", "ilqr_hip_set_linearization", "ilqr_hip_get_quadratics", "ilqr_hip_set_quadratics",
    "ilqr_hip_get_value_function", "ilqr_hip_step", "ilqr_hip_step_stance", "ilqr_hip_set_contact_mode", "ilqr_hip_set_friction", "ilqr_hip_set_joint_limits", "ilqr_hip_set_joint_limit_stiffness", "ilqr_hip_enable_profiling", "ilqr_hip_get_stage_ms", "ilqr_hip_get_adopt_mismatches", "ilqr_hip_get_iterations_enqueued", "ilqr_hip_get_speculative_iterations", "ilqr_hip_get_split_iterations", "ilqr_hip_set_profiled_stages",
    "ilqr_hip_payload_width", "ilqr_hip_comm_available", "ilqr_hip_comm_get_unique_id", "ilqr_hip_comm_init", "ilqr_hip_comm_destroy", "ilqr_hip_comm_world", "ilqr_hip_comm_rank",
    "ilqr_hip_gather_first_knot",
    "ilqr_hip_reference_kinematics", "ilqr_hip_reference_com_velocity", "ilqr_hip_foot_clearance", "ilqr_hip_gravity_compensation", "ilqr_hip_stream",
]



class ILQRError(RuntimeError):
    pass


_libs = {}
# test library with every cross-check kernel family compiled in (csrc/Makefile, -DILQR_LEGACY_KERNELS); the product library
# holds the default family only and refuses a handle whose environment selects another
LEGACY_LIB_PATH = os.path.join(_HERE, "lib", "libilqr_hip_legacy.so")


def load_library(path=None):
    """Load libilqr_hip.so (or the library at `path`); fails loudly when the HIP extension has not been built."""
    path = LIB_PATH if path is None else path
    lib = _libs.get(path)
    if lib is None:
        if not os.path.exists(path):
            raise ILQRError("HIP extension missing: %s (run __graft_entry__.build())" % path)
        lib = C.CDLL(path)
        lib.ilqr_hip_last_error.restype = C.c_char_p
        lib.ilqr_hip_stream.restype = C.c_void_p
        _libs[path] = lib
    return lib


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def reference_kinematics(x):
    """(com[3], ee[2,3]) as RobotUtils::loadReferences computes them (robot_utils.cpp:369-403)."""
    L = load_library()
    x = _c64(x)
    com, ee = np.zeros(3), np.zeros((2, 3))
    rc = L.ilqr_hip_reference_kinematics(_p(x), _p(com), _p(ee))
    if rc:
        raise ILQRError(STATUS.get(rc, str(rc)))
    return com, ee


def reference_com_velocity(x):
    """CoM-velocity reference J_com(q) qvel as RobotUtils::loadReferences computes it (robot_utils.cpp:383-391)."""
    L = load_library()
    x = _c64(x)
    cv = np.zeros(3)
    rc = L.ilqr_hip_reference_com_velocity(_p(x), _p(cv))
    if rc:
        raise ILQRError(STATUS.get(rc, str(rc)))
    return cv


def foot_clearance(qpos):
    """Height of the lowest point of each foot's collision hull above the floor (get_contacts.py:96-147); [left, right]."""
    L = load_library()
    q = _c64(qpos)
    if q.shape != (26,):
        raise ValueError("qpos must have 26 entries (MuJoCo order)")
    clr = np.zeros(2)
    rc = L.ilqr_hip_foot_clearance(_p(q), _p(clr))
    if rc:
        raise ILQRError(STATUS.get(rc, str(rc)))
    return clr


def gravity_compensation(x, gravity):
    """qfrc_bias[6+i] at zero velocity (RobotUtils::computeGravComp, robot_utils.cpp:844-866)."""
    L = load_library()
    x, g = _c64(x), _c64(gravity)
    u = np.zeros(NU)
    rc = L.ilqr_hip_gravity_compensation(_p(x), _p(g), _p(u))
    if rc:
        raise ILQRError(STATUS.get(rc, str(rc)))
    return u


class BatchedILQR:
    """iLQR for B independent rollouts on one GPU (reference include/ilqr/ilqr.hpp:17-45)."""

    def __init__(self, batch, N=25, dt=0.02, device=0, lib_path=None):
        self.L = load_library(lib_path)
        self.B, self.N, self.dt = int(batch), int(N), float(dt)
        self.max_iter = 10
        h = C.c_void_p()
        rc = self.L.ilqr_hip_create(C.byref(h), int(device), self.B, self.N, C.c_double(self.dt))
        self.h = h
        if rc:
            msg = self.L.ilqr_hip_last_error(h).decode() if h else ""
            self.close()
            raise ILQRError("ilqr_hip_create failed: %s %s" % (STATUS.get(rc, rc), msg))

    def set_dedup_saturated_retry(self, on=True):
        """Skip lambda retries whose lambda is already saturated (bit-identical repeats of the pass that has just failed); off by default."""
        self._chk(self.L.ilqr_hip_set_dedup_saturated_retry(self.h, int(bool(on))))

    def reload_environment(self):
        """Re-read the diagnostic environment switches (read once, at creation) for this handle."""
        self._chk(self.L.ilqr_hip_reload_environment(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.ilqr_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise ILQRError("%s: %s" % (STATUS.get(rc, rc), self.L.ilqr_hip_last_error(self.h).decode()))

    # ---- problem data (RobotUtils setters)
    def set_problem(self, prob):
        L, h = self.L, self.h
        self._chk(L.ilqr_hip_set_cost_weights(h, _p(_c64(prob["Q"])), _p(_c64(prob["R"])), _p(_c64(prob["Qf"]))))
        self._chk(L.ilqr_hip_set_task_weights(h, *[C.c_double(float(v)) for v in prob["task_weights"]]))
        self._chk(L.ilqr_hip_set_constraint_weights(h, C.c_double(prob["w_joint"]), C.c_double(prob["w_ctrl"])))
        g = prob["gravity"]
        self.gravity = np.array(g, dtype=np.float64)
        self._chk(L.ilqr_hip_set_gravity(h, C.c_double(g[0]), C.c_double(g[1]), C.c_double(g[2])))
        st = np.ascontiguousarray(prob["stance"], dtype=np.int32)
        self._chk(L.ilqr_hip_set_contact_schedule(h, st.ctypes.data_as(_ip), int(st.shape[0])))
        ee, cv = _c64(prob["ee_ref"]), _c64(prob["com_vel_ref"])
        self._chk(L.ilqr_hip_set_ee_references(h, _p(ee), _p(cv), int(ee.shape[0])))
        self.set_references(prob["x_ref"], prob["u_ref"], prob["com_ref"])

    def set_references(self, x_ref, u_ref, com_ref):
        x_ref, u_ref, com_ref = _c64(x_ref), _c64(u_ref), _c64(com_ref)
        if x_ref.shape[1:] != (self.N + 1, NX) or u_ref.shape[1:] != (self.N, NU) or com_ref.shape[1:] != (self.N + 1, 3):
            raise ILQRError("reference size mismatch")  # iLQR::solve returns false (ilqr.cpp:526-532)
        self._chk(self.L.ilqr_hip_set_references(self.h, _p(x_ref), _p(u_ref), _p(com_ref), int(x_ref.shape[0])))

    # ---- options (ilqr.hpp:22-24)
    def set_regularization(self, lam):
        self._chk(self.L.ilqr_hip_set_regularization(self.h, C.c_double(lam)))

    def set_max_iterations(self, n):
        self._chk(self.L.ilqr_hip_set_max_iterations(self.h, int(n)))
        self.max_iter = int(n)

    def set_tolerance(self, tol):
        self._chk(self.L.ilqr_hip_set_tolerance(self.h, C.c_double(tol)))

    def set_options(self, jacobian_mode=JAC_ANALYTIC, fd_eps=1e-5, early_exit=True):
        self._chk(self.L.ilqr_hip_set_options(self.h, int(jacobian_mode), C.c_double(fd_eps), int(bool(early_exit))))

    def set_early_exit_gate(self, on=True):
        """Per-handle switch of the early-exit gate: off = solve_async enqueues all iterations at once and never blocks the host."""
        self._chk(self.L.ilqr_hip_set_early_exit_gate(self.h, int(bool(on))))

    # ---- initializeWithReference / solve
    def initialize(self, x0, u_init=None, prev_xbar=None, prev_ubar=None):
        ks = [_c64(x0), None if u_init is None else _c64(u_init), None if prev_xbar is None else _c64(prev_xbar), None if prev_ubar is None else _c64(prev_ubar)]
        self._chk(self.L.ilqr_hip_initialize(self.h, *[_p(k) for k in ks]))

    def initialize_warm_resident(self, x0):
        self._chk(self.L.ilqr_hip_initialize_warm_resident(self.h, _p(_c64(x0))))

    def initialize_device(self, x0_ptr, u_init_ptr):
        self._chk(self.L.ilqr_hip_initialize_device(self.h, C.c_void_p(x0_ptr), C.c_void_p(u_init_ptr)))

    def solve(self, x0=None):
        cost = np.zeros(self.B)
        x0 = None if x0 is None else _c64(x0)
        self._chk(self.L.ilqr_hip_solve(self.h, _p(x0), _p(cost)))
        return cost

    def solve_async(self):
        self._chk(self.L.ilqr_hip_solve_async(self.h))

    def synchronize(self):
        self._chk(self.L.ilqr_hip_synchronize(self.h))

    # ---- accessors (ilqr.hpp:34-37)
    def _get(self, fn, shape, dtype=np.float64):
        out = np.zeros(shape, dtype=dtype)
        ptr = out.ctypes.data_as(_dp if dtype == np.float64 else _ip)
        self._chk(getattr(self.L, fn)(self.h, ptr))
        return out

    def xbar(self):
        return self._get("ilqr_hip_get_xbar", (self.B, self.N + 1, NX))

    def ubar(self):
        return self._get("ilqr_hip_get_ubar", (self.B, self.N, NU))

    def gains_K(self):
        return self._get("ilqr_hip_get_gains_K", (self.B, self.N, NU, NX))

    def gains_kff(self):
        return self._get("ilqr_hip_get_gains_kff", (self.B, self.N, NU))

    def cost(self):
        return self._get("ilqr_hip_get_cost", (self.B,))

    def iterations(self):
        return self._get("ilqr_hip_get_iterations", (self.B,), np.int32)

    def lambdas(self):
        return self._get("ilqr_hip_get_lambda", (self.B,))

    def trace(self):
        cost = np.zeros((self.B, self.max_iter + 1))
        alpha = np.zeros((self.B, self.max_iter))
        lam = np.zeros((self.B, self.max_iter))
        self._chk(self.L.ilqr_hip_get_trace(self.h, _p(cost), _p(alpha), _p(lam)))
        return cost, alpha, lam

    def first_knot_device(self):
        """Device pointers (u0[B][19], K0[B][19][51], cost[B]) -- the payload of the per-step gather."""
        u0, K0, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk(self.L.ilqr_hip_first_knot_device(self.h, C.byref(u0), C.byref(K0), C.byref(c)))
        return u0.value, K0.value, c.value

    def pack_first_knot_device(self, u0_ptr, K0_ptr=None, cost_ptr=None):
        """Write u0[B][19] (+ K0[B][19][51], cost[B]) into caller-owned device buffers (gather payload)."""
        self._chk(self.L.ilqr_hip_pack_first_knot_device(self.h, C.c_void_p(u0_ptr), C.c_void_p(K0_ptr), C.c_void_p(cost_ptr)))

    # ---- multi-GPU: RCCL behind the C ABI (include/ilqr_hip.h "multi-GPU")
    @staticmethod
    def comm_available():
        """True if librccl can be opened with every entry point the gather needs (check on EVERY rank before comm_init(world > 1))."""
        return bool(load_library().ilqr_hip_comm_available())

    @staticmethod
    def comm_unique_id():
        """128-byte RCCL id (rank 0 creates it and hands it to every rank)."""
        L = load_library()
        buf = C.create_string_buffer(128)
        rc = L.ilqr_hip_comm_get_unique_id(buf)
        if rc:
            raise ILQRError("ilqr_hip_comm_get_unique_id: %s" % STATUS.get(rc, rc))
        return buf.raw

    def comm_init(self, world, rank, unique_id=None):
        self._chk(self.L.ilqr_hip_comm_init(self.h, int(world), int(rank), unique_id))

    def comm_destroy(self):
        self._chk(self.L.ilqr_hip_comm_destroy(self.h))

    def gather_first_knot(self, recv_ptr, root=0, with_gains=False):
        """Enqueue the per-step gather of [u0 | cost | (K0)] rows to `root` (recv_ptr: device buffer there, None elsewhere)."""
        self._chk(self.L.ilqr_hip_gather_first_knot(self.h, int(root), int(bool(with_gains)), C.c_void_p(recv_ptr)))

    def compute_control(self, x_measured):
        u = np.zeros((self.B, NU))
        self._chk(self.L.ilqr_hip_compute_control(self.h, _p(_c64(x_measured)), _p(u)))
        return u

    # ---- stage entry points
    def set_trajectory(self, xbar, ubar):
        self._chk(self.L.ilqr_hip_set_trajectory(self.h, _p(_c64(xbar)), _p(_c64(ubar))))

    def stage_rollout(self):
        self._chk(self.L.ilqr_hip_stage_rollout(self.h))

    def stage_linearize(self):
        self._chk(self.L.ilqr_hip_stage_linearize(self.h))

    def stage_cost_quadratics(self):
        self._chk(self.L.ilqr_hip_stage_cost_quadratics(self.h))

    def stage_backward_pass(self):
        self._chk(self.L.ilqr_hip_stage_backward_pass(self.h))

    def stage_line_search(self):
        imp = np.zeros(self.B, dtype=np.int32)
        cost, alpha = np.zeros(self.B), np.zeros(self.B)
        self._chk(self.L.ilqr_hip_stage_line_search(self.h, imp.ctypes.data_as(_ip), _p(cost), _p(alpha)))
        return imp.astype(bool), cost, alpha

    def stage_total_cost(self):
        cost = np.zeros(self.B)
        self._chk(self.L.ilqr_hip_stage_total_cost(self.h, _p(cost)))
        return cost

    def linearization(self):
        A = np.zeros((self.B, self.N, NX, NX))
        Bm = np.zeros((self.B, self.N, NX, NU))
        self._chk(self.L.ilqr_hip_get_linearization(self.h, _p(A), _p(Bm)))
        return A, Bm

    def set_linearization(self, A, Bm):
        self._chk(self.L.ilqr_hip_set_linearization(self.h, _p(_c64(A)), _p(_c64(Bm))))

    def quadratics(self):
        lx, lu = np.zeros((self.B, self.N + 1, NX)), np.zeros((self.B, self.N, NU))
        lxx, luu = np.zeros((self.B, self.N + 1, NX, NX)), np.zeros((self.B, self.N, NU))
        self._chk(self.L.ilqr_hip_get_quadratics(self.h, _p(lx), _p(lu), _p(lxx), _p(luu)))
        return lx, lu, lxx, luu

    def set_quadratics(self, lx, lu, lxx, luu):
        self._chk(self.L.ilqr_hip_set_quadratics(self.h, _p(_c64(lx)), _p(_c64(lu)), _p(_c64(lxx)), _p(_c64(luu))))

    def value_function(self):
        Vx, Vxx = np.zeros((self.B, NX)), np.zeros((self.B, NX, NX))
        self._chk(self.L.ilqr_hip_get_value_function(self.h, _p(Vx), _p(Vxx)))
        return Vx, Vxx

    def step(self, x, u):
        x, u = _c64(x), _c64(u)
        xn = np.zeros_like(x)
        self._chk(self.L.ilqr_hip_step(self.h, int(x.shape[0]), _p(x), _p(u), _p(xn)))
        return xn

    def step_stance(self, x, u, stance_left, stance_right):
        """One step with explicit stance flags (they matter in contact mode only)."""
        x, u = _c64(x), _c64(u)
        xn = np.zeros_like(x)
        self._chk(self.L.ilqr_hip_step_stance(self.h, int(x.shape[0]), _p(x), _p(u), int(stance_left), int(stance_right), _p(xn)))
        return xn

    def set_contact_mode(self, mode, softness=0.0):
        """0: constraint-free step; 1: rigid stance constraints on the feet the contact schedule marks;
        2: unilateral; 3: unilateral + Coulomb limit (set_friction), sliding feet without tangential force; 4: with kinetic friction."""
        self._chk(self.L.ilqr_hip_set_contact_mode(self.h, int(mode), C.c_double(softness)))
        self.contact_mode = int(mode)

    def set_friction(self, mu):
        """Sliding friction coefficient of contact modes 3 / 4 (unilateral stance + Coulomb limit)."""
        self._chk(self.L.ilqr_hip_set_friction(self.h, C.c_double(mu)))

    def set_joint_limits(self, on=True):
        """Joint-limit rows of the plant (include/ilqr_hip.h): a hinge past its range that the step would still move outward is stopped."""
        self._chk(self.L.ilqr_hip_set_joint_limits(self.h, int(bool(on))))

    def set_joint_limit_stiffness(self, k):
        """Restoring stiffness of the joint-limit rows (include/ilqr_hip.h): qacc_i = -v_i / h - k r_i on a constrained hinge; 0 = the pure stop,
        1 / (2 h)^2 = MuJoCo's default solref time constant."""
        self._chk(self.L.ilqr_hip_set_joint_limit_stiffness(self.h, C.c_double(float(k))))

    def enable_profiling(self, on=True):
        self._chk(self.L.ilqr_hip_enable_profiling(self.h, int(bool(on))))

    STAGE_KEYS = ["iLQR_computeCost+forwardRollout", "iLQR_linearization", "iLQR_costQuadratics", "iLQR_backwardPass", "iLQR_lineSearch", "iLQR_control",
                  "iLQR_backwardPass_retry", "iLQR_lineSearch_retry"]

    def set_profiled_stages(self, keys=None):
        """Time only the given stages (keys of stage_ms()) while profiling is on; None = all."""
        mask = 0xFF if keys is None else sum(1 << self.STAGE_KEYS.index(k) for k in keys)
        self._chk(self.L.ilqr_hip_set_profiled_stages(self.h, C.c_uint(mask)))

    def adopt_mismatches(self):
        """Elements in which the concurrent nominal re-rollouts of the last solve differed from the trajectory they replaced."""
        n = C.c_ulonglong(0)
        self._chk(self.L.ilqr_hip_get_adopt_mismatches(self.h, C.byref(n)))
        return int(n.value)

    def iterations_enqueued(self):
        """Iterations the last solve launched (fewer than max_iterations once the whole batch has taken the convergence exit)."""
        return int(self.L.ilqr_hip_get_iterations_enqueued(self.h))

    def speculative_iterations(self):
        """Iterations of the last solve whose lambda retry ran beside the first pass (small passes; ILQR_SPEC=0 disables)."""
        return int(self.L.ilqr_hip_get_speculative_iterations(self.h))

    def split_iterations(self):
        """Iterations of the last solve whose concurrent region ran in two groups (early continuation; ILQR_SPLIT=0 disables)."""
        return int(self.L.ilqr_hip_get_split_iterations(self.h))

    def num_slices(self):
        """Batch slices a solve is enqueued as (ILQR_SLICES)."""
        return int(self.L.ilqr_hip_num_slices(self.h))

    def stage_ms(self):
        ms, n = np.zeros(8), np.zeros(8)
        self._chk(self.L.ilqr_hip_get_stage_ms(self.h, _p(ms), _p(n)))
        keys = ["iLQR_computeCost+forwardRollout", "iLQR_linearization", "iLQR_costQuadratics", "iLQR_backwardPass", "iLQR_lineSearch", "iLQR_control",
                "iLQR_backwardPass_retry", "iLQR_lineSearch_retry"]
        return dict(zip(keys, ms)), dict(zip(keys, n))

    @property
    def stream(self):
        return self.L.ilqr_hip_stream(self.h)


class BatchedMPC:
    """MPC::stepOnce for a batch (reference src/ilqr/mpc.cpp:40-127): window -> warm start -> solve ->
    u = ubar0 + K0 (x - xbar0).  `window(t_idx)` returns (x_ref, u_ref, com_ref) for the current step
    (RobotUtils::getReferenceWindow, robot_utils.cpp:422-443)."""

    def __init__(self, solver, window):
        self.ilqr, self.window = solver, window
        self.t_idx, self.has_prev = 0, False
        self.last_solve_cost = None

    def reset(self):
        self.t_idx, self.has_prev = 0, False

    def step_once(self, x_measured):
        x_ref, u_ref, com_ref = self.window(self.t_idx)
        self.ilqr.set_references(x_ref, u_ref, com_ref)
        if self.has_prev:
            self.ilqr.initialize_warm_resident(x_measured)
        else:
            self.ilqr.initialize(x_measured)
        self.last_solve_cost = self.ilqr.solve(x_measured)
        u = self.ilqr.compute_control(x_measured)
        self.has_prev = True
        self.t_idx += 1
        return u
