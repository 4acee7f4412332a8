"""Reference / contact-schedule loader of the MPC (host side, numpy only) -- SURVEY.md 8(f) row f1.

Mirrors, on the reference's own file formats:
  * RobotUtils::loadReferences       /root/reference/src/common/robot_utils.cpp:281-420
      q_ref CSV (26 columns, MuJoCo order: p, quat wxyz, 19 hinges) + v_ref CSV (25 columns), no header, one row per
      time step; rows with the wrong number of columns are skipped; u_ref = 0; per row the whole-body CoM
      (subtree_com of the pelvis), the CoM velocity J_com qvel and the world positions of the two ankle bodies.
  * RobotUtils::getReferenceWindow   robot_utils.cpp:422-443   rows t0 .. t0+N, clamped to the last row
  * RobotUtils::loadContactSchedule  robot_utils.cpp:445-492   CSV with one header line, one row of 0/1 per time step
  * RobotUtils::isStance             robot_utils.cpp:494-504   out-of-range -> stance
  * the solver's use of them         ilqr.cpp:703,729-734,767-791 with the HORIZON-LOCAL index t = 0..N for the
      stance flags, foot references and CoM-velocity references: whatever the MPC step, the
      solver reads rows 0..N of the full arrays.  `problem_at` reproduces that; `follow_schedule=True` reads rows
      t0..t0+N instead (what the reference presumably intended).
The kinematics come from the C ABI (`ilqr_hip_reference_kinematics`, `ilqr_hip_reference_com_velocity`: host code,
no GPU needed) or any callable with the same signature (the tests pass the oracle's).
"""
import numpy as np

NQ, NV, NX, NU = 26, 25, 51, 19


def _read_rows(path, ncol, skip_header=False):
    rows = []
    with open(path) as f:
        for k, line in enumerate(f):
            if skip_header and k == 0:
                continue
            vals = []
            for tok in line.strip().split(","):
                try:
                    vals.append(float(tok))
                except ValueError:       # the reference logs and drops the token (robot_utils.cpp:318-325)
                    continue
            if ncol is not None and len(vals) != ncol:
                continue                 # dimension mismatch: row skipped (robot_utils.cpp:336-341)
            if vals:
                rows.append(vals)
    return rows


# ---- offline data prep: /root/reference/get_contacts.py:18-41 (conventions), :96-147 (contact schedule)
def pinocchio_to_mujoco(q_pin):
    """[x y z qx qy qz qw joints] -> [x y z qw qx qy qz joints] (get_contacts.py convert_pinocchio_to_mujoco)."""
    q = np.array(q_pin, dtype=np.float64, copy=True)
    q[..., 3], q[..., 4:7] = np.asarray(q_pin)[..., 6], np.asarray(q_pin)[..., 3:6]
    return q


def differentiate_positions(q_mj, dt):
    """Velocity file of a position reference, MuJoCo's mj_differentiatePos rule row to row (how data/v_ref2.csv relates
    to data/q_ref2_mj.csv): world-frame linear velocity, BODY-frame angular velocity = log(q_t^-1 (x)
    q_{t+1}) / dt, plain hinge differences; the last row repeats the previous one."""
    q = np.asarray(q_mj, dtype=np.float64)
    T = q.shape[0]
    v = np.zeros((T, NV))
    for t in range(T - 1):
        qa, qb = q[t, 3:7], q[t + 1, 3:7]
        ca = qa * np.array([1.0, -1.0, -1.0, -1.0])
        dq = np.concatenate([[ca[0] * qb[0] - ca[1:] @ qb[1:]], ca[0] * qb[1:] + qb[0] * ca[1:] + np.cross(ca[1:], qb[1:])])
        sn = np.linalg.norm(dq[1:])
        speed = 2.0 * np.arctan2(sn, dq[0])
        if speed > np.pi:
            speed -= 2.0 * np.pi
        v[t, 0:3] = (q[t + 1, 0:3] - q[t, 0:3]) / dt
        v[t, 3:6] = (dq[1:] / sn if sn > 1e-15 else np.zeros(3)) * speed / dt
        v[t, 6:] = (q[t + 1, 7:] - q[t, 7:]) / dt
    if T > 1:
        v[T - 1] = v[T - 2]
    return v


def contact_schedule(q_mj, clearance):
    """Stance flags [T][2] (left, right) of a position reference, the way get_contacts.py:96-147 makes them: per row set
    qpos, run the kinematics and mark a foot as in stance when its collision hull touches the floor plane.  The tool
    tests `contact.dist < 0.001` on MuJoCo's contact list, but MuJoCo only lists a floor contact once the hull
    penetrates (margin 0), so the effective rule is "lowest hull point below z = 0"; it reproduces all 800 flags of
    the reference's data/contact_walking.csv and the 400 of contact_standing.csv (tests/test_references.py).
    `clearance(qpos[26]) -> [left, right]` is `solver.foot_clearance` (C ABI `ilqr_hip_foot_clearance`)."""
    q = np.atleast_2d(np.asarray(q_mj, dtype=np.float64))
    if q.shape[1] != NQ:
        raise ValueError("position rows must have %d columns" % NQ)
    flags = np.zeros((q.shape[0], 2), dtype=np.int32)
    for t in range(q.shape[0]):
        flags[t] = np.asarray(clearance(q[t])) < 0.0
    return flags


def write_contact_schedule(path, flags):
    """The tool's output format (get_contacts.py:141-143): header `left_foot,right_foot`, one 0/1 row per time step --
    what RobotUtils::loadContactSchedule / ReferenceData.load_contact_schedule read back."""
    with open(path, "w") as f:
        f.write("left_foot,right_foot\n")
        for a, b in np.asarray(flags, dtype=int):
            f.write("%d,%d\n" % (a, b))


def write_rows(path, rows):
    """Header-less CSV of float rows (the format of q_ref*.csv / v_ref*.csv), full double precision."""
    with open(path, "w") as f:
        for r in np.asarray(rows, dtype=np.float64):
            f.write(",".join(repr(float(a)) for a in r) + "\n")


def prepare_reference(q_pin_path, dt, clearance, out_prefix=None):
    """The whole offline preparation of a Pinocchio-convention position file (e.g. the 7840-row data/h1_walking_pin.csv,
    which ships without velocity and contact files): quaternion reorder, velocities by `differentiate_positions`,
    stance flags by `contact_schedule`.  Returns (q_mj, v, flags); with `out_prefix` also writes
    <prefix>_q_mj.csv, <prefix>_v.csv, <prefix>_contact.csv in the reference's formats."""
    q_pin = np.asarray(_read_rows(q_pin_path, NQ), dtype=np.float64)
    if q_pin.size == 0:
        raise ValueError("no %d-column rows in %s" % (NQ, q_pin_path))
    q_mj = pinocchio_to_mujoco(q_pin)
    v = differentiate_positions(q_mj, dt)
    flags = contact_schedule(q_mj, clearance)
    if out_prefix is not None:
        write_rows(out_prefix + "_q_mj.csv", q_mj)
        write_rows(out_prefix + "_v.csv", v)
        write_contact_schedule(out_prefix + "_contact.csv", flags)
    return q_mj, v, flags


class ReferenceData:
    """Full-length references as RobotUtils holds them (x_ref_full_, com_ref_full_, ..., contact_schedule_)."""

    def __init__(self, kin, com_vel=None):
        self.kin, self.com_vel = kin, com_vel
        self.x_ref = np.zeros((0, NX)); self.u_ref = np.zeros((0, NU))
        self.com_ref = np.zeros((0, 3)); self.com_vel_ref = np.zeros((0, 3)); self.ee_ref = np.zeros((0, 2, 3))
        self.contact = np.zeros((0, 2), dtype=np.int32)

    # -- RobotUtils::loadReferences
    def load(self, q_ref_path, v_ref_path):
        q, v = _read_rows(q_ref_path, None), _read_rows(v_ref_path, None)
        rows = [(a, b) for a, b in zip(q, v) if len(a) == NQ and len(b) == NV]   # lines are consumed pairwise
        if not rows:
            return False
        return self.set_states(np.array([a + b for a, b in rows], dtype=np.float64))

    def set_states(self, x_full):
        x_full = np.asarray(x_full, dtype=np.float64).reshape(-1, NX)
        T = x_full.shape[0]
        self.x_ref = x_full.copy()
        self.u_ref = np.zeros((T, NU))
        self.com_ref = np.zeros((T, 3)); self.ee_ref = np.zeros((T, 2, 3)); self.com_vel_ref = np.zeros((T, 3))
        for t in range(T):
            self.com_ref[t], self.ee_ref[t] = self.kin(x_full[t])
            if self.com_vel is not None:
                self.com_vel_ref[t] = self.com_vel(x_full[t])
        return T > 0

    # -- RobotUtils::loadContactSchedule
    def load_contact_schedule(self, path):
        rows = _read_rows(path, None, skip_header=True)
        self.contact = np.array([[int(v) for v in r] for r in rows], dtype=np.int32).reshape(len(rows), -1) if rows else np.zeros((0, 2), dtype=np.int32)
        return len(rows) > 0

    def is_stance(self, ee, t):
        if t < 0 or t >= self.contact.shape[0] or ee < 0 or ee >= self.contact.shape[1]:
            return True
        return int(self.contact[t, ee]) == 1

    # -- RobotUtils::getReferenceWindow
    def window(self, t0, N):
        T = self.x_ref.shape[0]
        idx = np.minimum(t0 + np.arange(N + 1), T - 1)
        return self.x_ref[idx], self.u_ref[idx[:N]], self.com_ref[idx]

    def _rows(self, arr, t0, N):
        """rows t0..t0+N of a full-length array; the reference throws past the end (robot_utils.cpp:525-549)."""
        if t0 + N >= arr.shape[0]:
            raise IndexError("reference index %d beyond the %d loaded rows" % (t0 + N, arr.shape[0]))
        return arr[t0:t0 + N + 1]

    def problem_at(self, t0, N, base_problem, follow_schedule=False):
        """Problem dict for the solve of MPC step t0: window of x_ref / u_ref / com_ref (MPC::extractReferenceWindow,
        mpc.cpp:163-166) + stance flags, foot and CoM-velocity references by the horizon-local index."""
        prob = dict(base_problem)
        xw, uw, cw = self.window(t0, N)
        s0 = t0 if follow_schedule else 0
        prob.update(N=N, x_ref=xw[None], u_ref=uw[None], com_ref=cw[None],
                    stance=np.array([[1 if self.is_stance(e, s0 + t) else 0 for e in range(2)] for t in range(N + 1)], dtype=np.int32)[None],
                    ee_ref=self._rows(self.ee_ref, s0, N)[None].copy(), com_vel_ref=self._rows(self.com_vel_ref, s0, N)[None].copy())
        return prob
