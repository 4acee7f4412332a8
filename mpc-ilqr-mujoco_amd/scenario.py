"""Problem data and synthetic batches for the H1 iLQR hot path (host-side, numpy only).

Mirrors the reference's configuration rules:
  * Config::buildCostMatrices      /root/reference/src/common/config.cpp:66-122 (diagonal Q, R, Qf)
  * shipped weights                /root/reference/config.yaml:24-58
  * standing pose                  /root/reference/src/common/robot_utils.cpp:567-583, data/q_standing.csv
  * reference construction         /root/reference/src/common/robot_utils.cpp:355-413 (com / foot refs by FK)
  * synthetic batch distributions  SURVEY.md section 8(d)
"""
import numpy as np

NQ, NV, NX, NU = 26, 25, 51, 19

# /root/reference/config.yaml:24-58 (as shipped)
SHIPPED_CONFIG = dict(
    Q_position_x=200.0, Q_position_y=50.0, Q_position_z=200.0, Q_quat_w=50.0, Q_quat_xyz=(50.0, 50.0, 50.0),
    Q_joint_pos=50.0, Q_vel_x=150.0, Q_vel_y=50.0, Q_vel_z=150.0, Q_ang_vel=75.0, Q_joint_vel=75.0,
    R_control=0.001, Qf_multiplier=2.0, Qf_position_x=5.0, Qf_position_y=2.0, Qf_position_z=5.0, Qf_vel_z=4.0,
    W_com_pos=100.0, W_com_vel=0.0, W_foot=400.0, W_foot_vel=400.0, W_upright=20.0, w_balance=30.0,
    joint_limit_weight=1500.0, torque_limit_weight=1500.0,
    gravity=(0.0, 0.0, -1.0), horizon=25, dt=0.02,
)

# /root/reference/robots/h1_description/mjcf/h1.xml:191-209
CTRLRANGE = np.array([200, 200, 200, 300, 40, 200, 200, 200, 300, 40, 200, 40, 40, 18, 18, 40, 40, 18, 18], dtype=np.float64)


def build_cost_matrices(cfg=SHIPPED_CONFIG):
    """Diagonals of Q, R, Qf exactly as Config::buildCostMatrices (config.cpp:66-122)."""
    Q = np.ones(NX)
    Q[0], Q[1], Q[2] = cfg["Q_position_x"], cfg["Q_position_y"], cfg["Q_position_z"]
    Q[3] = cfg["Q_quat_w"]
    Q[4:7] = cfg["Q_quat_xyz"]
    Q[7:NQ] = cfg["Q_joint_pos"]
    Q[NQ + 0], Q[NQ + 1], Q[NQ + 2] = cfg["Q_vel_x"], cfg["Q_vel_y"], cfg["Q_vel_z"]
    Q[NQ + 3:NQ + 6] = cfg["Q_ang_vel"]
    Q[NQ + 6:] = cfg["Q_joint_vel"]
    R = np.ones(NU) * cfg["R_control"]
    Qf = Q * cfg["Qf_multiplier"]
    Qf[0] *= cfg["Qf_position_x"]
    Qf[1] *= cfg["Qf_position_y"]
    Qf[2] *= cfg["Qf_position_z"]
    Qf[NQ + 2] *= cfg["Qf_vel_z"]
    return Q, R, Qf


def standing_state():
    x = np.zeros(NX)
    x[2] = 1.0432
    x[3] = 1.0
    return x


def make_problem(kin, N=25, cfg=SHIPPED_CONFIG, x_ref=None, stance=None, gravity=None):
    """Shared-reference problem dict. `kin(x) -> (com[3], ee[2,3])` supplies the FK used by
    loadReferences (robot_utils.cpp:369-403). x_ref: [N+1,51] (default: standing row repeated)."""
    Q, R, Qf = build_cost_matrices(cfg)
    if x_ref is None:
        x_ref = np.tile(standing_state(), (N + 1, 1))
    x_ref = np.asarray(x_ref, dtype=np.float64).reshape(1, N + 1, NX)
    com_ref = np.zeros((1, N + 1, 3))
    ee_ref = np.zeros((1, N + 1, 2, 3))
    for t in range(N + 1):
        com_ref[0, t], ee_ref[0, t] = kin(x_ref[0, t])
    if stance is None:
        stance = np.ones((N + 1, 2), dtype=np.int32)
    return dict(
        N=N, dt=cfg["dt"], Q=Q, R=R, Qf=Qf,
        task_weights=(cfg["W_com_pos"], cfg["W_com_vel"], cfg["W_foot"], cfg["W_foot_vel"], cfg["W_upright"], cfg["w_balance"]),
        w_joint=cfg["joint_limit_weight"], w_ctrl=cfg["torque_limit_weight"],
        gravity=tuple(cfg["gravity"] if gravity is None else gravity),
        x_ref=x_ref, u_ref=np.zeros((1, N, NU)), com_ref=com_ref,
        stance=np.asarray(stance, dtype=np.int32).reshape(1, N + 1, 2), ee_ref=ee_ref,
        com_vel_ref=np.zeros((1, N + 1, 3)),
    )


def _axis_angle_quat(w):
    ang = np.linalg.norm(w, axis=-1, keepdims=True)
    half = 0.5 * ang
    s = np.where(ang > 1e-12, np.sin(half) / np.maximum(ang, 1e-300), 0.5)
    return np.concatenate([np.cos(half), s * w], axis=-1)


def synthetic_batch(B, N, seed, u_gravcomp):
    """Seeded standing-balance batch: x0 = x_stand + delta, u_init = u_gravcomp + U(-1,1)
    clipped to 80 % of ctrlrange. Returns x0 [B,51], u_init [B,N,19]."""
    rng = np.random.default_rng(seed)
    x0 = np.tile(standing_state(), (B, 1))
    x0[:, 0:3] += rng.uniform(-0.02, 0.02, size=(B, 3))
    x0[:, 3:7] = _axis_angle_quat(rng.uniform(-0.05, 0.05, size=(B, 3)))
    x0[:, 7:NQ] += rng.uniform(-0.05, 0.05, size=(B, NQ - 7))
    x0[:, NQ:] += rng.uniform(-0.1, 0.1, size=(B, NV))
    u = np.asarray(u_gravcomp, dtype=np.float64)[None, None, :] + rng.uniform(-1.0, 1.0, size=(B, N, NU))
    u = np.clip(u, -0.8 * CTRLRANGE, 0.8 * CTRLRANGE)
    return x0, u


def walking_batch(B, N, seed, refdata_npz, sv, rf):
    """BASELINE.json configs[4]: B receding-horizon windows of the H1 walking reference (data/h1_walking_pin.csv rows, shipped as
    the fixture tests/golden/refdata_golden.npz), each with its own references and contact schedule: offline preparation as
    references.prepare_reference does it (Pinocchio -> MuJoCo quaternion order, velocities, stance flags from the foot hull),
    window start t0 drawn per rollout, initial state = the reference row perturbed, gravity-compensation controls + noise.
    `sv` = the solver module (host kinematics), `rf` = the references module.  Returns (problem with per-rollout sets, x0, u_init, t0)."""
    r = np.load(refdata_npz)
    q_mj = rf.pinocchio_to_mujoco(r["walking_pin_rows"])
    v = rf.differentiate_positions(q_mj, float(r["dt"]))
    flags = rf.contact_schedule(q_mj, sv.foot_clearance)
    rd = rf.ReferenceData(sv.reference_kinematics, sv.reference_com_velocity)
    rd.set_states(np.concatenate([q_mj, v], axis=1)); rd.contact = flags
    T = q_mj.shape[0]
    base = make_problem(sv.reference_kinematics, N=N, gravity=(0.0, 0.0, -1.0))
    rng = np.random.default_rng(seed)
    t0 = rng.integers(0, T - N - 1, size=B)
    probs = {}
    keys = ("x_ref", "u_ref", "com_ref", "stance", "ee_ref", "com_vel_ref")
    stacks = {k: [] for k in keys}
    for b in range(B):
        if int(t0[b]) not in probs:
            probs[int(t0[b])] = rd.problem_at(int(t0[b]), N, base, follow_schedule=True)
        for k in keys:
            stacks[k].append(probs[int(t0[b])][k][0])
    prob = dict(base); prob["N"] = N
    for k in keys:
        prob[k] = np.stack(stacks[k])
    x0 = rd.x_ref[t0].copy()
    x0[:, 7:26] += rng.uniform(-0.02, 0.02, (B, 19)); x0[:, 0:3] += rng.uniform(-0.01, 0.01, (B, 3)); x0[:, 26:] *= 0.5
    ug = sv.gravity_compensation(standing_state(), prob["gravity"])
    ui = np.tile(ug, (B, N, 1)) + rng.uniform(-0.5, 0.5, (B, N, 19))
    return prob, x0, ui, t0
