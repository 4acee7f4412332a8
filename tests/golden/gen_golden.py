#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (run in the BUILD container only).

The reference (/root/reference) has no tests/golden outputs for this path and cannot be built or
imported here (MuJoCo/Pinocchio/CasADi/Eigen absent) => these goldens are INDEPENDENT restatements
written differently from oracle/ (which they pin), plus excerpts of the reference's own data files:

  dynamics_golden.npz : one MuJoCo-semantics step computed by a world-frame Kane (virtual power)
                        formulation in NumPy with constants parsed straight from h1.xml
                        (oracle uses a body-frame articulated-body algorithm).
  cost_golden.npz     : per-term gradient/Hessian of the six task terms via torch autograd (fp64) of
                        an independently written torch FK with constants parsed from h1.urdf, the
                        line-search cost (computeTotalCost) and the soft-limit penalties.
  riccati_golden.npz  : NumPy restatement of iLQR::backwardPass (ilqr.cpp:250-309) on synthetic inputs
                        incl. the LLT-failure branch.
  solve_golden.npz    : the WHOLE solve loop (ilqr.cpp:521-660) restated with the pieces above on a 6-knot horizon: Kane-step
                        rollout, central-difference Jacobians of it, torch-autograd cost quadratics, NumPy Riccati, line search.
  refdata_golden.npz  : rows of data/q_ref2_mj.csv, data/v_ref2.csv (reference data files) used as a
                        known-answer test of the state conventions.
"""
import math
import os
import xml.etree.ElementTree as ET

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_default_dtype(torch.float64)


# ----------------------------------------------------------------------------- model parsing
def fl(s):
    return np.array([float(t) for t in s.split()])


def q2R(q):
    w, x, y, z = np.asarray(q, dtype=float) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def load_mjcf():
    root = ET.parse(os.path.join(REF, "robots/h1_description/mjcf/h1.xml")).getroot()
    bodies = []

    def rec(e, parent):
        for b in e.findall("body"):
            i = len(bodies)
            ine = b.find("inertial")
            j = b.find("joint")
            Rq = q2R(fl(ine.get("quat", "1 0 0 0")))
            bodies.append(dict(name=b.get("name"), parent=parent, pos=fl(b.get("pos", "0 0 0")), R=q2R(fl(b.get("quat", "1 0 0 0"))),
                               m=float(ine.get("mass")), c=fl(ine.get("pos")), I=Rq @ np.diag(fl(ine.get("diaginertia"))) @ Rq.T,
                               axis=None if j is None else fl(j.get("axis")), rng=None if j is None else fl(j.get("range"))))
            rec(b, i)

    rec(root.find("worldbody"), -1)
    ctrl = [fl(m.get("ctrlrange")) for m in root.find("actuator").findall("motor")]
    return bodies, np.array(ctrl)


def load_urdf():
    root = ET.parse(os.path.join(REF, "robots/h1_description/urdf/h1.urdf")).getroot()
    links = {l.get("name"): l for l in root.findall("link")}
    joints = [j for j in root.findall("joint") if j.get("type") == "revolute"]
    order = ["pelvis"] + [j.find("child").get("link") for j in joints]
    out = []
    for name in order:
        ine = links[name].find("inertial")
        d = dict(name=name, m=float(ine.find("mass").get("value")), c=fl(ine.find("origin").get("xyz")))
        if name == "pelvis":
            d.update(parent=-1, pos=np.zeros(3), rpy=np.zeros(3), axis=None)
        else:
            j = [jj for jj in joints if jj.find("child").get("link") == name][0]
            d.update(parent=order.index(j.find("parent").get("link")), pos=fl(j.find("origin").get("xyz")),
                     rpy=fl(j.find("origin").get("rpy", "0 0 0")), axis=fl(j.find("axis").get("xyz")))
        out.append(d)
    return out


# ----------------------------------------------------------------------------- Kane dynamics (numpy)
def skew(a):
    return np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])


def axis_rot(axis, th):
    K = skew(axis)
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * (K @ K)


def kane_residual(bodies, q, v, qacc, grav, armature):
    """Generalized inertia+bias forces F(qacc) = M qacc + bias in MuJoCo coordinates (world-frame Kane)."""
    nb = len(bodies)
    quat = q[3:7] / np.linalg.norm(q[3:7])
    R = [None] * nb; p = [None] * nb; om = [None] * nb; al = [None] * nb; acc = [None] * nb; z = [None] * nb
    R[0] = q2R(quat); p[0] = q[0:3].copy()
    om[0] = R[0] @ v[3:6]; al[0] = R[0] @ qacc[3:6]; acc[0] = qacc[0:3].copy()
    for i in range(1, nb):
        b = bodies[i]; pa = b["parent"]
        Rj = b["R"] @ axis_rot(b["axis"], q[7 + i - 1])
        R[i] = R[pa] @ Rj
        d = R[pa] @ b["pos"]
        p[i] = p[pa] + d
        z[i] = R[i] @ b["axis"]
        qd, qdd = v[6 + i - 1], qacc[6 + i - 1]
        om[i] = om[pa] + z[i] * qd
        al[i] = al[pa] + np.cross(om[pa], z[i]) * qd + z[i] * qdd
        acc[i] = acc[pa] + np.cross(al[pa], d) + np.cross(om[pa], np.cross(om[pa], d))
    F = np.zeros(25)
    for i in range(nb):
        b = bodies[i]
        e = R[i] @ b["c"]
        ci = p[i] + e
        ac = acc[i] + np.cross(al[i], e) + np.cross(om[i], np.cross(om[i], e))
        Iw = R[i] @ b["I"] @ R[i].T
        fl_ = b["m"] * (ac - grav)
        tq = Iw @ al[i] + np.cross(om[i], Iw @ om[i])
        # partial velocities of body i w.r.t. the generalized speeds
        F[0:3] += fl_
        F[3:6] += R[0].T @ (np.cross(ci - p[0], fl_) + tq)
        a = i
        while a > 0:
            F[6 + a - 1] += z[a] @ (np.cross(ci - p[a], fl_) + tq)
            a = bodies[a]["parent"]
    F[6:] += armature * qacc[6:]
    return F


def kane_step(bodies, ctrlrange, x, u, h, grav, damping=1.0, armature=0.1):
    q, v = x[:26].copy(), x[26:].copy()
    bias = kane_residual(bodies, q, v, np.zeros(25), grav, armature)
    M = np.stack([kane_residual(bodies, q, v, np.eye(25)[k], grav, armature) - bias for k in range(25)], axis=1)
    D = np.concatenate([np.zeros(6), damping * np.ones(19)])
    tau = np.concatenate([np.zeros(6), np.clip(u, ctrlrange[:, 0], ctrlrange[:, 1])])
    qacc = np.linalg.solve(M + h * np.diag(D), tau - D * v - bias)
    vn = v + h * qacc
    qn = q.copy()
    qn[0:3] += h * vn[0:3]
    qn[7:] += h * vn[6:]
    quat = q[3:7] / np.linalg.norm(q[3:7])
    w = vn[3:6]; ang = np.linalg.norm(w) * h
    e = np.array([1.0, 0, 0, 0]) if ang < 1e-15 else np.concatenate([[math.cos(ang / 2)], math.sin(ang / 2) * w / np.linalg.norm(w)])
    a, b = quat, e
    r = np.array([a[0] * b[0] - a[1:] @ b[1:], *(a[0] * b[1:] + b[0] * a[1:] + np.cross(a[1:], b[1:]))])
    qn[3:7] = r / np.linalg.norm(r)
    return np.concatenate([qn, vn]), qacc, M, bias


# ----------------------------------------------------------------------------- complex-safe Kane dynamics (round 3)
# The same world-frame Kane restatement, written so that it accepts complex arguments: exact Jacobians by complex-step
# differentiation (imag f(x + i eps e_k) / eps, eps = 1e-30: no subtraction, no truncation error), and the stance-constrained
# step of the contact row as a plain KKT system -- independent of the oracle's / the kernels'
# articulated-body formulation (unit-wrench propagation, C = J Mhat^-1 J^T, Cholesky).
def cnorm(a):
    return np.sqrt((a * a).sum())


def q2R_c(q):
    w, x, y, z = q / cnorm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def axis_rot_c(axis, th):
    K = skew(axis)
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


FEET = (5, 10)      # left_ankle_link, right_ankle_link (robot_utils.cpp:44-47)


def kane_eval_c(bodies, q, v, qacc, grav, armature):
    """F = M qacc + bias in MuJoCo coordinates (as kane_residual) and, per foot, the spatial velocity and acceleration of
    the ankle link in LINK coordinates, Featherstone convention: [omega; v_O] and [alpha; a_O - omega x v_O], plus the world
    up axis in link coordinates."""
    nb = len(bodies)
    dt = np.result_type(q.dtype, v.dtype, qacc.dtype)
    R = [None] * nb; p = [None] * nb; om = [None] * nb; al = [None] * nb; acc = [None] * nb; z = [None] * nb; vo = [None] * nb
    R[0] = q2R_c(q[3:7]); p[0] = q[0:3].astype(dt)
    om[0] = R[0] @ v[3:6]; al[0] = R[0] @ qacc[3:6]; acc[0] = qacc[0:3].astype(dt); vo[0] = v[0:3].astype(dt)
    for i in range(1, nb):
        b = bodies[i]; pa = b["parent"]
        R[i] = R[pa] @ (b["R"] @ axis_rot_c(b["axis"], q[7 + i - 1]))
        d = R[pa] @ b["pos"]
        p[i] = p[pa] + d
        z[i] = R[i] @ b["axis"]
        qd, qdd = v[6 + i - 1], qacc[6 + i - 1]
        om[i] = om[pa] + z[i] * qd
        al[i] = al[pa] + np.cross(om[pa], z[i]) * qd + z[i] * qdd
        acc[i] = acc[pa] + np.cross(al[pa], d) + np.cross(om[pa], np.cross(om[pa], d))
        vo[i] = vo[pa] + np.cross(om[pa], d)
    F = np.zeros(25, dtype=dt)
    for i in range(nb):
        b = bodies[i]
        e = R[i] @ b["c"]
        ci = p[i] + e
        ac = acc[i] + np.cross(al[i], e) + np.cross(om[i], np.cross(om[i], e))
        Iw = R[i] @ b["I"] @ R[i].T
        fl_ = b["m"] * (ac - grav)
        tq = Iw @ al[i] + np.cross(om[i], Iw @ om[i])
        F[0:3] += fl_
        F[3:6] += R[0].T @ (np.cross(ci - p[0], fl_) + tq)
        a = i
        while a > 0:
            F[6 + a - 1] += z[a] @ (np.cross(ci - p[a], fl_) + tq)
            a = bodies[a]["parent"]
    F[6:] += armature * qacc[6:]
    feet = []
    for f in FEET:
        wl, vl = R[f].T @ om[f], R[f].T @ vo[f]
        feet.append(dict(vel=np.concatenate([wl, vl]), acc=np.concatenate([R[f].T @ al[f], R[f].T @ acc[f] - np.cross(wl, vl)]), up=R[f].T @ np.array([0.0, 0.0, 1.0])))
    return F, feet


def integrate_c(q, v, qacc, h):
    vn = v + h * qacc
    qn = q.astype(vn.dtype).copy()
    qn[0:3] += h * vn[0:3]
    qn[7:] += h * vn[6:]
    quat = q[3:7] / cnorm(q[3:7])
    w = vn[3:6]
    s = (w * w).sum() * h * h                       # (angle)^2
    if abs(s.real if np.iscomplexobj(s) else s) < 1e-12:
        c, so = 1 - s / 8 + s * s / 384, 0.5 - s / 48 + s * s / 3840       # cos(a/2), sin(a/2)/a
    else:
        a_ = np.sqrt(s); c, so = np.cos(a_ / 2), np.sin(a_ / 2) / a_
    e = np.concatenate([[c], so * h * w])
    a, b = quat, e
    r = np.array([a[0] * b[0] - a[1:] @ b[1:], *(a[0] * b[1:] + b[0] * a[1:] + np.cross(a[1:], b[1:]))])
    qn[3:7] = r / cnorm(r)
    return np.concatenate([qn, vn])


def kane_step_c(bodies, ctrlrange, x, u, h, grav, stance=(1, 1), contact=0, soft=1e-5, damping=1.0, armature=0.1, want=False, mu=1.0, lock=(), lock_acc=None):
    """One MuJoCo-semantics step of the constraint-free plant (contact = 0) or with the feet the schedule
    marks as stance held by velocity-level rigid constraints over the step (contact = 1 bilateral, 2 unilateral: a foot whose
    normal force would pull is released and the rest solved again, once):
        Mhat qacc - J^T lambda = tau - D v - bias ,   J qacc + gamma + v_f / h + soft lambda = 0 ,   Mhat = M + armature + h D
    with J, gamma from a_f(qacc) = gamma + J qacc (spatial acceleration of the ankle link in link coordinates).
    contact = 3 (round 4): unilateral + Coulomb limit.  A foot whose constraint force leaves the friction cone, |f_t| > mu f_n
    (f_n = force along the world up axis, f_t the rest of the force; MuJoCo's default sliding friction is mu = 1), cannot stick:
    its two tangential translation rows are dropped -- the rotation rows and the normal row stay -- and the set is solved again, once
    (`release` variant: a slipping foot receives no tangential force).
    contact = 4: the same decision, but the sliding foot keeps kinetic friction: a tangential force mu lambda_n along the unit direction in
    which the sticking solution pulled (the direction that opposes the slip), i.e. its normal multiplier acts along up + mu t while the
    constraint row stays the normal one -- an unsymmetric system, solved once.
    lock: hinges (0..18) whose acceleration is prescribed, qacc_i = -v_i / h (joint-limit rows, kane_step_lim): rigid rows E qacc = -v_L / h
    with multipliers of their own in every system solved here; lock_acc (optional, one value per locked hinge): the prescribed
    accelerations instead (kane_step_lim with a restoring stiffness: -v_i / h - k r_i)."""
    q, v = x[:26], x[26:]
    z25 = np.zeros(25)
    bias, feet0 = kane_eval_c(bodies, q, v, z25, grav, armature)
    cols = [kane_eval_c(bodies, q, v, np.eye(25)[k], grav, armature) for k in range(25)]
    M = np.stack([c[0] - bias for c in cols], axis=1)
    D = np.concatenate([np.zeros(6), damping * np.ones(19)])
    ur = u.real if np.iscomplexobj(u) else u
    uc = np.where(ur < ctrlrange[:, 0], ctrlrange[:, 0], np.where(ur > ctrlrange[:, 1], ctrlrange[:, 1], u))
    tau = np.concatenate([np.zeros(6), uc])
    Mh = M + h * np.diag(D)
    rhs = tau - D * v - bias
    act = [bool(contact) and stance[0] == 1, bool(contact) and stance[1] == 1]
    slide = [False, False]
    tdir = [None, None]
    lam = np.zeros(12, dtype=rhs.dtype)
    re_ = (lambda a: a.real) if np.iscomplexobj(rhs) else (lambda a: a)
    nl = len(lock)
    E = np.zeros((nl, 25)); bl = np.zeros(nl, dtype=rhs.dtype)
    for r_, i_ in enumerate(lock):
        E[r_, 6 + i_] = 1.0; bl[r_] = -v[6 + i_] / h if lock_acc is None else lock_acc[r_]
    stage = 0                      # 0: rigid set; 1: after the unilateral check; 2: after the Coulomb check
    while True:
        rows = [f for f in range(2) if act[f]]
        if not rows:
            if nl:
                qacc = np.linalg.solve(np.block([[Mh, -E.T], [E, np.zeros((nl, nl))]]), np.concatenate([rhs, bl]))[:25]
            else:
                qacc = np.linalg.solve(Mh, rhs)
            lam[:] = 0
            break
        # row selection per foot: all six components of the link's spatial velocity, or -- sliding -- rotation + normal translation
        Ssel = []
        for f in rows:
            if slide[f]:
                Sf = np.zeros((4, 6), dtype=feet0[f]["up"].dtype); Sf[0:3, 0:3] = np.eye(3); Sf[3, 3:6] = feet0[f]["up"]
            else:
                Sf = np.eye(6)
            Ssel.append(Sf)
        # force map of the multipliers: the constraint rows themselves, except (contact = 4) the normal multiplier of a sliding foot, which
        # also carries the kinetic friction mu lambda_n along tdir (the direction in which the sticking solution pulled)
        Fsel = []
        for Sf, f in zip(Ssel, rows):
            Ff = Sf.copy()
            if slide[f] and contact == 4:
                Ff[3, 3:6] = feet0[f]["up"] + mu * tdir[f]
            Fsel.append(Ff)
        Jfull = {f: np.stack([cols[k][1][f]["acc"] - feet0[f]["acc"] for k in range(25)], axis=1) for f in rows}
        J = np.concatenate([Sf @ Jfull[f] for Sf, f in zip(Ssel, rows)], axis=0)
        JF = np.concatenate([Ff @ Jfull[f] for Ff, f in zip(Fsel, rows)], axis=0)
        b = np.concatenate([Sf @ (-feet0[f]["vel"] / h - feet0[f]["acc"]) for Sf, f in zip(Ssel, rows)])
        nc = J.shape[0]
        if nl:
            KKT = np.block([[Mh, -JF.T, -E.T], [J, soft * np.eye(nc), np.zeros((nc, nl))], [E, np.zeros((nl, nc)), np.zeros((nl, nl))]])
            sol = np.linalg.solve(KKT, np.concatenate([rhs, b, bl]))
        else:
            KKT = np.block([[Mh, -JF.T], [J, soft * np.eye(nc)]])
            sol = np.linalg.solve(KKT, np.concatenate([rhs, b]))
        qacc = sol[:25]
        lam[:] = 0
        o = 25
        for Ff, f in zip(Fsel, rows):
            lam[6 * f:6 * f + 6] = Ff.T @ sol[o:o + Ff.shape[0]]; o += Ff.shape[0]
        again = False
        if contact >= 2 and stage == 0:
            stage = 1
            for f in rows:
                fz = feet0[f]["up"] @ lam[6 * f + 3:6 * f + 6]
                if re_(fz) < 0.0:
                    act[f] = False; again = True
            if again:
                continue
        if contact >= 3 and stage == 1:
            stage = 2
            for f in rows:
                if not act[f]:
                    continue
                fo = lam[6 * f + 3:6 * f + 6]
                fn = feet0[f]["up"] @ fo
                ft2 = fo @ fo - fn * fn
                if re_(ft2) > mu * mu * re_(fn) * re_(fn):
                    slide[f] = True; again = True
                    tdir[f] = (fo - fn * feet0[f]["up"]) / np.sqrt(ft2)
            if again:
                continue
        break
    xn = integrate_c(q, v, qacc, h)
    if want:
        return xn, qacc, Mh, lam, act, slide
    return xn


def kane_step_lim(bodies, ctrlrange, x, u, h, grav, stiffness=0.0, **kw):
    """The step with joint-limit rows (h1.xml jnt_range, enforced by mj_step; restated rigid and at velocity level like the stance rows):
    a hinge past its range that the unlimited step would still move outward, (q_i > hi_i and v_i + h qacc_i > 0) or (q_i < lo_i and
    v_i + h qacc_i < 0), is stopped over the step: E qacc = -v_L / h joins the KKT system, and the step (stance decisions included) is
    taken again with that set.  Returns (x_next, lock set, x_next of the unlimited step, margins of the decisions).
    stiffness k > 0 (round 6): the hard limit of MuJoCo's solref reference acceleration a_ref = -b v - k r (r_i = q_i - hi_i > 0 or
    q_i - lo_i < 0 the violation; b = 1 / h is MuJoCo's damping for its clamped time constant 2 h, k = 1 / (2 h)^2 its stiffness): the
    row prescribes qacc_i = -v_i / h - k r_i, i.e. v_i+ = -h k r_i pushes the hinge back, and the row is active when the unlimited
    step falls short of that acceleration on the outward side: q_i > hi_i and qacc_i > -v_i / h - k r_i (mirrored below lo_i)."""
    kw = dict(kw); kw.pop("want", None)
    xn0, qacc0, _, _, _, _ = kane_step_c(bodies, ctrlrange, x, u, h, grav, want=True, **kw)
    rng = np.array([b["rng"] for b in bodies if b["rng"] is not None])
    assert rng.shape == (19, 2)
    th, qd = x[7:26], x[32:51]
    r = np.where(th > rng[:, 1], th - rng[:, 1], np.where(th < rng[:, 0], th - rng[:, 0], 0.0))
    vnext = qd + h * (qacc0[6:] + stiffness * r)          # (v_i + h qacc_i for k = 0)
    lock = [i for i in range(19) if (th[i] > rng[i, 1] and vnext[i] > 0.0) or (th[i] < rng[i, 0] and vnext[i] < 0.0)]
    viol = [i for i in range(19) if th[i] > rng[i, 1] or th[i] < rng[i, 0]]
    margin = min([abs(vnext[i]) for i in viol], default=np.inf)
    if not lock:
        return xn0, lock, xn0, margin
    acc = None if stiffness == 0.0 else [-qd[i] / h - stiffness * r[i] for i in lock]
    xn = kane_step_c(bodies, ctrlrange, x, u, h, grav, lock=tuple(lock), lock_acc=acc, **kw)
    return xn, lock, xn0, margin


def jacobians_complex_step(f, x, u, eps=1e-30):
    """A = df/dx, B = df/du of a complex-safe step f(x, u) by complex-step differentiation (exact to rounding)."""
    nx, nu = x.shape[0], u.shape[0]
    A = np.zeros((nx, nx)); B = np.zeros((nx, nu))
    for i in range(nx):
        xp = x.astype(complex); xp[i] += 1j * eps
        A[:, i] = f(xp, u.astype(complex)).imag / eps
    for i in range(nu):
        up = u.astype(complex); up[i] += 1j * eps
        B[:, i] = f(x.astype(complex), up).imag / eps
    return A, B


def jacobians_tangent_free(bodies, ctrlrange, x, u, h, grav, damping=1.0, armature=0.1, eps=1e-30):
    """The same Jacobians for the constraint-free step at 1/25 of the cost: d qacc / d z = -Mhat^-1 d r / d z with the residual
    r = F(q, v, qacc) + D v - tau differentiated at fixed qacc (ONE complex residual evaluation per state coordinate instead of
    a whole step with its 26), then the integrator by complex step."""
    q, v = x[:26], x[26:]
    xn, qacc, Mh, _, _, _ = kane_step_c(bodies, ctrlrange, x, u, h, grav, want=True)
    D = np.concatenate([np.zeros(6), damping * np.ones(19)])
    free_u = ((u >= ctrlrange[:, 0]) & (u <= ctrlrange[:, 1])).astype(float)
    Mi = np.linalg.inv(Mh)
    nx, nu = 51, 19
    A = np.zeros((nx, nx)); B = np.zeros((nx, nu))
    for col in range(nx + nu):
        dq = np.zeros(25)
        if 3 <= col < nx:
            xp = x.astype(complex); xp[col] += 1j * eps
            dr = kane_eval_c(bodies, xp[:26], xp[26:], qacc.astype(complex), grav, armature)[0].imag / eps
            if col >= 26:
                dr[col - 26] += D[col - 26]
            dq = -Mi @ dr
        elif col >= nx:
            dq = Mi[:, 6 + col - nx] * free_u[col - nx]
        xp = x.astype(complex)
        if col < nx:
            xp[col] += 1j * eps
        d = integrate_c(xp[:26], xp[26:], qacc + 1j * eps * dq, h).imag / eps
        if col < nx:
            A[:, col] = d
        else:
            B[:, col - nx] = d
    return A, B


def gen_dynamics():
    bodies, ctrl = load_mjcf()
    rng = np.random.default_rng(123)
    xs, us, gs, xn, qa = [], [], [], [], []
    for k in range(6):
        x = np.zeros(51); x[2] = 1.0432; x[3] = 1.0
        if k > 0:
            x[0:3] += rng.uniform(-0.1, 0.1, 3)
            qq = rng.normal(size=4); x[3:7] = qq / np.linalg.norm(qq) if k > 2 else x[3:7]
            x[7:26] = rng.uniform(-0.4, 0.4, 19)
            x[26:] = rng.uniform(-1.0, 1.0, 25) * (0.0 if k == 1 else 1.0)
        u = rng.uniform(-30, 30, 19)
        if k == 4:
            u[3] = 500.0; u[13] = -100.0  # exercise ctrl clamping
        g = np.array([0, 0, -9.81]) if k % 2 == 0 else np.array([0.0, 0.0, -1.0])
        if k == 5:
            x[3:7] *= 1.3  # un-normalised quaternion input (normalised inside kinematics)
        xnext, qacc, M, bias = kane_step(bodies, ctrl, x, u, 0.02, g)
        xs.append(x); us.append(u); gs.append(g); xn.append(xnext); qa.append(qacc)
    np.savez(os.path.join(HERE, "dynamics_golden.npz"), x=np.array(xs), u=np.array(us), gravity=np.array(gs), x_next=np.array(xn), qacc=np.array(qa), h=0.02)
    print("dynamics golden: M cond", np.linalg.cond(M))


# ----------------------------------------------------------------------------- torch cost terms
def t_axis_rot(k, th):
    c, s = torch.cos(th), torch.sin(th)
    o, z = torch.ones(()), torch.zeros(())
    if k == 0:
        rows = [[o, z, z], [z, c, -s], [z, s, c]]
    elif k == 1:
        rows = [[c, z, s], [z, o, z], [-s, z, c]]
    else:
        rows = [[c, -s, z], [s, c, z], [z, z, o]]
    return torch.stack([torch.stack(r) for r in rows])


def rpy_R(rpy):
    r, p, y = rpy
    Rx = np.array([[1, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]])
    Ry = np.array([[math.cos(p), 0, math.sin(p)], [0, 1, 0], [-math.sin(p), 0, math.cos(p)]])
    Rz = np.array([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


class PinTorch:
    """Pinocchio-convention FK (free-flyer + revolute), spatial velocities propagated body-frame
    the way pinocchio::forwardKinematics does, written directly in torch."""

    def __init__(self):
        self.L = load_urdf()

    def fk(self, xp):
        L = self.L
        qx, qy, qz, qw = xp[3], xp[4], xp[5], xp[6]
        R0 = torch.stack([
            torch.stack([1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)]),
            torch.stack([2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)]),
            torch.stack([2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)])])
        oR = [R0]; op = [xp[0:3]]
        vl = [xp[26:29]]; va = [xp[29:32]]  # body-frame spatial velocity (linear, angular)
        for i in range(1, len(L)):
            b = L[i]; pa = b["parent"]
            k = int(np.argmax(np.abs(b["axis"])))
            Rj = torch.tensor(rpy_R(b["rpy"])) @ t_axis_rot(k, xp[7 + i - 1])
            pj = torch.tensor(b["pos"])
            oR.append(oR[pa] @ Rj)
            op.append(op[pa] + oR[pa] @ pj)
            # v_i = liMi.actInv(v_parent) + S qd
            w_p, v_p = va[pa], vl[pa]
            lin = Rj.T @ (v_p + torch.linalg.cross(w_p, pj))
            ang = Rj.T @ w_p + torch.tensor(np.eye(3)[k]) * xp[26 + 6 + i - 1]
            vl.append(lin); va.append(ang)
        return oR, op, vl, va

    def com(self, xp, with_vel=False):
        L = self.L
        oR, op, vl, va = self.fk(xp)
        M = sum(b["m"] for b in L)
        c = sum(b["m"] * (op[i] + oR[i] @ torch.tensor(b["c"])) for i, b in enumerate(L)) / M
        if not with_vel:
            return c
        vc = sum(b["m"] * (oR[i] @ (vl[i] + torch.linalg.cross(va[i], torch.tensor(b["c"])))) for i, b in enumerate(L)) / M
        return c, vc

    def frame(self, xp, idx):
        oR, op, vl, va = self.fk(xp)
        return op[idx], oR[idx] @ vl[idx]  # position; LOCAL_WORLD_ALIGNED linear velocity


def gen_costs():
    pin = PinTorch()
    rng = np.random.default_rng(7)
    out = {}
    xs = []
    for k in range(3):
        x = np.zeros(51); x[2] = 1.0432; x[3] = 1.0
        x[0:3] += rng.uniform(-0.05, 0.05, 3)
        qq = np.array([1.0, 0, 0, 0]) + rng.uniform(-0.2, 0.2, 4) * (k > 0); x[3:7] = qq / np.linalg.norm(qq)
        x[7:26] = rng.uniform(-0.3, 0.3, 19)
        x[26:] = rng.uniform(-0.5, 0.5, 25)
        xs.append(x)
    xs = np.array(xs)
    refs = dict(com=np.array([0.01, -0.02, 0.95]), comvel=np.array([0.1, 0.0, -0.05]), ee=np.array([0.03, 0.21, 0.08]),
                ps=np.array([0.02, 0.01]))
    w = dict(com=100.0, comvel=7.0, eepos=400.0, eevel=400.0, upright=20.0, balance=30.0)

    def perm(x):
        xp = x.clone(); xp[3], xp[4], xp[5], xp[6] = x[4], x[5], x[6], x[3]
        return xp

    terms = {
        "com": lambda xp: w["com"] * ((pin.com(xp) - torch.tensor(refs["com"])) ** 2).sum(),
        "comvel": lambda xp: w["comvel"] * ((pin.com(xp, True)[1] - torch.tensor(refs["comvel"])) ** 2).sum(),
        "eepos_L": lambda xp: w["eepos"] * ((pin.frame(xp, 5)[0] - torch.tensor(refs["ee"])) ** 2).sum(),
        "eepos_R": lambda xp: w["eepos"] * ((pin.frame(xp, 10)[0] - torch.tensor(refs["ee"] * np.array([1, -1, 1]))) ** 2).sum(),
        "eevel_L": lambda xp: w["eevel"] * (pin.frame(xp, 5)[1] ** 2).sum(),
        "eevel_R": lambda xp: w["eevel"] * (pin.frame(xp, 10)[1] ** 2).sum(),
    }

    def upright(xp):  # derivatives.cpp:646-666 (slot labelling quirk kept)
        qw, qx, qy, qz = xp[3], xp[4], xp[5], xp[6]
        r = torch.stack([2 * (qx * qz + qw * qy), 2 * (qy * qz - qw * qx), (1 - 2 * (qx * qx + qy * qy)) - 1.0])
        return 0.5 * w["upright"] * (r ** 2).sum()

    def balance(xp):  # derivatives.cpp:668-707
        c, vc = pin.com(xp, True)
        om = torch.sqrt(c[2] / 9.81)
        r = c[0:2] + vc[0:2] * om - torch.tensor(refs["ps"])
        return 0.5 * w["balance"] * (r ** 2).sum()

    terms["upright"] = upright
    terms["balance"] = balance
    for name, f in terms.items():
        G, H = [], []
        for x in xs:
            xp = perm(torch.tensor(x)).requires_grad_(True)
            g = torch.autograd.grad(f(xp), xp)[0]
            Hm = torch.autograd.functional.hessian(f, perm(torch.tensor(x)))
            G.append(g.numpy()); H.append(Hm.numpy())
        out["grad_" + name] = np.array(G); out["hess_" + name] = np.array(H)
    out["x"] = xs
    for k, v in refs.items():
        out["ref_" + k] = v
    for k, v in w.items():
        out["w_" + k] = v

    # MuJoCo-side quantities for computeTotalCost: CoM with MJCF masses, true-quaternion upright
    bodies, ctrl = load_mjcf()
    coms, ees = [], []
    for x in xs:
        nb = len(bodies); R = [None] * nb; p = [None] * nb
        R[0] = q2R(x[3:7]); p[0] = x[0:3]
        for i in range(1, nb):
            b = bodies[i]; pa = b["parent"]
            R[i] = R[pa] @ b["R"] @ axis_rot(b["axis"], x[7 + i - 1]); p[i] = p[pa] + R[pa] @ b["pos"]
        M = sum(b["m"] for b in bodies)
        coms.append(sum(b["m"] * (p[i] + R[i] @ b["c"]) for i, b in enumerate(bodies)) / M)
        ees.append(np.stack([p[5], p[10]]))
    out["com_mj"] = np.array(coms); out["ee_mj"] = np.array(ees)
    out["jrange"] = np.array([b["rng"] for b in bodies[1:]]); out["ctrlrange"] = ctrl
    np.savez(os.path.join(HERE, "cost_golden.npz"), **out)
    print("cost golden written:", sorted(k for k in out if k.startswith("hess_")))


# ----------------------------------------------------------------------------- Riccati (numpy)
def riccati_numpy(A, B, lx, lu, lxx, luu, lam):
    """ilqr.cpp:250-309 with NumPy (np.linalg.cholesky for the LLT check, np.linalg.solve for ldlt().solve)."""
    N = A.shape[0]; n, m = B.shape[1], B.shape[2]
    Vx, Vxx = lx[N].copy(), lxx[N].copy()
    K = np.zeros((N, m, n)); k = np.zeros((N, m)); bumped = []
    for t in range(N - 1, -1, -1):
        Qx = lx[t] + A[t].T @ Vx
        Qu = lu[t] + B[t].T @ Vx
        Qxx = lxx[t] + A[t].T @ Vxx @ A[t]
        Quu = np.diag(luu[t]) + B[t].T @ Vxx @ B[t]
        Qxu = A[t].T @ Vxx @ B[t]
        Quu = Quu + lam * np.eye(m)
        try:
            np.linalg.cholesky(Quu)
        except np.linalg.LinAlgError:
            Quu = Quu + 1e-4 * np.eye(m); bumped.append(t)
        K[t] = -np.linalg.solve(Quu, Qxu.T)
        k[t] = -np.linalg.solve(Quu, Qu)
        Vx = Qx + K[t].T @ Quu @ k[t] + K[t].T @ Qu + Qxu @ k[t]
        Vxx = Qxx + K[t].T @ Quu @ K[t] + K[t].T @ Qxu.T + Qxu @ K[t]
        Vxx = 0.5 * (Vxx + Vxx.T)
    return K, k, Vx, Vxx, bumped


def gen_riccati():
    rng = np.random.default_rng(99)
    N, n, m = 6, 51, 19
    cases = {}
    for name in ("spd", "bump"):
        A = np.eye(n)[None] + 0.05 * rng.normal(size=(N, n, n))
        B = 0.1 * rng.normal(size=(N, n, m))
        lx = rng.normal(size=(N + 1, n)); lu = rng.normal(size=(N, m))
        lxx = np.zeros((N + 1, n, n))
        for t in range(N + 1):
            G = rng.normal(size=(n, n)); lxx[t] = G @ G.T / n + np.diag(rng.uniform(1, 100, n))
        luu = rng.uniform(1e-3, 1e-2, size=(N, m))
        if name == "bump":  # indefinite terminal Hessian + tiny R -> Quu not PD at some knots
            lxx[N] = lxx[N] - 2.0 * np.diag(np.diag(lxx[N])); luu[:] = 1e-9
        K, k, Vx, Vxx, bumped = riccati_numpy(A, B, lx, lu, lxx, luu, 1e-6)
        cases[name] = dict(A=A, B=B, lx=lx, lu=lu, lxx=lxx, luu=luu, K=K, k=k, Vx=Vx, Vxx=Vxx, bumped=np.array(bumped, dtype=np.int64))
        print("riccati", name, "bumped knots", bumped, "|K|max", np.abs(K).max())
    flat = {f"{c}_{k}": v for c, d in cases.items() for k, v in d.items()}
    np.savez_compressed(os.path.join(HERE, "riccati_golden.npz"), lam=1e-6, **flat)


# ----------------------------------------------------------------------------- reference data excerpt
def gen_refdata():
    q = np.loadtxt(os.path.join(REF, "data/q_ref2_mj.csv"), delimiter=",")
    v = np.loadtxt(os.path.join(REF, "data/v_ref2.csv"), delimiter=",")
    qp = np.loadtxt(os.path.join(REF, "data/q_ref2_pin.csv"), delimiter=",")
    cw = np.loadtxt(os.path.join(REF, "data/contact_walking.csv"), delimiter=",", skiprows=1).astype(np.int32)
    rows = slice(0, 80)
    # contact-schedule tool (get_contacts.py:96-147): the reference's own input / output pairs (q_ref2_mj.csv ->
    # contact_walking.csv, q_standing.csv -> contact_standing.csv) and, for rows of h1_walking_pin.csv beyond the 400
    # that ship with a schedule, foot clearances from an independent numpy FK over ALL vertices of the ankle STL
    import struct
    raw = open(os.path.join(REF, "robots/h1_description/meshes/left_ankle_link.STL"), "rb").read()
    ntri = struct.unpack("<I", raw[80:84])[0]
    verts = np.unique(np.frombuffer(raw[84:84 + 50 * ntri], dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))["v"].reshape(-1, 3).astype(np.float64), axis=0)
    bodies, _ = load_mjcf()
    names = [b["name"] for b in bodies]
    feet = [names.index("left_ankle_link"), names.index("right_ankle_link")]

    def axis_R(ax, th):
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K

    def clearance(qrow):
        R, p = [q2R(qrow[3:7])], [qrow[:3]]
        for i in range(1, len(bodies)):
            b = bodies[i]
            R.append(R[b["parent"]] @ b["R"] @ axis_R(b["axis"], qrow[6 + i]))
            p.append(p[b["parent"]] + R[b["parent"]] @ b["pos"])
        return [float((verts @ R[i].T + p[i])[:, 2].min()) for i in feet]

    clr2 = np.array([clearance(r) for r in q])
    assert np.array_equal((clr2 < 0).astype(np.int32), cw), "hull-below-plane rule does not reproduce contact_walking.csv"
    qs = np.loadtxt(os.path.join(REF, "data/q_standing.csv"), delimiter=",")
    cs = np.loadtxt(os.path.join(REF, "data/contact_standing.csv"), delimiter=",", skiprows=1).astype(np.int32)
    wp = np.loadtxt(os.path.join(REF, "data/h1_walking_pin.csv"), delimiter=",")[360:560]
    wp_mj = wp.copy(); wp_mj[:, 3] = wp[:, 6]; wp_mj[:, 4:7] = wp[:, 3:6]
    clrw = np.array([clearance(r) for r in wp_mj])
    np.savez_compressed(os.path.join(HERE, "refdata_golden.npz"), q_ref2_mj=q[rows], v_ref2=v[rows], q_ref2_pin=qp[rows], contact_walking=cw[rows], dt=0.02,
                        q_ref2_mj_full=q, contact_walking_full=cw, clearance_ref2=clr2, q_standing=qs[:3], contact_standing=cs[:3],
                        walking_pin_rows=wp, walking_pin_row0=360, walking_pin_clearance=clrw)
    print("refdata rows", q[rows].shape, v[rows].shape, "contact flags reproduced:", cw.shape, "walking_pin stance counts", (clrw < 0).sum(0), "min |clr|", np.abs(clrw).min(), np.abs(clr2).min())


# ----------------------------------------------------------------------------- whole solve loop (numpy / torch)
def gen_solve(N=6, out_name="solve_golden.npz", n_seeds=2, max_iter=3, jac="central", contact=0, grav=(0.0, 0.0, -1.0), rng_seed=2024,
              swing=(2, 4), u_scale=1.0, u_bias=None):
    """(Defaults = the round-2 golden, file for file.  Round 3 adds: jac = "tangent" -- exact Jacobians of the constraint-free
    Kane step by complex-step differentiation, N = 25 -- and contact = 2 with jac = "complex": the stance-constrained step as a
    NumPy KKT system with complex-step Jacobians of the whole constrained step.)
    NumPy / torch float64 restatement of the WHOLE iLQR::solve loop (ilqr.cpp:521-660) on a short horizon, glued from the
    independent pieces above: rollout through the Kane step, Jacobians by central differences of that step (eps 1e-4: a
    numerical stand-in for the exact derivatives the HIP path computes analytically), cost quadratics from the
    torch-autograd task terms (Pinocchio conventions, permutation quirk), NumPy Riccati, 8-alpha line search with the
    reference's lambda / retry / exit rules.  Pins the glue between the individually pinned stages (cost trace, accepted
    step sizes, gains) for the oracle (CPU test) and the HIP path (-m gpu)."""
    bodies, ctrl = load_mjcf()
    pin = PinTorch()
    h = 0.02
    grav = np.array(grav, dtype=float)
    nq, nx, nu = 26, 51, 19
    Q = np.ones(nx); Q[0], Q[1], Q[2] = 200.0, 50.0, 200.0; Q[3] = 50.0; Q[4:7] = 50.0; Q[7:nq] = 50.0
    Q[nq], Q[nq + 1], Q[nq + 2] = 150.0, 50.0, 150.0; Q[nq + 3:nq + 6] = 75.0; Q[nq + 6:] = 75.0
    R = np.ones(nu) * 0.001
    Qf = Q * 2.0; Qf[0] *= 5.0; Qf[1] *= 2.0; Qf[2] *= 5.0; Qf[nq + 2] *= 4.0
    w = dict(com=100.0, comvel=0.0, eepos=400.0, eevel=400.0, upright=20.0, balance=30.0, joint=1500.0, ctrl=1500.0)
    jr = np.array([b["rng"] for b in bodies[1:]])
    stance = np.ones((N + 1, 2), dtype=np.int32)
    if swing is not None:
        stance[swing[0]:swing[1], 0] = 0                                     # the left foot swings at these knots (default: 2, 3)
    xs = np.zeros(nx); xs[2] = 1.0432; xs[3] = 1.0
    x_ref = np.tile(xs, (N + 1, 1)); u_ref = np.zeros((N, nu))

    def fk_mj(x):
        nb = len(bodies); Rw = [None] * nb; p = [None] * nb
        Rw[0] = q2R(x[3:7]); p[0] = x[0:3]
        for i in range(1, nb):
            b = bodies[i]; pa = b["parent"]
            Rw[i] = Rw[pa] @ b["R"] @ axis_rot(b["axis"], x[7 + i - 1]); p[i] = p[pa] + Rw[pa] @ b["pos"]
        M = sum(b["m"] for b in bodies)
        return sum(b["m"] * (p[i] + Rw[i] @ b["c"]) for i, b in enumerate(bodies)) / M, np.stack([p[5], p[10]])

    com_ref = np.zeros((N + 1, 3)); ee_ref = np.zeros((N + 1, 2, 3))
    for t in range(N + 1):
        com_ref[t], ee_ref[t] = fk_mj(x_ref[t])
    if swing is not None:
        ee_ref[swing[0]:swing[1], 0, 2] += 0.03                              # swing-foot target above the ground

    def f(x, u, t=0):
        if contact or jac != "central":
            return kane_step_c(bodies, ctrl, x, u, h, grav, stance=stance[t], contact=contact)
        return kane_step(bodies, ctrl, x, u, h, grav)[0]

    def limits(rng):
        m = 0.1 * (rng[:, 1] - rng[:, 0])
        return rng[:, 0] + m, rng[:, 1] - m

    jlo, jhi = limits(jr); ulo, uhi = limits(ctrl)

    def support(t):
        L, Rt = stance[t, 0] == 1, stance[t, 1] == 1
        if L and Rt:
            return 0.5 * (ee_ref[t, 0, :2] + ee_ref[t, 1, :2])
        if L:
            return ee_ref[t, 0, :2]
        if Rt:
            return ee_ref[t, 1, :2]
        return None

    def knot_cost(t, x, u):          # ilqr.cpp:363-518
        Qd = Qf if t == N else Q
        e = x - x_ref[t]
        c = 0.5 * e @ (Qd * e)
        if u is not None:
            eu = u - u_ref[t]; c += 0.5 * eu @ (R * eu)
        qw, qx, qy, qz = x[3:7]
        r = np.array([2 * (qx * qz + qw * qy), 2 * (qy * qz - qw * qx), 1 - 2 * (qx * qx + qy * qy) - 1.0])
        c += 0.5 * w["upright"] * r @ r
        ps = support(t)
        if ps is not None:
            com, _ = fk_mj(x)
            om = math.sqrt(com[2] / 9.81)
            rb = com[:2] + x[nq:nq + 2] * om - ps
            c += 0.5 * w["balance"] * rb @ rb
        uu = np.zeros(nu) if u is None else u
        c += w["ctrl"] * (np.maximum(uu - uhi, 0) ** 2 + np.maximum(ulo - uu, 0) ** 2).sum()
        q = x[7:nq]
        c += w["joint"] * (np.maximum(q - jhi, 0) ** 2 + np.maximum(jlo - q, 0) ** 2).sum()
        return c

    def total_cost(X, U):
        return sum(knot_cost(t, X[t], U[t]) for t in range(N)) + knot_cost(N, X[N], None)

    def perm(x):
        xp = x.clone(); xp[3], xp[4], xp[5], xp[6] = x[4], x[5], x[6], x[3]
        return xp

    def task_terms(t):
        terms = [lambda xp: w["com"] * ((pin.com(xp) - torch.tensor(com_ref[t])) ** 2).sum()]
        for ee, idx in ((0, 5), (1, 10)):
            if stance[t, ee] != 1:
                terms.append(lambda xp, ee=ee, idx=idx: w["eepos"] * ((pin.frame(xp, idx)[0] - torch.tensor(ee_ref[t, ee])) ** 2).sum())
            else:
                terms.append(lambda xp, idx=idx: w["eevel"] * (pin.frame(xp, idx)[1] ** 2).sum())

        def upright(xp):
            qw, qx, qy, qz = xp[3], xp[4], xp[5], xp[6]
            r = torch.stack([2 * (qx * qz + qw * qy), 2 * (qy * qz - qw * qx), (1 - 2 * (qx * qx + qy * qy)) - 1.0])
            return 0.5 * w["upright"] * (r ** 2).sum()
        terms.append(upright)
        ps = support(t)
        if ps is not None:
            def balance(xp):
                c, vc = pin.com(xp, True)
                om = torch.sqrt(c[2] / 9.81)
                r = c[0:2] + vc[0:2] * om - torch.tensor(ps)
                return 0.5 * w["balance"] * (r ** 2).sum()
            terms.append(balance)
        return terms

    def quadratics(X, U):            # ilqr.cpp:133-244
        lx = np.zeros((N + 1, nx)); lu = np.zeros((N, nu)); lxx = np.zeros((N + 1, nx, nx)); luu = np.zeros((N, nu))
        for t in range(N + 1):
            Qd = Qf if t == N else Q
            lx[t] = Qd * (X[t] - x_ref[t]); lxx[t] = np.diag(Qd)
            fsum = lambda xp: sum(term(xp) for term in task_terms(t))
            xp0 = perm(torch.tensor(X[t]))
            g = torch.autograd.grad(fsum(xp0.clone().requires_grad_(True)), None) if False else None
            xpr = xp0.clone().requires_grad_(True)
            lx[t] += torch.autograd.grad(fsum(xpr), xpr)[0].numpy()          # slots of the permuted state, added as they are
            lxx[t] += torch.autograd.functional.hessian(fsum, xp0).numpy()
            q = X[t, 7:nq]
            lx[t, 7:nq] += 2 * w["joint"] * (np.maximum(q - jhi, 0) - np.maximum(jlo - q, 0))
            lxx[t][np.arange(7, nq), np.arange(7, nq)] += 2 * w["joint"] * ((q > jhi) | (q < jlo))
            if t < N:
                lu[t] = R * (U[t] - u_ref[t]) + 2 * w["ctrl"] * (np.maximum(U[t] - uhi, 0) - np.maximum(ulo - U[t], 0))
                luu[t] = R + 2 * w["ctrl"] * ((U[t] > uhi) | (U[t] < ulo))
        return lx, lu, lxx, luu

    def linearize(X, U, eps=1e-4):
        A = np.zeros((N, nx, nx)); B = np.zeros((N, nx, nu))
        if jac == "tangent":
            for t in range(N):
                A[t], B[t] = jacobians_tangent_free(bodies, ctrl, X[t], U[t], h, grav)
            return A, B
        if jac == "complex":
            for t in range(N):
                A[t], B[t] = jacobians_complex_step(lambda xx, uu: f(xx, uu, t), X[t], U[t])
            return A, B
        for t in range(N):
            for i in range(nx):
                d = np.zeros(nx); d[i] = eps
                A[t][:, i] = (f(X[t] + d, U[t]) - f(X[t] - d, U[t])) / (2 * eps)
            for i in range(nu):
                d = np.zeros(nu); d[i] = eps
                B[t][:, i] = (f(X[t], U[t] + d) - f(X[t], U[t] - d)) / (2 * eps)
        return A, B

    alphas = [1.0, 0.8, 0.6, 0.4, 0.2, 0.1, 0.05, 0.01]

    def line_search(x0, X, U, K, k):
        J0 = total_cost(X, U)
        for a in alphas:
            Xn = [x0.copy()]; Un = []
            for t in range(N):
                u = U[t] + a * k[t] + K[t] @ (Xn[t] - X[t])
                Un.append(u); Xn.append(f(Xn[t], u, t))
            Jn = total_cost(np.array(Xn), np.array(Un))
            if Jn < J0 - 1e-6:
                return True, np.array(Xn), np.array(Un), Jn, a
        return False, X, U, J0, 0.0

    out = {}
    tol = 1e-4
    rng = np.random.default_rng(rng_seed)
    ub = np.zeros(nu) if u_bias is None else np.asarray(u_bias, dtype=float)
    uin = []
    for seed in range(n_seeds):
        x0 = xs.copy()
        x0[0:3] += rng.uniform(-0.02, 0.02, 3)
        wv = rng.uniform(-0.05, 0.05, 3); ang = np.linalg.norm(wv)
        x0[3:7] = np.concatenate([[math.cos(ang / 2)], math.sin(ang / 2) * wv / ang])
        x0[7:nq] += rng.uniform(-0.05, 0.05, 19); x0[nq:] += rng.uniform(-0.1, 0.1, 25)
        U = ub + u_scale * rng.uniform(-1.0, 1.0, (N, nu))
        uin.append(U.copy())
        X = [x0.copy()]
        for t in range(N):
            X.append(f(X[t], U[t], t))
        X = np.array(X)
        lam = 1e-6
        J = total_cost(X, U)
        trace_c, trace_a, trace_l = [J], [], []
        Ks = None
        it = 0
        for it_ in range(max_iter):
            it = it_ + 1
            Jprev = J
            X[0] = x0
            for t in range(N):
                X[t + 1] = f(X[t], U[t], t)
            A, B = linearize(X, U)
            lx, lu, lxx, luu = quadratics(X, U)
            lam_used = lam
            K, k, _, _, _ = riccati_numpy(A, B, lx, lu, lxx, luu, lam)
            ok, Xn, Un, Jn, a = line_search(x0, X, U, K, k)
            if not ok:
                lam = min(lam * 10.0, 1e-3); lam_used = lam
                K, k, _, _, _ = riccati_numpy(A, B, lx, lu, lxx, luu, lam)
                ok, Xn, Un, Jn, a = line_search(x0, X, U, K, k)
                if not ok:
                    trace_c.append(J); trace_a.append(0.0); trace_l.append(lam_used); Ks = K
                    if it_ > 1:
                        break
                    continue
            X, U, J = Xn, Un, Jn
            Ks = K
            trace_c.append(J); trace_a.append(a); trace_l.append(lam_used)
            lam = max(lam / 2.0, 1e-6)
            if abs(J - Jprev) < tol or J > 1e6:
                break
        print("solve golden seed", seed, "iterations", it, "costs", trace_c, "alphas", trace_a)
        out["x0_%d" % seed] = x0; out["u_init_%d" % seed] = Un * 0 + 0 if False else None
        out["x0_%d" % seed] = x0
        out["iters_%d" % seed] = it
        out["trace_cost_%d" % seed] = np.array(trace_c); out["trace_alpha_%d" % seed] = np.array(trace_a); out["trace_lambda_%d" % seed] = np.array(trace_l)
        out["K_%d" % seed] = Ks; out["xbar_%d" % seed] = X; out["ubar_%d" % seed] = U
    out = {k_: v for k_, v in out.items() if v is not None}
    out["u_init"] = np.array(uin)
    out.update(N=N, h=h, gravity=grav, Q=Q, R=R, Qf=Qf, stance=stance, x_ref=x_ref, u_ref=u_ref, com_ref=com_ref, ee_ref=ee_ref,
               task_weights=np.array([w["com"], w["comvel"], w["eepos"], w["eevel"], w["upright"], w["balance"]]), w_joint=w["joint"], w_ctrl=w["ctrl"],
               max_iter=max_iter, tol=tol)
    if contact:
        out.update(contact=contact, soft=1e-5)
    np.savez(os.path.join(HERE, out_name), **out)
    print("solve golden written:", out_name)


def gen_solve_round3():
    """Round 3: (i) the full N = 25 horizon with EXACT Jacobians of the Kane step (gains then pin far below the 1e-4 the
    central differences of the 6-knot golden allowed); (ii) the contact row: gravity -9.81, unilateral stance constraints as a
    NumPy KKT system, complex-step Jacobians of the constrained step, gravity-compensating initial controls."""
    gen_solve(N=25, out_name="solve_golden_n25.npz", n_seeds=2, max_iter=4, jac="tangent", rng_seed=77, swing=(8, 14))
    bodies, ctrl = load_mjcf()
    xs = np.zeros(51); xs[2] = 1.0432; xs[3] = 1.0
    g981 = np.array([0.0, 0.0, -9.81])
    ug = kane_eval_c(bodies, xs[:26], xs[26:], np.zeros(25), g981, 0.1)[0][6:]          # qfrc_bias of the hinges at rest
    gen_solve(N=6, out_name="solve_golden_contact.npz", n_seeds=1, max_iter=3, jac="complex", contact=2, grav=(0.0, 0.0, -9.81), rng_seed=99,
              swing=None, u_scale=1.0, u_bias=ug)


def gen_friction():
    """friction_golden.npz: single steps of the stance row in contact mode 3 (unilateral + Coulomb release) -- states where the cone
    is inactive (the step equals mode 2), where one foot slips and where both do, with the margin by which each decision is taken
    (the oracle / the kernels must take the same branch: cases within 5 % of the cone's surface are rejected here)."""
    bodies, ctrl = load_mjcf()
    rng = np.random.default_rng(2026)
    h, grav = 0.02, np.array([0.0, 0.0, -9.81])
    out = dict(x=[], u=[], stance=[], mu=[], x_next=[], x_next_mode2=[], x_next_mode4=[], slide=[], act=[], lam=[], lam_mode4=[])
    tries = 0
    want = {(0, 0): 3, (1, 0): 2, (0, 1): 2, (1, 1): 3}
    while any(v > 0 for v in want.values()) and tries < 2000:
        tries += 1
        x = np.zeros(51); x[2] = 1.0432; x[3] = 1.0
        x[0:3] += rng.uniform(-0.02, 0.02, 3)
        aa = rng.uniform(-0.05, 0.05, 3); ang = np.linalg.norm(aa); x[3] = np.cos(ang / 2); x[4:7] = np.sin(ang / 2) / ang * aa
        x[7:26] = rng.uniform(-0.15, 0.15, 19)
        x[26:] = rng.uniform(-0.4, 0.4, 25)
        x[26:29] += rng.uniform(-0.6, 0.6, 3) * (tries % 2)            # a lateral push on every second draw
        u = rng.uniform(-20, 20, 19)
        mu = [1.0, 0.6, 0.3, 0.1][tries % 4]
        stance = (1, 1)
        xn, qacc, Mh, lam, act, slide = kane_step_c(bodies, ctrl, x, u, h, grav, stance=stance, contact=3, mu=mu, want=True)
        xn2, _, _, lam2, act2, _ = kane_step_c(bodies, ctrl, x, u, h, grav, stance=stance, contact=2, want=True)
        xn4, _, _, lam4, _, slide4 = kane_step_c(bodies, ctrl, x, u, h, grav, stance=stance, contact=4, mu=mu, want=True)
        assert list(slide4) == list(slide)
        # margins of the decisions on the mode-2 solution (what the Coulomb check looks at)
        _, feet0 = kane_eval_c(bodies, x[:26], x[26:], np.zeros(25), grav, 0.1)
        ok = True
        for f in range(2):
            if not act2[f]:
                ok = False          # keep to states where both feet push (the unilateral branch has its own goldens)
                continue
            fo = lam2[6 * f + 3:6 * f + 6]; fn = feet0[f]["up"] @ fo; ft = np.sqrt(max(fo @ fo - fn * fn, 0.0))
            if abs(ft / (mu * fn) - 1.0) < 0.05 or fn < 1.0:
                ok = False
        key = (int(slide[0]), int(slide[1]))
        if not ok or want.get(key, 0) <= 0:
            continue
        want[key] -= 1
        for k, val_ in (("x", x), ("u", u), ("stance", np.array(stance)), ("mu", mu), ("x_next", xn), ("x_next_mode2", xn2), ("x_next_mode4", xn4), ("lam_mode4", lam4), ("slide", np.array(key)), ("act", np.array(act, dtype=int)), ("lam", lam)):
            out[k].append(val_)
    assert all(v == 0 for v in want.values()), want
    np.savez(os.path.join(HERE, "friction_golden.npz"), h=h, gravity=grav, soft=1e-5, **{k: np.array(v) for k, v in out.items()})
    sl = np.array(out["slide"])
    print("friction golden:", len(out["x"]), "cases; sliding patterns", sl.tolist(), "max |x_next - x_next_mode2|", [float(np.abs(a - b).max()) for a, b in zip(out["x_next"], out["x_next_mode2"])])


def gen_limits():
    """joint_limit_golden.npz: single steps with joint-limit rows (kane_step_lim): hinges past their range moving outward (stopped), moving
    back in (left alone), several at once, on the constraint-free plant and with unilateral stance; decisions with |v_i+| < 0.05 rad/s
    are rejected (the oracle / the kernels must take the same branch)."""
    bodies, ctrl = load_mjcf()
    rngj = np.array([b["rng"] for b in bodies if b["rng"] is not None])
    rng = np.random.default_rng(2027)
    h, grav = 0.02, np.array([0.0, 0.0, -9.81])
    out = dict(x=[], u=[], stance=[], contact=[], x_next=[], x_next_unlimited=[], lock=[])
    want = {(0, 0): 2, (0, 1): 3, (0, 2): 3, (2, 0): 1, (2, 1): 3, (2, 2): 2}       # (contact mode, min(#locked, 2))
    tries = 0
    while any(v > 0 for v in want.values()) and tries < 4000:
        tries += 1
        contact = [0, 2][tries % 2]
        x = np.zeros(51); x[2] = 1.0432; x[3] = 1.0
        aa = rng.uniform(-0.05, 0.05, 3); ang = np.linalg.norm(aa); x[3] = np.cos(ang / 2); x[4:7] = np.sin(ang / 2) / ang * aa
        x[7:26] = rng.uniform(-0.15, 0.15, 19)
        x[26:] = rng.uniform(-0.4, 0.4, 25)
        nv = int(rng.integers(0, 4))
        for i in rng.choice(19, size=nv, replace=False):
            up = rng.random() < 0.5
            x[7 + i] = rngj[i, 1] + rng.uniform(0.01, 0.08) if up else rngj[i, 0] - rng.uniform(0.01, 0.08)
            x[32 + i] = rng.uniform(0.3, 2.0) * (1 if up else -1) * (1 if rng.random() < 0.75 else -1)     # mostly outward
        u = rng.uniform(-20, 20, 19)
        stance = (1, 1) if tries % 3 else (1, 0)
        xn, lock, xn0, margin = kane_step_lim(bodies, ctrl, x, u, h, grav, stance=stance, contact=contact)
        key = (contact, min(len(lock), 2))
        if margin < 0.05 or want.get(key, 0) <= 0 or not np.all(np.isfinite(xn)):
            continue
        want[key] -= 1
        lk = np.zeros(19, dtype=int); lk[lock] = 1
        for k, val_ in (("x", x), ("u", u), ("stance", np.array(stance)), ("contact", contact), ("x_next", xn), ("x_next_unlimited", xn0), ("lock", lk)):
            out[k].append(val_)
    assert all(v == 0 for v in want.values()), want
    np.savez(os.path.join(HERE, "joint_limit_golden.npz"), h=h, gravity=grav, soft=1e-5, jrange=rngj, **{k: np.array(v) for k, v in out.items()})
    print("joint-limit golden:", len(out["x"]), "cases; locked hinges", [np.flatnonzero(l).tolist() for l in out["lock"]],
          "max |x_next - unlimited|", [float(np.abs(a - b).max()) for a, b in zip(out["x_next"], out["x_next_unlimited"])])


def gen_limits_stiff():
    """joint_limit_stiffness_golden.npz: the joint-limit rows with the restoring stiffness k = 1 / (2 h)^2 = 625 (kane_step_lim): hinges past
    their range moving outward, standing still, and moving back in too slowly (all constrained: pushed back at v+ = -h k r) or fast enough
    (left alone), on the constraint-free plant and with unilateral stance; decisions closer than 0.05 rad/s to the switch are rejected."""
    bodies, ctrl = load_mjcf()
    rngj = np.array([b["rng"] for b in bodies if b["rng"] is not None])
    rng = np.random.default_rng(2031)
    h, grav, k = 0.02, np.array([0.0, 0.0, -9.81]), 625.0
    out = dict(x=[], u=[], stance=[], contact=[], x_next=[], x_next_unlimited=[], lock=[])
    want = {(0, 0): 1, (0, 1): 3, (0, 2): 3, (2, 0): 1, (2, 1): 2, (2, 2): 2}
    slow_in = 0
    tries = 0
    while any(v > 0 for v in want.values()) and tries < 4000:
        tries += 1
        contact = [0, 2][tries % 2]
        x = np.zeros(51); x[2] = 1.0432; x[3] = 1.0
        aa = rng.uniform(-0.05, 0.05, 3); ang = np.linalg.norm(aa); x[3] = np.cos(ang / 2); x[4:7] = np.sin(ang / 2) / ang * aa
        x[7:26] = rng.uniform(-0.15, 0.15, 19)
        x[26:] = rng.uniform(-0.4, 0.4, 25)
        nv = int(rng.integers(0, 4))
        inward = []
        for i in rng.choice(19, size=nv, replace=False):
            up = rng.random() < 0.5
            x[7 + i] = rngj[i, 1] + rng.uniform(0.01, 0.08) if up else rngj[i, 0] - rng.uniform(0.01, 0.08)
            sgn = 1 if rng.random() < 0.5 else -1          # half of them already move back in: slowly (still constrained) or fast (left alone)
            x[32 + i] = rng.uniform(0.05, 2.5) * (1 if up else -1) * sgn
            if sgn < 0:
                inward.append(int(i))
        u = rng.uniform(-20, 20, 19)
        stance = (1, 1) if tries % 3 else (1, 0)
        xn, lock, xn0, margin = kane_step_lim(bodies, ctrl, x, u, h, grav, stiffness=k, stance=stance, contact=contact)
        key = (contact, min(len(lock), 2))
        if margin < 0.05 or want.get(key, 0) <= 0 or not np.all(np.isfinite(xn)):
            continue
        want[key] -= 1
        slow_in += len([i for i in inward if i in lock])
        lk = np.zeros(19, dtype=int); lk[lock] = 1
        for kk, val_ in (("x", x), ("u", u), ("stance", np.array(stance)), ("contact", contact), ("x_next", xn), ("x_next_unlimited", xn0), ("lock", lk)):
            out[kk].append(val_)
    assert all(v == 0 for v in want.values()), want
    assert slow_in >= 1, "no case with a hinge that moves back in too slowly"
    # what the row does: v_i+ = -h k r_i on every constrained hinge
    for x, xn, lk in zip(out["x"], out["x_next"], out["lock"]):
        for i in np.flatnonzero(lk):
            r = x[7 + i] - rngj[i, 1] if x[7 + i] > rngj[i, 1] else x[7 + i] - rngj[i, 0]
            assert abs(xn[32 + i] + h * k * r) < 1e-9, (i, xn[32 + i], -h * k * r)
    np.savez(os.path.join(HERE, "joint_limit_stiffness_golden.npz"), h=h, gravity=grav, soft=1e-5, stiffness=k, jrange=rngj, **{kk: np.array(v) for kk, v in out.items()})
    print("joint-limit stiffness golden:", len(out["x"]), "cases; constrained hinges", [np.flatnonzero(l).tolist() for l in out["lock"]], "of which moving back in:", slow_in)


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "limits":
        gen_limits()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "limits_stiff":
        gen_limits_stiff()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "friction":
        gen_friction()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "refdata":
        gen_refdata()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "solve":
        gen_solve()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "solve3":
        gen_solve_round3()
        sys.exit(0)
    gen_dynamics()
    gen_costs()
    gen_riccati()
    gen_refdata()
    gen_solve()
