#!/usr/bin/env python3
"""Generate the Unitree H1 model-constant headers from the reference's robot description.

Reads (data only, no code):
  /root/reference/robots/h1_description/mjcf/h1.xml   -> dynamics truth (MuJoCo inertials, ranges)
  /root/reference/robots/h1_description/urdf/h1.urdf  -> Pinocchio-side inertials (CoM cost terms)

Writes the SAME constant table to
  oracle/h1_model_data.h                       (oracle copy)
  mpc-ilqr-mujoco_amd/csrc/h1_model_data.h     (product copy)
so the oracle never includes product code and vice versa.  Run only in the build container
(the GPU box has no /root/reference); the generated headers are committed.
"""
import math
import os
import re
import sys
import xml.etree.ElementTree as ET

import numpy as np

REF = os.environ.get("ILQR_REFERENCE_ROOT", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def quat_to_R(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ])


def rpy_to_R(rpy):
    r, p, y = rpy
    Rx = np.array([[1, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]])
    Ry = np.array([[math.cos(p), 0, math.sin(p)], [0, 1, 0], [-math.sin(p), 0, math.cos(p)]])
    Rz = np.array([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def fl(s):
    return np.array([float(t) for t in s.split()])


def parse_mjcf(path):
    root = ET.parse(path).getroot()
    wb = root.find("worldbody")
    bodies = []

    def rec(elem, parent):
        for b in elem.findall("body"):
            idx = len(bodies)
            inert = b.find("inertial")
            j = b.find("joint")
            fj = b.find("freejoint")
            d = dict(
                name=b.get("name"), parent=parent,
                pos=fl(b.get("pos", "0 0 0")),
                quat=fl(b.get("quat", "1 0 0 0")),
                mass=float(inert.get("mass")),
                ipos=fl(inert.get("pos")),
                iquat=fl(inert.get("quat", "1 0 0 0")),
                diag=fl(inert.get("diaginertia")),
                free=fj is not None,
            )
            if j is not None:
                d["axis"] = fl(j.get("axis"))
                d["range"] = fl(j.get("range"))
                d["jname"] = j.get("name")
            bodies.append(d)
            rec(b, idx)

    rec(wb, -1)
    dflt = root.find("default").find("default").find("joint")
    damping = float(dflt.get("damping"))
    armature = float(dflt.get("armature"))
    ctrl = {}
    for m in root.find("actuator").findall("motor"):
        ctrl[m.get("joint")] = fl(m.get("ctrlrange"))
    return bodies, damping, armature, ctrl


def parse_urdf(path):
    root = ET.parse(path).getroot()
    links = {}
    for l in root.findall("link"):
        inert = l.find("inertial")
        if inert is None:
            continue
        o = inert.find("origin")
        links[l.get("name")] = dict(
            mass=float(inert.find("mass").get("value")),
            com=fl(o.get("xyz")),
            rpy=fl(o.get("rpy", "0 0 0")),
        )
    joints = {}
    for j in root.findall("joint"):
        if j.get("type") != "revolute":
            continue
        o = j.find("origin")
        joints[j.find("child").get("link")] = dict(
            xyz=fl(o.get("xyz")), rpy=fl(o.get("rpy", "0 0 0")),
            axis=fl(j.find("axis").get("xyz")),
            parent=j.find("parent").get("link"),
            lower=float(j.find("limit").get("lower")), upper=float(j.find("limit").get("upper")),
        )
    return links, joints


def carr(name, a, fmt="%.17g"):
    a = np.asarray(a)
    flat = ", ".join(fmt % v for v in a.reshape(-1))
    dims = "".join("[%d]" % d for d in a.shape)
    return "H1_CONST double %s%s = {%s};\n" % (name, dims, flat)


def iarr(name, a):
    a = np.asarray(a)
    flat = ", ".join("%d" % v for v in a.reshape(-1))
    dims = "".join("[%d]" % d for d in a.shape)
    return "H1_CONST int %s%s = {%s};\n" % (name, dims, flat)


def main():
    bodies, damping, armature, ctrl = parse_mjcf(os.path.join(REF, "robots/h1_description/mjcf/h1.xml"))
    links, ujoints = parse_urdf(os.path.join(REF, "robots/h1_description/urdf/h1.urdf"))
    nb = len(bodies)
    assert nb == 20, nb
    assert bodies[0]["free"]

    parent = [b["parent"] for b in bodies]
    pos = np.array([b["pos"] for b in bodies])
    pos[0] = 0.0  # free joint: body pos is overridden by qpos
    rfix = np.array([quat_to_R(b["quat"]) for b in bodies])
    mass = np.array([b["mass"] for b in bodies])
    com = np.array([b["ipos"] for b in bodies])
    inertia = np.array([quat_to_R(b["iquat"]) @ np.diag(b["diag"]) @ quat_to_R(b["iquat"]).T for b in bodies])
    axis = [-1]
    for b in bodies[1:]:
        a = b["axis"]
        k = int(np.argmax(np.abs(a)))
        assert abs(a[k] - 1.0) < 1e-12 and abs(np.abs(a).sum() - 1.0) < 1e-12, a
        axis.append(k)
    jrange = np.array([b["range"] for b in bodies[1:]])
    ctrlrange = np.array([ctrl[b["jname"]] for b in bodies[1:]])

    # URDF side (Pinocchio model): same tree; joint origins from xyz/rpy
    u_pos = np.zeros((nb, 3))
    u_rfix = np.zeros((nb, 3, 3))
    u_rfix[0] = np.eye(3)
    u_mass = np.zeros(nb)
    u_com = np.zeros((nb, 3))
    for i, b in enumerate(bodies):
        ln = links[b["name"]]
        assert np.allclose(ln["rpy"], 0)
        u_mass[i] = ln["mass"]
        u_com[i] = ln["com"]
        if i > 0:
            uj = ujoints[b["name"]]
            assert uj["parent"] == bodies[parent[i]]["name"]
            u_pos[i] = uj["xyz"]
            u_rfix[i] = rpy_to_R(uj["rpy"])
            k = int(np.argmax(np.abs(uj["axis"])))
            assert k == axis[i] and abs(uj["axis"][k] - 1.0) < 1e-12
            assert np.allclose(u_pos[i], pos[i], atol=1e-12), (b["name"], u_pos[i], pos[i])
            assert np.allclose(u_rfix[i], rfix[i], atol=1e-5)
            assert abs(uj["lower"] - jrange[i - 1][0]) < 1e-9 and abs(uj["upper"] - jrange[i - 1][1]) < 1e-9

    # ancestor-or-self table over hinge joints (index 0..18 = body 1..19)
    anc = np.zeros((19, 19), dtype=int)  # anc[k][j] = 1 if joint k is ancestor-or-self of joint j
    for j in range(1, nb):
        a = j
        while a > 0:
            anc[a - 1][j - 1] = 1
            a = parent[a]
    depth = [0] * nb
    for i in range(1, nb):
        depth[i] = depth[parent[i]] + 1

    out = []
    out.append("// GENERATED by tools/gen_h1_model.py from the reference's robot description -- do not edit.\n")
    out.append("// Source data: robots/h1_description/mjcf/h1.xml:46-209 (MuJoCo inertials, joint ranges,\n")
    out.append("// ctrlrange, default damping/armature h1.xml:7) and robots/h1_description/urdf/h1.urdf\n")
    out.append("// (link masses / CoM offsets used by the Pinocchio-side cost terms, derivatives.cpp:29).\n")
    out.append("#ifndef H1_MODEL_DATA_H\n#define H1_MODEL_DATA_H\n\n")
    out.append("/* storage qualifier of the tables (a HIP translation unit sets it to __constant__) */\n#ifndef H1_CONST\n#define H1_CONST static const\n#endif\n")
    out.append("#define H1_NB 20   /* bodies: pelvis + 19 hinge links */\n")
    out.append("#define H1_NJ 19   /* hinge joints == actuators */\n")
    out.append("#define H1_NQ 26\n#define H1_NV 25\n#define H1_NX 51\n#define H1_NU 19\n\n")
    out.append("/* body names (index = MuJoCo body id - 1):\n")
    for i, b in enumerate(bodies):
        out.append("   %2d %-28s parent %2d depth %d\n" % (i, b["name"], parent[i], depth[i]))
    out.append("*/\n")
    out.append(iarr("H1_PARENT", parent))
    out.append(iarr("H1_AXIS", axis))
    out.append(iarr("H1_DEPTH", depth))
    out.append(iarr("H1_ANC", anc))
    out.append("#define H1_DAMPING %.17g\n#define H1_ARMATURE %.17g\n" % (damping, armature))
    out.append(carr("H1_POS", pos))
    out.append(carr("H1_RFIX", rfix))
    out.append(carr("H1_MASS", mass))
    out.append(carr("H1_COM", com))
    out.append(carr("H1_INERTIA", inertia))
    out.append(carr("H1_JRANGE", jrange))
    out.append(carr("H1_CTRLRANGE", ctrlrange))
    out.append("/* URDF (Pinocchio) side */\n")
    out.append(carr("H1U_POS", u_pos))
    out.append(carr("H1U_RFIX", u_rfix))
    out.append(carr("H1U_MASS", u_mass))
    out.append(carr("H1U_COM", u_com))
    out.append("#define H1_EE_LEFT 5    /* left_ankle_link */\n#define H1_EE_RIGHT 10  /* right_ankle_link */\n")
    out.append("\n#endif\n")
    text = "".join(out)
    for dst in ("oracle/h1_model_data.h", "mpc-ilqr-mujoco_amd/csrc/h1_model_data.h"):
        with open(os.path.join(ROOT, dst), "w") as f:
            f.write(text)
    # compile-time variant for the fully unrolled (register-resident) dynamics: every table is constexpr so
    # that the unrolled code folds the constants (identity rotations, zero offsets, axis selection)
    ce = text.replace("H1_MODEL_DATA_H", "H1_MODEL_CONSTEXPR_H")
    ce = ce.replace("#ifndef H1_CONST\n#define H1_CONST static const\n#endif\n", "")
    ce = ce.replace("H1_CONST ", "static constexpr ")
    for name in ("H1_NB", "H1_NJ", "H1_NQ", "H1_NV", "H1_NX", "H1_NU", "H1_DAMPING", "H1_ARMATURE", "H1_EE_LEFT", "H1_EE_RIGHT"):
        ce = re.sub(r"#define %s [^\n]*\n" % name, "", ce)
    ce = re.sub(r"\bH1U?_([A-Z]+)\b", lambda m: ("CU_" if m.group(0).startswith("H1U_") else "C_") + m.group(1), ce)
    ce = ce.replace("#ifndef C_MODEL", "#ifndef H1_MODEL_CONSTEXPR_H").replace("#define C_MODEL", "#define H1_MODEL_CONSTEXPR_H")
    with open(os.path.join(ROOT, "mpc-ilqr-mujoco_amd/csrc/h1_model_constexpr.h"), "w") as f:
        f.write("#pragma once\nnamespace h1c {\n" + ce.replace("#ifndef H1_MODEL_CONSTEXPR_H\n#define H1_MODEL_CONSTEXPR_H\n", "").replace("#endif\n", "") + "}  // namespace h1c\n")
    print("total mass mjcf %.6f urdf %.6f" % (mass.sum(), u_mass.sum()))
    print("wrote headers; bodies:", [b["name"] for b in bodies])


if __name__ == "__main__":
    sys.exit(main())
