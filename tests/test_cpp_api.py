"""The C++ mirror of the reference's classes (include/ilqr_hip.hpp) compiles against the C ABI, fails loudly without a
GPU, and -- on a GPU -- reproduces the Python wrapper call for call (two MPC steps: cold start, warm start)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import conftest
pkg = conftest.load_package()
sc = pkg.scenario
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "mpc-ilqr-mujoco_amd", "lib")


def _build(tmp_path):
    exe = str(tmp_path / "cpp_api_demo")
    cmd = ["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "cpp_api_demo.cpp"),
           "-L", LIBDIR, "-lilqr_hip", "-Wl,-rpath," + LIBDIR, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def _inputs(path, prob, x0):
    N = prob["N"]
    parts = [[N, prob["dt"]], prob["Q"], prob["R"], prob["Qf"], prob["task_weights"], [prob["w_joint"], prob["w_ctrl"]], prob["gravity"], x0,
             prob["x_ref"][0].ravel(), prob["com_ref"][0].ravel(), prob["ee_ref"][0].ravel(), prob["stance"][0].ravel().astype(np.float64)]
    np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in parts]).tofile(path)


def test_cpp_mirror_builds_and_fails_loudly_without_gpu(tmp_path):
    from mpc_ilqr_mujoco_amd import solver as sv
    sv.load_library()                      # the library must exist (built by __graft_entry__.build())
    exe = _build(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    prob = sc.make_problem(sv.reference_kinematics)
    _inputs(str(tmp_path / "in.bin"), prob, sc.standing_state())
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "ilqr_hip_create" in r.stderr and not os.path.exists(str(tmp_path / "out.bin"))


@pytest.mark.gpu
def test_cpp_mirror_matches_python_wrapper(tmp_path):
    from mpc_ilqr_mujoco_amd import solver as sv
    exe = _build(tmp_path)
    stance = np.ones((26, 2), dtype=np.int32); stance[4:9, 0] = 0
    prob = sc.make_problem(sv.reference_kinematics, stance=stance, gravity=(0.0, 0.0, -2.0))
    x0 = sc.synthetic_batch(1, 25, 3, np.zeros(19))[0][0]
    _inputs(str(tmp_path / "in.bin"), prob, x0)
    logdir = tmp_path / "cpp_logs"; logdir.mkdir()
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(logdir)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr)
    # profiling table in the reference's layout (main/humanoid_mpc.cpp:195-226) with the MPC_* and iLQR_* keys
    assert "=== Performance Profiling ===" in r.stdout and "Function" in r.stdout and "Avg(ms)" in r.stdout
    for key in ("MPC_stepOnce", "MPC_extractReference", "MPC_warmStart", "MPC_iLQR_solve", "MPC_computeControl",
                "iLQR_forwardRollout", "iLQR_linearization", "iLQR_costQuadratics", "iLQR_backwardPass", "iLQR_lineSearch"):
        line = [l for l in r.stdout.splitlines() if l.startswith(key)]
        assert len(line) == 1 and int(line[0].split()[1]) == 2, key
    out = np.fromfile(str(tmp_path / "out.bin")).reshape(2, 2 + 19 + 19 + 51)
    s = sv.BatchedILQR(1); s.set_problem(prob); s.set_max_iterations(3)
    mpc = sv.BatchedMPC(s, lambda t: (prob["x_ref"], prob["u_ref"], prob["com_ref"]))
    from mpc_ilqr_mujoco_amd import mpc_loop as ml
    pylogs = ml.MPCLogs(str(tmp_path / "py_logs"), prob["dt"])
    x = x0[None].copy()
    for step in range(2):
        u = mpc.step_once(x)
        assert out[step, 0] == 1.0 and out[step, 1] == mpc.last_solve_cost[0]
        assert np.array_equal(out[step, 2:21], u[0]) and np.array_equal(out[step, 21:40], s.ubar()[0, 0])
        assert np.array_equal(out[step, 40:], s.gains_K()[0, 0, 0])
        pylogs.log(mpc.t_idx, mpc.last_solve_cost[0], 0.0, x[0], u[0], prob["x_ref"][0, 0], prob["u_ref"][0, 0], s.xbar()[0, 0], s.ubar()[0, 0])
        x = s.step(x, u)
    pylogs.close()
    s.close()
    # the three CSV files of the C++ mirror equal the Python runner's, text for text (solve_time_ms aside)
    for name in ("q_optimal.csv", "u_optimal.csv"):
        assert open(str(logdir / name)).read() == open(str(tmp_path / "py_logs" / name)).read(), name
    a = open(str(logdir / "mpc_log.csv")).read().splitlines(); b = open(str(tmp_path / "py_logs" / "mpc_log.csv")).read().splitlines()
    assert a[0] == b[0] and len(a) == len(b) == 3
    for la, lb in zip(a[1:], b[1:]):
        fa, fb = la.split(","), lb.split(",")
        assert fa[:3] == fb[:3] and fa[4:] == fb[4:] and float(fa[3]) > 0.0


def test_exact_signature_eigen_adapter_compiles_against_a_robotutils_shaped_class():
    """include/ilqr_hip_eigen.hpp (iLQR(RobotUtils&, int, double, const std::string&), solve(const Eigen::VectorXd&, ...),
    MPC::stepOnce(const Eigen::VectorXd&, Eigen::VectorXd&): reference include/ilqr/ilqr.hpp:19-45, mpc.hpp:20-47) is built by
    __graft_entry__.build() against the stand-in Eigen of tests/cpp/fake_eigen; without a GPU the demo fails loudly."""
    import __graft_entry__ as ge
    exe = ge.build_cpp_demos()["cpp_eigen_drop_in_demo"]
    assert os.path.exists(exe)
    src = open(os.path.join(ROOT, "include", "ilqr_hip_eigen.hpp")).read()
    for sig in ("iLQR(Robot& robot, int N, double dt, const std::string&", "bool solve(const Eigen::VectorXd& x0, const std::vector<Eigen::VectorXd>& x_ref",
                "bool stepOnce(const Eigen::VectorXd& x_measured, Eigen::VectorXd& u_apply)", "MPC(Robot& robot, int N, double dt, const std::string&"):
        assert sig in src, sig


@pytest.mark.gpu
def test_exact_signature_eigen_adapter_equals_the_std_vector_mirror_bit_for_bit(tmp_path):
    """The same two MPC steps through ilqr_hip_eigen::MPC<RobotUtils> (problem data pulled out of the robot's getters) and through
    ilqr_hip::MPC (data pushed by the caller): identical costs, controls, nominal controls and gain rows; the adapter's iLQR on its
    own reproduces the MPC's first step (checked inside the demo)."""
    import __graft_entry__ as ge
    from mpc_ilqr_mujoco_amd import solver as sv
    exes = ge.build_cpp_demos()
    stance = np.ones((26, 2), dtype=np.int32); stance[4:9, 0] = 0
    prob = sc.make_problem(sv.reference_kinematics, stance=stance, gravity=(0.0, 0.0, -2.0))
    x0 = sc.synthetic_batch(1, 25, 3, np.zeros(19))[0][0]
    _inputs(str(tmp_path / "in.bin"), prob, x0)
    outs = {}
    for name in ("cpp_api_demo", "cpp_eigen_drop_in_demo"):
        out = str(tmp_path / (name + ".bin"))
        r = subprocess.run([exes[name], str(tmp_path / "in.bin"), out], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (name, r.returncode, r.stderr)
        outs[name] = np.fromfile(out)
    assert outs["cpp_api_demo"].size == 2 * (2 + 19 + 19 + 51) and np.array_equal(outs["cpp_api_demo"], outs["cpp_eigen_drop_in_demo"])


@pytest.mark.gpu
def test_config0_as_shipped_walking_mpc_through_the_cpp_mirror_matches_oracle(tmp_path):
    """BASELINE.json configs[0] as config.yaml ships it: walking references q_ref2_mj / v_ref2 /
    contact_walking, gravity [0, 0, -1], N = 25, the reference's solver defaults (10 iterations, tolerance 1e-4, lambda 1e-6),
    six receding-horizon steps of MPC::stepOnce through the C++ mirror (batch 1), each against the oracle's warm-started solve
    from the same measured state: cost, applied control, first nominal control, first gain row."""
    from mpc_ilqr_mujoco_amd import references as rf
    from mpc_ilqr_mujoco_amd import solver as sv
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    exe = str(tmp_path / "cpp_mpc_walk_demo")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "cpp_mpc_walk_demo.cpp"),
                        "-L", LIBDIR, "-lilqr_hip", "-Wl,-rpath," + LIBDIR, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    g = np.load(os.path.join(ROOT, "tests", "golden", "refdata_golden.npz"))
    rd = rf.ReferenceData(sv.reference_kinematics, sv.reference_com_velocity)
    rd.set_states(np.concatenate([g["q_ref2_mj"], g["v_ref2"]], axis=1))
    rd.contact = g["contact_walking"].astype(np.int32)
    N, steps = 25, 6
    base = sc.make_problem(sv.reference_kinematics, N=N, gravity=(0.0, 0.0, -1.0))       # shipped weights (W_com_vel = 0)
    probs = [rd.problem_at(t, N, base) for t in range(steps)]
    assert probs[0]["stance"].min() == 0 and np.array_equal(probs[0]["stance"], probs[3]["stance"])     # horizon-local contact index
    assert not np.array_equal(probs[0]["x_ref"], probs[3]["x_ref"])                                       # absolute reference window
    x0 = rd.x_ref[0].copy(); x0[7:26] += np.random.default_rng(5).uniform(-0.01, 0.01, 19)
    p0 = probs[0]
    parts = [[N, p0["dt"], steps], p0["Q"], p0["R"], p0["Qf"], p0["task_weights"], [p0["w_joint"], p0["w_ctrl"]], p0["gravity"], x0,
             p0["ee_ref"][0].ravel(), p0["com_vel_ref"][0].ravel(), p0["stance"][0].ravel().astype(np.float64)]
    for pr in probs:
        parts += [pr["x_ref"][0].ravel(), pr["com_ref"][0].ravel()]
    np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in parts]).tofile(str(tmp_path / "in.bin"))
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr)
    out = np.fromfile(str(tmp_path / "out.bin")).reshape(steps, 2 + 51 + 19 + 19 + 51 + (N + 1) * 51 + N * 19)
    o = ol.Oracle(N, p0["dt"])
    o.set_options(max_iter=10)                     # once: lambda survives a solve, in the reference (ilqr.hpp:54) as in both restatements
    prev = None
    for step in range(steps):
        ok, cost, xm = out[step, 0], out[step, 1], out[step, 2:53]
        u_apply, ub0, k0 = out[step, 53:72], out[step, 72:91], out[step, 91:142]
        xbar_gpu = out[step, 142:142 + (N + 1) * 51].reshape(N + 1, 51); ubar_gpu = out[step, 142 + (N + 1) * 51:].reshape(N, 19)
        assert ok == 1.0 and np.isfinite(cost)
        o.set_problem(probs[step])
        if prev is None:
            o.initialize(xm, None)                 # cold start: gravity compensation at the measured state + N rollouts
        else:
            o.initialize(xm, None, prev[0], prev[1])
        okc, c = o.solve(xm)
        oub = o.get("ubar")
        # every step is compared from IDENTICAL inputs: the oracle warm-starts from the trajectory the mirror itself carried over
        # (with a 1e-4 exit tolerance two independent chains may part ways at a borderline exit; that is not what is tested here)
        prev = (xbar_gpu, ubar_gpu)
        assert okc and abs(c - cost) <= 1e-5 * abs(c), (step, c, cost)
        assert np.abs(ub0 - oub[0]).max() <= 1e-5 * max(1.0, np.abs(oub[0]).max()), step
        assert np.abs(ubar_gpu - oub).max() <= 1e-5 * max(1.0, np.abs(oub).max()) and np.array_equal(ubar_gpu[0], ub0)
        assert np.array_equal(u_apply, ub0)        # xbar[0] == x_measured after the solve: u = ubar[0] (Appendix D #11)
        K0 = o.get("K")[0]
        assert np.abs(k0 - K0[0]).max() <= 1e-5 * np.abs(K0).max(), step
    if steps > 1:
        assert not np.array_equal(out[0, 2:53], out[1, 2:53])
