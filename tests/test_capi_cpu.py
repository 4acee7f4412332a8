"""CPU checks of the boundary: the C-ABI library builds, loads and exports every symbol include/ilqr_hip.h
declares; the GPU-free host helpers agree with the oracle; the product fails loudly without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from conftest import ROOT, load_package

pkg = load_package()
sc = pkg.scenario


def _lib():
    import __graft_entry__ as ge
    ge.build_library()
    from mpc_ilqr_mujoco_amd import solver as sv
    return sv, sv.load_library()


def test_library_exports_every_declared_symbol():
    sv, L = _lib()
    hdr = open(os.path.join(ROOT, "include", "ilqr_hip.h")).read()
    declared = set(re.findall(r"\b(ilqr_hip_[A-Za-z_0-9]+)\s*\(", hdr))
    assert declared == set(sv.EXPORTS), declared ^ set(sv.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_host_helpers_match_oracle_cpu():
    sv, L = _lib()
    rng = np.random.default_rng(0)
    for _ in range(5):
        x = sc.standing_state(); x[7:26] = rng.uniform(-0.5, 0.5, 19); x[3:7] = sc._axis_angle_quat(rng.uniform(-1, 1, 3))
        c1, e1 = sv.reference_kinematics(x); c2, e2 = ol.reference_kinematics(x)
        assert np.abs(c1 - c2).max() < 1e-13 and np.abs(e1 - e2).max() < 1e-13
        o = ol.Oracle(25, 0.02); o.set_problem(sc.make_problem(ol.reference_kinematics, gravity=(0, 0, -9.81)))
        assert np.abs(sv.gravity_compensation(x, (0, 0, -9.81)) - o.grav_comp(x)).max() < 1e-11


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sv, L = _lib()
    with pytest.raises(sv.ILQRError):
        sv.BatchedILQR(4)
    h = C.c_void_p()
    assert L.ilqr_hip_create(C.byref(h), 0, 4, 25, C.c_double(0.02)) == 3  # ILQR_ERR_NO_DEVICE
    assert L.ilqr_hip_create(None, 0, 4, 25, C.c_double(0.02)) == 1        # ILQR_ERR_ARG


def test_cost_matrix_rule_and_scenario():
    Q, R, Qf = sc.build_cost_matrices()
    assert Q[0] == 200 and Q[1] == 50 and Q[2] == 200 and Q[3] == 50 and Q[7] == 50 and Q[26] == 150 and Q[27] == 50 and Q[29] == 75 and Q[50] == 75
    assert np.all(R == 0.001)
    assert Qf[0] == 2000 and Qf[1] == 200 and Qf[2] == 2000 and Qf[28] == 1200 and Qf[10] == 100
    x0, u = sc.synthetic_batch(16, 25, 0, np.zeros(19))
    assert np.allclose(np.linalg.norm(x0[:, 3:7], axis=1), 1.0) and np.all(np.abs(u) <= 0.8 * sc.CTRLRANGE + 1e-12)
    x1, _ = sc.synthetic_batch(16, 25, 0, np.zeros(19))
    assert np.array_equal(x0, x1)


def test_multi_gpu_entry_points_validate_their_arguments():
    """RCCL sits behind the C ABI (ilqr_hip_comm_* / ilqr_hip_gather_first_knot): symbols present, argument validation
    without a GPU, payload layout [u0(19) | cost | K0(19 x 51)]."""
    sv, L = _lib()
    assert L.ilqr_hip_payload_width(0) == 20 and L.ilqr_hip_payload_width(1) == 20 + 19 * 51
    assert L.ilqr_hip_comm_get_unique_id(None) == 1                      # ILQR_ERR_ARG
    assert L.ilqr_hip_comm_init(None, 2, 0, None) == 1
    assert L.ilqr_hip_gather_first_knot(None, 0, 0, None) == 1
    assert L.ilqr_hip_comm_destroy(None) == 1
    assert L.ilqr_hip_comm_world(None) == -1 and L.ilqr_hip_comm_rank(None) == -1
    assert L.ilqr_hip_get_adopt_mismatches(None, None) == 1
    assert L.ilqr_hip_get_iterations_enqueued(None) == -1
    assert L.ilqr_hip_set_profiled_stages(None, 0xFF) == 1


def test_cpp_multi_gpu_demo_builds():
    """The C++ sharding demo (one thread + one handle per GPU, RCCL gather behind the C ABI) builds in build()."""
    import subprocess
    import __graft_entry__ as ge
    exe = ge.build_cpp_demos()["cpp_multi_gpu_demo"]
    assert os.path.exists(exe)
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "1", "2", "0", "/tmp/never_written.bin"], capture_output=True, text=True)
        assert r.returncode == 1 and "ilqr_hip_create" in r.stderr        # fails loudly without a GPU


def test_bench_picks_the_dominant_kernel_group_on_exclusive_equivalent_time():
    """bench.py: the linearisation pair, the cost quadratics and the nominal re-rollout overlap on three streams -- their spans are
    scaled by (longest / sum) before the groups are ranked; the group's stage keys are what the timed region keeps event pairs for."""
    import importlib.util, types
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    args = types.SimpleNamespace(contact=False)
    groups = bench.kernel_groups(args, 4096, 25, 1)
    stage = {"iLQR_backwardPass": 20.0, "iLQR_backwardPass_retry": 15.0, "iLQR_lineSearch": 9.0, "iLQR_lineSearch_retry": 7.0,
             "iLQR_linearization": 37.0, "iLQR_costQuadratics": 19.0, "iLQR_computeCost+forwardRollout": 9.0}
    name, keys = bench.dominant_group(groups, stage)
    assert name == "k_backward_pack" and keys == ["iLQR_backwardPass", "iLQR_backwardPass_retry"]      # 35 ms against 37 * 37 / 65 = 21 ms
    stage["iLQR_backwardPass"] = 2.0; stage["iLQR_backwardPass_retry"] = 1.0
    name, keys = bench.dominant_group(groups, stage)
    assert name == "k_lin_primal_s+k_lin_tangent2" and keys == ["iLQR_linearization"]


def test_bench_names_a_baseline_config_only_when_it_runs_one_and_stamps_the_kernel_sources():
    """bench.py: config.workload says 'BASELINE.json configs[k]' only when (batch, horizon, iterations, contact, world) is that
    config; the PMC traffic record (profiles/traffic_latest.json) is attached only to the build it was collected on -- the run
    signature carries a hash of the kernel sources."""
    import importlib.util, types
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    a = types.SimpleNamespace(contact=False, batch=4096, horizon=25, iters=10)
    g = (0.0, 0.0, -1.0)
    assert bench.workload_label(a, 4096, 25, 10, 1, g).startswith("BASELINE.json configs[2]:")
    assert bench.workload_label(a, 4096, 25, 10, 8, g).startswith("BASELINE.json configs[3]")
    assert bench.workload_label(a, 4096, 25, 10, 2, g).startswith("per-GPU shape of BASELINE.json configs[2]/[3] on 2 GPUs")
    for B, N, it in ((1024, 50, 10), (1024, 25, 10), (4096, 25, 5), (1, 25, 10)):
        assert bench.workload_label(a, B, N, it, 1, g).startswith("custom:")
    assert bench.workload_label(types.SimpleNamespace(contact=True), 4096, 25, 10, 1, (0, 0, -9.81)).startswith("custom (contact row f4")
    sig = bench.run_signature(a, 1)
    assert len(sig["csrc_sha"]) == 16 and sig["csrc_sha"] == bench.csrc_hash() and sig["batch"] == 4096


def test_bench_gpus_n_without_a_launcher_never_times_one_rank_silently():
    """`python bench.py --gpus 2` started plainly (the way the driver starts `--gpus 1`) must either run two ranks -- the parent
    starts torch.distributed.run as a child and relays rank 0's line -- or exit non-zero; it must never print `n_gpus: 1` with
    exit code 0 (a SCALE run that did not wrap the command would have measured one GPU).  Without a GPU both
    forms fail loudly: too few devices for one rank per GPU, and -- with --rehearse-single-gpu, which does start the child
    ranks -- the ranks themselves refuse to run without a device."""
    import subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode != 0 and "needs 2 visible GPUs" in r.stderr
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8", "--steps", "1", "--warmup", "0", "--rehearse-single-gpu",
                            "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode != 0 and "starting" in r.stderr and "torch.distributed.run" in r.stderr
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    # a launcher environment that disagrees with --gpus is refused as well (WORLD_SIZE = 1 with --gpus 2)
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env1, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_operand_layout_index_maps(tmp_path):
    """csrc/riccati_pack.h: slot <-> state bijection, fold partners exactly 32 slots apart, contracted rows in tiles 2 / 3, image indices
    distinct and inside their knot regions (tests/cpp/pack_layout_check.cpp, host compile of the same constexpr functions the kernels use)."""
    exe = str(tmp_path / "pack_layout_check")
    r = subprocess.run(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "pack_layout_check.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "pack layout ok" in r.stdout, r.stdout
