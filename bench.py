#!/usr/bin/env python3
"""Headline benchmark: iLQR iterations/s, Unitree H1 (nx=51, nu=19), N=25, batched standing-balance.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one MPC step of the hot path over the whole batch with inputs resident in HBM:
cold-start initializeWithReference (gravity-comp + noise controls, N rollouts) followed by
iLQR::solve with exactly `--iters` iterations per rollout (early exit disabled, the headline mode of
BASELINE.md section 3) and the per-step gather of {u0, cost} rows to rank 0 through the product's own collective
(ilqr_hip_gather_first_knot: RCCL grouped send/recv behind the C ABI when N > 1, a device copy when N = 1).
Workload: BASELINE.json configs[2] "Batch=4096 full iLQR, N=25, 1xMI355X" (the config the metric
"iLQR iterations/sec" is quoted on); weak scaling: 4096 rollouts per GPU (configs[3] at 8 GPUs).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector == matrix peak (AMD spec; the microarch guide lists no fp64 row)
HBM_PEAK_GBS = 8000.0
# algorithmic work per rollout: Riccati backward pass, minimal-reuse formulation (exact count)
RICCATI_FLOPS_PER_KNOT = 914786.0
# Dynamics step, analytic (A_t, B_t), cost quadratics and one line-search trial: COUNTED on the oracle's restatement by its
# op counter (oracle/opcount.cpp; add / sub / mul / div / sqrt / sin / cos = 1 flop each, nothing fused), at knot 7 of rollout 0
# of the seed-0 standing batch under the shipped weights -- tests/test_oracle_golden.py pins these four numbers to the counter.
STEP_FLOPS = 27167.0                 # one dynamics step (h1_step)
JACOBIAN_FLOPS_PER_KNOT = 203199.0   # tangent scheme: 1 forward dynamics + 48 inverse-dynamics tangents + Mhat + 67 solves + integrator
QUAD_FLOPS_PER_KNOT = 135167.0       # closed-form gradient and Hessian of the six task terms + limits (stance knot)
ALPHA_TRIAL_FLOPS_PER_KNOT = 32366.0 # u = ubar + alpha k + K dx, the step, 1/N of computeTotalCost
FLOP_SOURCE = ("Riccati: exact count of the minimal-reuse formulation (SURVEY.md 8(d)); dynamics step / analytic Jacobians / cost quadratics / line-search "
               "trial: counted by the oracle's op counter (oracle/opcount.cpp), pinned by tests/test_oracle_golden.py")
# whole-iteration budget of SURVEY.md 8(d) at N = 25 (scaled by N / 25 for other horizons): 27.8 MFLOP and 2.43 MB per rollout-iteration
ITER_FLOPS_N25, ITER_BYTES_N25 = 27.8e6, 2.43e6
# the same total from the counted figures: Riccati + N x (nominal step + Jacobians + line-search trial) + (N + 1) x quadratics
ITER_FLOPS_COUNTED_N25 = 25 * (RICCATI_FLOPS_PER_KNOT + STEP_FLOPS + JACOBIAN_FLOPS_PER_KNOT + ALPHA_TRIAL_FLOPS_PER_KNOT) + 26 * QUAD_FLOPS_PER_KNOT
# v_mfma_f64_16x16x4_f64 per knot issued by the one-wave Riccati kernels, 2048 flops each: operand layout (riccati_pack.hip, the default
# with analytic Jacobians), folded / generic on the standard layout (riccati_wave.hip)
MFMA_PER_KNOT = {"packed": 354, "folded": 429, "generic": 569}


JAC_FD = False      # set from --jacobians: forward-difference Jacobians carry no structure, the generic Riccati kernel runs


def riccati_variant():
    bk = os.environ.get("ILQR_BACKWARD", "wave")
    return "generic" if ("generic" in bk or JAC_FD) else ("folded" if "fold" in bk else "packed")


FD_STEPS_PER_KNOT = 71.0             # base step + 51 state columns + 19 control columns (robot_utils.cpp:126-160)


def workload_label(args, B, N, iters, world, gravity):
    """Names a BASELINE.json config only when (batch, horizon, stage, contact, iterations) IS that config; else 'custom: ...'."""
    body = ("batch=%d/GPU full iLQR (rollout+Jacobians+cost quadratics+Riccati+8-alpha line search), H1 standing balance, N=%d, dt=0.02, "
            "%d fixed iterations per rollout, shipped config.yaml weights, gravity %s" % (B, N, iters, list(gravity)))
    if getattr(args, "workload", "default") == "config3" and world == 1:
        return "global batch of BASELINE.json configs[3] (32768 rollouts) on ONE GPU: " + body
    if getattr(args, "workload", "default") == "config4" and world == 1:
        return ("global batch of BASELINE.json configs[4] on ONE GPU: batch=%d windows of the H1 walking reference (data/h1_walking_pin.csv rows), N=%d, dt=0.02, per-rollout references and "
                "contact schedule (contact-scheduled costs), unilateral rigid stance constraints on the scheduled feet (contact mode 2), analytic Jacobians of the constrained step, "
                "%d fixed iterations per rollout, shipped config.yaml weights, gravity %s -- the reference's shipped config.yaml value, NOT physical gravity: the `contact` line of the "
                "default workload (--contact) runs under [0, 0, -9.81] and is not comparable with this one; the reference rows come from the committed fixture tests/golden/refdata_golden.npz "
                "(the reference repository does not travel to the GPU box)" % (B, N, iters, list(gravity)))
    if args.contact:
        return "custom (contact row f4: unilateral rigid stance on the scheduled feet): " + body
    if B == 4096 and N == 25 and iters == 10:
        if world == 1:
            return "BASELINE.json configs[2]: " + body
        if world == 8:
            return "BASELINE.json configs[3] (32768 rollouts = 4096 per GPU x 8): " + body
        return "per-GPU shape of BASELINE.json configs[2]/[3] on %d GPUs (configs[3] is the 8-GPU point of this weak-scaling series): " % world + body
    return "custom: " + body


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096, help="rollouts per GPU")
    ap.add_argument("--horizon", type=int, default=25)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--print-signature", action="store_true", help="print the run signature tools/pmc_summary.py --stamp expects, and exit")
    ap.add_argument("--rehearse-single-gpu", action="store_true",
                    help="rehearsal of the N > 1 path on a one-GPU box: every rank uses device 0, gloo instead of RCCL, payload staged through the host")
    ap.add_argument("--try-rccl", action="store_true",
                    help="with --rehearse-single-gpu: attempt the RCCL communicator anyway (two ranks on one device: RCCL refuses, which exercises the all-ranks fallback)")
    ap.add_argument("--gather-gains", action="store_true", help="include K0[19x51] in the per-step gather payload")
    ap.add_argument("--stage", choices=["full", "rollout_jacobians"], default="full",
                    help="full (headline) or BASELINE.json configs[1]: forward rollout + Jacobians only (use with --batch 1024)")
    ap.add_argument("--contact", action="store_true",
                    help="not the headline: contact row f4 (unilateral rigid stance on the scheduled feet, physical gravity -9.81, two-lane kernels, analytic Jacobians of the constrained step)")
    ap.add_argument("--no-contact-line", action="store_true", help="skip the short contact-mode measurement added to the default line")
    ap.add_argument("--jacobians", choices=["analytic", "fd"], default="analytic",
                    help="fd: the reference's own scheme, forward differences with eps 1e-5 (robot_utils.cpp:120-160: 71 steps per knot) -- the like-for-like line "
                         "against cpu_baseline, which times the oracle with the same scheme; the default line carries a short measurement of it as its `fd` object")
    ap.add_argument("--no-fd-line", action="store_true", help="skip the short forward-difference measurement added to the default line")
    ap.add_argument("--workload", choices=["default", "config3", "config4"], default="default",
                    help="default: --batch / --horizon as given (BASELINE configs[2] by default).  config3 / config4: the GLOBAL batch of BASELINE.json "
                         "configs[3] (32768 standing rollouts, N = 25) / configs[4] (8192 windows of the H1 walking reference, N = 50, contact-scheduled "
                         "costs, unilateral stance constraints) on ONE GPU -- the N = 1 anchors of the 8-GPU scaling series")
    return ap.parse_args()


def cpu_baseline(pkg, prob, x0, ui, iters, budget_s, contact=False):
    """Oracle (CPU restatement, kind 'port') timed on the host cores on a bounded sample of the same batch."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    _, flags = ol.lib_native()          # the timed copy: -O3 -march=native, built on this host (oracle/Makefile); the checker build stays portable
    o = ol.Oracle(prob["N"], prob["dt"], native=True)
    o.set_problem(prob)
    if contact:      # same plant and the same kind of Jacobians as the GPU's contact mode: exact derivatives of the constrained step
        o.set_contact_mode(2)
        o.set_options(max_iter=iters, early_exit=0, jac_mode=0)     # (the oracle takes them by forward-mode AD)
    else:            # Jacobians by forward differences, eps 1e-5: the reference's own scheme (robot_utils.cpp:120-160) and the
        o.set_options(max_iter=iters, early_exit=0, jac_mode=1, fd_eps=1e-5)   # faster of the oracle's two modes on a CPU
    cores = ol.max_threads()
    n0 = max(cores, 1)
    t0 = time.perf_counter()
    tot0, *_ = o.batch_solve(x0[:n0], ui[:n0], nthreads=cores)
    dt0 = time.perf_counter() - t0
    rate0 = tot0 / dt0
    n1 = int(min(x0.shape[0], max(n0, (budget_s * rate0 / iters) // cores * cores)))
    t0 = time.perf_counter()
    tot1, *_ = o.batch_solve(x0[:n1], ui[:n1], nthreads=cores)
    dt1 = time.perf_counter() - t0
    return dict(value=tot1 / dt1, unit="iLQR iterations/s", cores=int(cores), kind="port",
                sample="oracle (CPU restatement, " + ("rigid-stance plant, exact (forward-mode AD) Jacobians of the constrained step as on the GPU" if contact else "forward-difference Jacobians as in the reference; its forward-mode-AD variant is ~2x slower") + ") on the first %d rollouts of the same batch, %d fixed iterations each, OpenMP over rollouts, %.1f s; compiler flags: %s" % (n1, iters, dt1, flags))


def csrc_hash():
    """sha256 over the kernel sources (csrc/*.hip, *.h, Makefile): a PMC traffic record describes THIS build only."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mpc-ilqr-mujoco_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.cpp")) + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def run_signature(args, n_slices):
    """What a PMC traffic record must have been collected on to describe this run (tools/pmc_summary.py --stamp): the
    configuration AND the kernel sources (a kernel change without a re-collection must not inherit the old traffic figure)."""
    return {"csrc_sha": csrc_hash(), "batch": args.batch, "horizon": args.horizon, "iters": args.iters, "contact": bool(args.contact), "slices": int(n_slices),
            "backward": os.environ.get("ILQR_BACKWARD", "wave"), "ls": os.environ.get("ILQR_LS", "s"), "rollout": os.environ.get("ILQR_ROLLOUT", "s"),
            "dyn": os.environ.get("ILQR_DYN", ""), "lint": os.environ.get("ILQR_LINT", "0")}


def kernel_groups(args, B, N, n_slices):
    """Per kernel group: stage keys, algorithmic flops and HBM bytes of ONE full-batch launch."""
    D = 8.0
    Bl = B / n_slices      # rollouts per kernel launch (a solve is enqueued slice by slice, include/ilqr_hip.h)
    bk_env = os.environ.get("ILQR_BACKWARD", "wave")
    bk_name = {"wg": "k_backward_mfma", "va": "k_backward"}.get(bk_env[:2], "k_backward_pack" if riccati_variant() == "packed" else "k_backward_wave")
    primal = "k_lin_primal_r" if os.environ.get("ILQR_ROLLOUT", "s")[:1] == "r" and not args.contact else "k_lin_primal_s"
    one_knot = os.environ.get("ILQR_LINT", "0") == "1"
    tangent = ("k_lin_tangent_c" if one_knot else "k_lin_tangent2c") if args.contact else ("k_lin_tangent" if one_knot else "k_lin_tangent2")
    return {
        bk_name: dict(stages=["iLQR_backwardPass", "iLQR_backwardPass_retry"], unit="fp64 MFMA",
                      flops=RICCATI_FLOPS_PER_KNOT * N * Bl,
                      bytes=D * Bl * (N * (2601 + 969 + 2601 + 51 + 19 + 19 + 969 + 19) + 2 * (2601 + 51))),
        ("k_line_search_r" if os.environ.get("ILQR_LS", "s")[:1] == "r" else "k_line_search_s+k_traj_knot_cost"):
                      dict(stages=["iLQR_lineSearch", "iLQR_lineSearch_retry"], unit="fp64 VALU",
                      flops=ALPHA_TRIAL_FLOPS_PER_KNOT * N * Bl,      # the accepted alpha's trial is the algorithmic work
                      bytes=D * Bl * N * ((969 + 19 + 51 + 19) + 8 * (51 + 19))),
        # (group names = the kernels' names as rocprofv3 / profiles/traffic_latest.json list them, template arguments stripped)
        ("k_fd_steps_s+k_fd_finish" if JAC_FD else primal + "+" + tangent): dict(stages=["iLQR_linearization"], unit="fp64 VALU",
                                     flops=(FD_STEPS_PER_KNOT * STEP_FLOPS if JAC_FD else JACOBIAN_FLOPS_PER_KNOT) * N * Bl,
                                     bytes=D * Bl * N * ((70 + 2601 + 969) if JAC_FD else (70 + 493 + 493 + 70 + 2601 + 969))),      # x, u, (dump written + read,) A_t, B_t
        "k_quad_kin+k_cost_quadratics": dict(stages=["iLQR_costQuadratics"], unit="fp64 VALU",
                                  flops=QUAD_FLOPS_PER_KNOT * (N + 1) * Bl,
                                  bytes=D * Bl * (N + 1) * (70 + 2601 + 51 + 19 + 19)),           # x, u, lxx, lx, lu, luu (the per-knot record between the two kernels is not algorithmic)
        "k_rollout_s": dict(stages=["iLQR_computeCost+forwardRollout"], unit="fp64 VALU",
                            flops=STEP_FLOPS * N * Bl, bytes=D * Bl * N * (51 + 19 + 51)),
    }


def dominant_group(kernels, stage_ms):
    """(name, stage keys) of the kernel group with the largest exclusive-equivalent device time: the linearisation pair, the cost
    quadratics and the nominal re-rollout run on three streams at the same time, so their spans are scaled by (longest span /
    sum of spans), i.e. the region is counted once and split in proportion."""
    total = {n: sum(stage_ms.get(x, 0.0) for x in k["stages"]) for n, k in kernels.items()}
    conc = [n for n in total if n.startswith("k_lin_") or n.startswith("k_quad_kin") or n.startswith("k_rollout")]
    tot = sum(total[n] for n in conc)
    scale = (max(total[n] for n in conc) / tot) if tot > 0 else 1.0
    excl = {n: total[n] * (scale if n in conc else 1.0) for n in total}
    name = max(excl, key=lambda n: excl[n])
    return name, kernels[name]["stages"]


def stage_bench(args, s, sv, x0_d, ui_d, B, N, world, rank, dev, prob):
    """BASELINE.json configs[1]: forward rollout + Jacobians only (iLQR::forwardRolloutNominal + computeLinearization,
    ilqr.cpp:119-131) over the batch; one step = cold-start rollout of every trajectory + A_t, B_t of every knot."""
    import torch
    import torch.distributed as dist

    def one():
        s.initialize_device(x0_d.data_ptr(), ui_d.data_ptr())     # N rollout steps per trajectory (enqueued on the handle's stream)
        s.stage_linearize()                                       # A_t, B_t for all knots; synchronises the stream
    for _ in range(max(1, args.warmup)):
        one()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    A, Bm = s.linearization()
    assert np.all(np.isfinite(A)) and np.all(np.isfinite(Bm))
    if rank == 0:
        ms = 1e3 * el / args.steps
        by = 8.0 * B * N * (70 + 493 + 493 + 70 + 2601 + 969)
        fl = (JACOBIAN_FLOPS_PER_KNOT + STEP_FLOPS) * N * B
        tf, gbs = fl / (ms * 1e-3) / 1e12, by / (ms * 1e-3) / 1e9
        # the roof this pass sits closer to, as the main path picks it
        roof = ({"bound": "mfma", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS} if tf / FP64_PEAK_TFLOPS >= gbs / HBM_PEAK_GBS
                else {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS})
        roof.update({"frac_compute": tf / FP64_PEAK_TFLOPS, "frac_hbm": gbs / HBM_PEAK_GBS, "traffic": None,
                     "note": "whole step (rollout + primal dump + tangent sweeps), host-timed; algorithmic flops = one dynamics step + one analytic (A_t, B_t) "
                             "per knot (FLOP_SOURCE below), algorithmic bytes = trajectory + primal dump round trip + A_t + B_t"})
        is_cfg1 = (B == 1024 and N == 25)
        print(json.dumps({
            "metric": "rollout + Jacobian passes/sec (H1 nx=51 nu=19 N=%d)" % N, "value": world * B * args.steps / el, "unit": "trajectories/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[1]: " if is_cfg1 else "custom (shape of BASELINE.json configs[1] at another size): ") +
                                   "batch=%d H1 standing rollouts, N=%d, forward rollout + analytic Jacobians only, gravity %s" % (B, N, list(prob["gravity"])),
                       "batch_per_gpu": B, "horizon": N, "flop_source": FLOP_SOURCE},
            "roofline": roof}))
    s.close()
    if world > 1:
        dist.destroy_process_group()


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) started WITHOUT a launcher: this process never touches a GPU; it starts
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child, relays rank 0's JSON line and exits with the
    child's code.  One rank per GPU needs N visible devices (unless --rehearse-single-gpu); if the ranks cannot be started the
    exit code is non-zero -- a plain `--gpus 8` must never time ONE rank and print n_gpus: 1 (mpc.cpp:97-113 is the consumer
    the per-step gather serves)."""
    import socket
    import subprocess
    import torch                       # device_count() does not initialise the GPU on this image (task statement)
    have = torch.cuda.device_count()
    if not args.rehearse_single_gpu and have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d needs %d visible GPUs, this node shows %d; refusing to time fewer ranks than asked for "
                         "(use --rehearse-single-gpu to rehearse the N > 1 path on one device)\n" % (args.gpus, args.gpus, have))
        return 2
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["ILQR_BENCH_SELF_LAUNCHED"] = "1"
    sys.stderr.write("bench.py: --gpus %d without a launcher: starting %s\n" % (args.gpus, " ".join(cmd)))
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in child.stdout.splitlines() if l.startswith("{")]
    for l in child.stdout.splitlines():
        if not l.startswith("{"):
            sys.stderr.write(l + "\n")
    if child.returncode == 0 and len(lines) == 1 and json.loads(lines[0]).get("n_gpus") == args.gpus:
        print(lines[0])
        return 0
    sys.stderr.write("bench.py: the %d-rank run did not produce its line (child exit code %d, %d JSON lines)\n" % (args.gpus, child.returncode, len(lines)))
    return child.returncode if child.returncode != 0 else 3


def main():
    args = parse()
    if args.print_signature:
        slices = max(1, int(os.environ.get("ILQR_SLICES", "1")))
        print(json.dumps(run_signature(args, slices), sort_keys=True))
        return
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    global JAC_FD
    JAC_FD = args.jacobians == "fd"
    if JAC_FD:
        args.no_fd_line = True
    if args.workload != "default":
        if args.gpus != 1:
            raise SystemExit("--workload config3 / config4 are the one-GPU anchors (the whole global batch on one device); the 8-GPU points are --gpus 8 with the default workload")
        args.batch, args.horizon = (32768, 25) if args.workload == "config3" else (8192, 50)
        args.no_contact_line = True
        if args.workload == "config4":
            args.contact = True          # (contact mode 2 with the walking problem's own gravity, see below)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: refusing to time a different number of ranks than asked for" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    if args.rehearse_single_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_single_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge
    pkg = ge._load_package()
    from mpc_ilqr_mujoco_amd import sharding as sh
    from mpc_ilqr_mujoco_amd import solver as sv
    sc = pkg.scenario
    B, N, iters = args.batch, args.horizon, args.iters

    if args.workload == "config4":
        # the same builder the -m gpu test of configs[4] uses (tests/test_gpu_configs.py walking_problem): 8192 windows, per-rollout sets
        from mpc_ilqr_mujoco_amd import references as rf
        prob, x0, ui, _t0 = sc.walking_batch(B, N, args.seed, os.path.join(ROOT, "tests", "golden", "refdata_golden.npz"), sv, rf)
        x0, ui = np.ascontiguousarray(x0), np.ascontiguousarray(ui)
        lo, hi = 0, B
    else:
        prob = sc.make_problem(sv.reference_kinematics, N=N, gravity=(0.0, 0.0, -9.81)) if args.contact else sc.make_problem(sv.reference_kinematics, N=N)
        ug = sv.gravity_compensation(sc.standing_state(), prob["gravity"])
        # ONE seeded global batch of world x B rollouts; rank r owns the contiguous range shard_range gives it,
        # so a rollout's inputs -- and therefore its results -- do not depend on the number of GPUs
        x0g, uig = sc.synthetic_batch(B * world, N, args.seed, ug)
        lo, hi = sh.shard_range(B * world, rank, world)
        x0, ui = np.ascontiguousarray(x0g[lo:hi]), np.ascontiguousarray(uig[lo:hi])
        del x0g, uig
    s = sv.BatchedILQR(B, N=N, dt=prob["dt"], device=local_rank)
    s.set_problem(prob)
    s.set_max_iterations(iters)
    if args.contact:
        s.set_contact_mode(2)
    jac_mode = sv.JAC_FD_FORWARD if JAC_FD else sv.JAC_ANALYTIC
    s.set_options(jacobian_mode=jac_mode, fd_eps=1e-5, early_exit=False)
    s.enable_profiling(True)

    dev = torch.device("cuda", local_rank)
    x0_d = torch.from_numpy(x0).to(dev)
    ui_d = torch.from_numpy(ui).to(dev)
    # The ONE collective of an MPC step is the product's own: ilqr_hip_gather_first_knot packs the rows
    # [u0 | cost | (K0)] on the device and gathers them on rank 0 as grouped RCCL send/recv behind the C ABI.  The process group
    # the driver's launcher set up only carries the 128-byte RCCL id from rank 0 to the other ranks (and the barrier / timing
    # reduction).  One GPU: the same entry point degenerates to a device copy.  --rehearse-single-gpu (two ranks on ONE device,
    # which RCCL refuses): every rank runs the one-rank form and the rows are staged through the host over gloo.
    W = sh.payload_width(args.gather_gains)
    rccl, rccl_error = False, None
    if world > 1 and (not args.rehearse_single_gpu or args.try_rccl):
        # Every rank reports whether its communicator came up; unless ALL did, every rank falls back together to the harness's
        # own gather below (torch.distributed) -- the driver's scaling run must not die on a communicator problem, and the JSON
        # says which collective carried the rows.
        cdev = "cpu" if args.rehearse_single_gpu else dev
        # (ncclCommInitRank is collective: first make sure EVERY rank can load librccl, or the others would wait in it for ever)
        avail = torch.tensor([1 if sv.BatchedILQR.comm_available() else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(avail, op=dist.ReduceOp.MIN)
        ok = 1
        try:
            if int(avail.item()) != 1:
                raise RuntimeError("librccl cannot be loaded on every rank")
            idt = torch.zeros(128, dtype=torch.uint8, device=cdev)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(sv.BatchedILQR.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, src=0)
            # ncclCommInitRank is collective: if it fails on ONE rank, its peers stay inside it.  It therefore runs on a helper
            # thread with a deadline; a rank that is still inside after the deadline cannot join the all-ranks fallback safely
            # (the handle is mid-call), so it ends the whole run loudly -- the launcher then takes the other ranks down.
            import threading
            box = {}

            def _init():
                try:
                    s.comm_init(world, rank, bytes(idt.cpu().numpy().tobytes()))
                except Exception as ee:  # noqa: BLE001
                    box["err"] = ee
            th = threading.Thread(target=_init, daemon=True)
            th.start()
            th.join(float(os.environ.get("ILQR_COMM_INIT_TIMEOUT_S", "180")))
            if th.is_alive():
                sys.stderr.write("bench.py rank %d: ilqr_hip_comm_init (ncclCommInitRank) did not return within its deadline; aborting the run\n" % rank)
                sys.stderr.flush()
                os._exit(4)
            if "err" in box:
                raise box["err"]
        except Exception as e:  # noqa: BLE001 -- reported in the JSON line
            ok, rccl_error = 0, repr(e)
        flag = torch.tensor([ok], dtype=torch.int32, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        rccl = int(flag.item()) == 1
        if not rccl:
            s.comm_destroy()
            if rank == 0 and rccl_error is None:
                rccl_error = "a peer rank failed to initialise its RCCL communicator"
    if rccl:
        recv = torch.zeros(world * B, W, dtype=torch.float64, device=dev) if rank == 0 else None
    else:
        s.comm_init(1, 0)
        recv = torch.zeros(B, W, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()

    stage_ms, stage_n = {}, {}          # every stage, from ONE untimed step after the timed region (all event pairs on)
    timed_ms, timed_n = {}, {}          # the dominant kernel group only, from the timed steps themselves

    if args.stage == "rollout_jacobians":
        return stage_bench(args, s, sv, x0_d, ui_d, B, N, world, rank, dev, prob)

    def record_stages(ms_acc, n_acc):
        ms, n = s.stage_ms()
        for k in ms:
            ms_acc[k] = ms_acc.get(k, 0.0) + ms[k]
            n_acc[k] = n_acc.get(k, 0.0) + n[k]

    def one_step(timed):
        s.initialize_device(x0_d.data_ptr(), ui_d.data_ptr())
        s.solve_async()
        # the gather follows the solve on the handle's own stream: one host synchronisation per MPC step
        s.gather_first_knot(None if recv is None else recv.data_ptr(), root=0, with_gains=args.gather_gains)
        s.synchronize()
        g = recv
        if world > 1 and not rccl:      # one-GPU rehearsal (host-staged over gloo), or the fallback when RCCL did not come up
            g = sh.gather_first_knot(recv.cpu() if args.rehearse_single_gpu else recv, dst=0)
            # the harness's gather reads `recv` on torch's stream; the next step's ilqr_hip_gather_first_knot rewrites it on the
            # handle's stream -- order the two
            torch.cuda.synchronize()
        if timed:
            record_stages(timed_ms, timed_n)
        return g

    # Event pairs around every launch cost time themselves (all eight stages: 1.3 ms of a 95 ms step), so the timed steps carry
    # them only for the kernel group the roofline is about.  Which group that is comes from one untimed probe step with every
    # stage timed; the full per-stage table of the JSON line comes from one more untimed step after the timed region.
    one_step(False)
    probe_ms, probe_n = {}, {}
    record_stages(probe_ms, probe_n)
    dom_stages = dominant_group(kernel_groups(args, B, N, max(1, s.num_slices())), probe_ms)[1]
    s.set_profiled_stages(dom_stages)
    for _ in range(args.warmup):
        one_step(False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    it_done = s.iterations()
    assert np.all(it_done == iters), "fixed-iteration mode must run exactly --iters iterations per rollout"
    cost = s.cost()
    assert np.all(np.isfinite(cost))
    # one more (untimed) step whose gathered payload is checked: rank 0 holds world x B rows in global rollout order; it also
    # carries the event pairs of every stage (stage_ms_per_step, kernels.* of the JSON line)
    s.set_profiled_stages(None)
    gathered = one_step(False)
    record_stages(stage_ms, stage_n)
    gather_check = None
    if rank == 0:
        g = gathered.cpu().numpy()
        assert g.shape == (world * B, sh.payload_width(args.gather_gains)) and np.all(np.isfinite(g))
        assert np.array_equal(g[lo:hi, 19], s.cost()) and np.array_equal(g[lo:hi, :19], s.ubar()[:, 0])
        gather_check = "rank 0 received %d rows in global rollout order" % g.shape[0]

    # second number: the same MPC step with the reference's convergence exit enabled -- iterations
    # actually executed per second; not the headline value
    s.set_options(jacobian_mode=jac_mode, fd_eps=1e-5, early_exit=True)
    s.enable_profiling(False)
    one_step(False)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ee_iters = 0.0
    for _ in range(max(1, args.steps // 2)):
        one_step(False)
        ee_iters += float(s.iterations().sum())
    ee_spec = s.speculative_iterations()     # of the last step: iterations whose lambda retry ran beside the first pass (passes of <= 512 rollouts)
    torch.cuda.synchronize()
    ee_elapsed = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([ee_elapsed, -ee_iters], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ee_elapsed = float(t[0].item())
        t2 = torch.tensor([ee_iters], dtype=torch.float64, device=dev)
        dist.all_reduce(t2, op=dist.ReduceOp.SUM)
        ee_iters = float(t2.item())
    ee_steps = max(1, args.steps // 2)

    # third number (not the headline): the same batch under physical gravity with the scheduled feet held in
    # stance (unilateral rigid stance constraints, analytic Jacobians of the constrained step), fixed iterations
    contact_line = None
    if not args.contact and not args.no_contact_line and world == 1:
        probc = sc.make_problem(sv.reference_kinematics, N=N, gravity=(0.0, 0.0, -9.81))
        ugc = sv.gravity_compensation(sc.standing_state(), probc["gravity"])
        x0c, uic = sc.synthetic_batch(B, N, args.seed, ugc)
        s.set_problem(probc); s.set_contact_mode(2); s.set_options(jacobian_mode=sv.JAC_ANALYTIC, early_exit=False)
        x0c_d, uic_d = torch.from_numpy(x0c).to(dev), torch.from_numpy(uic).to(dev)

        def cstep():
            s.initialize_device(x0c_d.data_ptr(), uic_d.data_ptr()); s.solve_async(); s.synchronize()
        cstep()
        torch.cuda.synchronize(); tc0 = time.perf_counter()
        csteps = 2
        for _ in range(csteps):
            cstep()
        torch.cuda.synchronize(); tc = time.perf_counter() - tc0
        assert np.all(s.iterations() == iters) and np.all(np.isfinite(s.cost()))
        contact_line = {"value": B * iters * csteps / tc, "unit": "iterations/s", "ms_per_step": 1e3 * tc / csteps,
                        "workload": "same batch, gravity [0, 0, -9.81], both feet scheduled in stance: unilateral rigid stance constraints in rollout / line search, "
                                    "analytic Jacobians of the constrained step, %d fixed iterations" % iters}
        # ... and with the joint-limit rows of the plant switched on as well (no hinge of this batch leaves its range: what
        # the option costs when nothing is stopped -- its kernels are instantiations of their own and decide the set every step)
        s.set_joint_limits(True)
        cstep()
        torch.cuda.synchronize(); tl0 = time.perf_counter()
        for _ in range(csteps):
            cstep()
        torch.cuda.synchronize(); tlim = time.perf_counter() - tl0
        assert np.all(s.iterations() == iters) and np.all(np.isfinite(s.cost()))
        contact_line["with_joint_limit_rows"] = {"value": B * iters * csteps / tlim, "unit": "iterations/s", "ms_per_step": 1e3 * tlim / csteps}
        s.set_joint_limits(False)
        s.set_contact_mode(0)

    # (not the headline): the default workload -- the constraint-free plant -- with the joint-limit rows of the plant
    # switched on (ilqr_hip_set_joint_limits): what the option costs on the headline's own batch, where no hinge leaves its
    # range (its kernels are instantiations of their own -- k_rollout_s<3>, k_line_search_s<3, .>, k_lin_primal_s<true>, k_lin_tangent2<., true>
    # -- and decide the set of stopped hinges every step)
    limits_line = None
    if not args.contact and not args.no_contact_line and not JAC_FD and world == 1 and args.workload == "default":
        s.set_problem(prob); s.set_contact_mode(0); s.set_options(jacobian_mode=sv.JAC_ANALYTIC, early_exit=False)
        s.set_joint_limits(True)

        def lstep():
            s.initialize_device(x0_d.data_ptr(), ui_d.data_ptr()); s.solve_async(); s.synchronize()
        lstep()
        torch.cuda.synchronize(); tl0 = time.perf_counter()
        lsteps = 2
        for _ in range(lsteps):
            lstep()
        torch.cuda.synchronize(); tlj = time.perf_counter() - tl0
        assert np.all(s.iterations() == iters) and np.all(np.isfinite(s.cost())) and s.adopt_mismatches() == 0
        limits_line = {"value": B * iters * lsteps / tlj, "unit": "iterations/s", "ms_per_step": 1e3 * tlj / lsteps,
                       "workload": "the default batch and gravity (no contact rows) with the joint-limit rows of the plant on (hinge ranges of h1.xml as velocity-level "
                                   "stops, robot_utils.cpp:113-114); no hinge of this batch leaves its range; analytic Jacobians, %d fixed iterations" % iters}
        s.set_joint_limits(False)

    # fourth number (not the headline): the same batch with the REFERENCE'S OWN Jacobian scheme, forward differences
    # with eps 1e-5 (RobotUtils::linearizeDynamicsFD, robot_utils.cpp:120-160: one base step + 70 perturbed steps per knot), fixed
    # iterations -- the like-for-like line against cpu_baseline (the oracle is timed with the same scheme).  Its own roofline: the FD
    # kernel pair's span (HIP events on its launch stream) against 71 dynamics steps per knot.
    fd_line = None
    if not args.contact and not args.no_fd_line and not JAC_FD and world == 1 and args.workload == "default":
        s.set_problem(prob); s.set_contact_mode(0); s.set_options(jacobian_mode=sv.JAC_FD_FORWARD, fd_eps=1e-5, early_exit=False)

        def fstep():
            s.initialize_device(x0_d.data_ptr(), ui_d.data_ptr()); s.solve_async(); s.synchronize()
        fstep()
        s.enable_profiling(True); s.set_profiled_stages(["iLQR_linearization"])      # event pairs around the linearisation's launches only
        torch.cuda.synchronize(); tf0 = time.perf_counter()
        fsteps = 2
        for _ in range(fsteps):
            fstep()
        torch.cuda.synchronize(); tf = time.perf_counter() - tf0
        assert np.all(s.iterations() == iters) and np.all(np.isfinite(s.cost()))
        fms, fn = s.stage_ms()      # (of the last solve: one span per iteration)
        lin_ms = float(fms["iLQR_linearization"]) / max(1.0, float(fn["iLQR_linearization"]))
        fd_flops = FD_STEPS_PER_KNOT * STEP_FLOPS * N * B
        fd_bytes = 8.0 * B * N * (70 + 2601 + 969)
        fd_line = {"value": B * iters * fsteps / tf, "unit": "iterations/s", "ms_per_step": 1e3 * tf / fsteps,
                   "jacobians": "forward differences, eps 1e-5, 71 steps per knot (robot_utils.cpp:120-160); the generic one-wave Riccati kernel (no row structure to fold)",
                   "roofline": {"kernel": "k_fd_steps_s+k_fd_finish", "bound": "fp64 VALU (latency: one wave per SIMD)", "avg_launch_ms": lin_ms,
                                "algorithmic_flops_per_launch": fd_flops, "algorithmic_bytes_per_launch": fd_bytes,
                                "frac_compute": fd_flops / (lin_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if lin_ms > 0 else None,
                                "frac_hbm": fd_bytes / (lin_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if lin_ms > 0 else None,
                                "note": "span of the pair's launches inside the solve (beside the cost quadratics and the re-rollout on their own streams)"}}
        s.enable_profiling(False); s.set_profiled_stages(None)
        s.set_options(jacobian_mode=sv.JAC_ANALYTIC, early_exit=False)

    # fifth number (not the headline): the default workload with ilqr_hip_set_dedup_saturated_retry -- lambda retries whose lambda is
    # already at its cap repeat the failed pass bit for bit (ilqr.cpp:619-644 recomputes them) and are skipped; every observable of the
    # solve is unchanged (-m gpu test).  What a user of MPC::stepOnce can have; the headline executes every pass the reference executes.
    dedup_line = None
    if not args.contact and not JAC_FD and world == 1 and args.workload == "default" and not args.no_fd_line:
        s.set_problem(prob); s.set_contact_mode(0); s.set_options(jacobian_mode=sv.JAC_ANALYTIC, early_exit=False)
        s.set_dedup_saturated_retry(True)

        def dstep():
            s.initialize_device(x0_d.data_ptr(), ui_d.data_ptr()); s.solve_async(); s.synchronize()
        dstep()
        torch.cuda.synchronize(); td0 = time.perf_counter()
        dsteps = 2
        for _ in range(dsteps):
            dstep()
        torch.cuda.synchronize(); td = time.perf_counter() - td0
        assert np.all(s.iterations() == iters) and np.all(np.isfinite(s.cost()))
        dedup_line = {"value": B * iters * dsteps / td, "unit": "iterations/s", "ms_per_step": 1e3 * td / dsteps,
                      "note": "NOT the headline: same workload, same results bit for bit, with the lambda retries whose lambda is already saturated (min(10 lambda, 1e-3) == lambda: "
                              "an exact repeat of the pass that has just failed) skipped -- ilqr_hip_set_dedup_saturated_retry, off by default"}
        s.set_dedup_saturated_retry(False)

    if rank == 0:
        total_iters = float(world) * B * iters * args.steps
        value = total_iters / elapsed
        # dominant kernel = largest device time over the timed region (HIP events on the streams the kernels are
        # launched on); the lambda-retry launches of backward pass / line search are the same kernels on a subset
        # of the rollouts, so they count towards the kernel's total but the roofline uses the full-batch launches
        # per kernel group: stage keys, algorithmic flops and HBM bytes of ONE full-batch launch
        n_slices = max(1, s.num_slices())
        kernels = kernel_groups(args, B, N, n_slices)
        # measured HBM traffic: a replay of the builder's own rocprofv3 --pmc passes (profiles/traffic_latest.json), valid
        # only for the run it was collected on -- the file carries the signature of that run and is ignored otherwise
        traffic_file = os.path.join(ROOT, "profiles", "traffic_latest.json")
        measured = json.load(open(traffic_file)) if os.path.exists(traffic_file) else {}
        sig = run_signature(args, n_slices)
        stamp = measured.get("_stamp")
        traffic_reason = None
        if not measured:
            traffic_reason = "profiles/traffic_latest.json not present"
        elif stamp is None or any(stamp.get(k) != v for k, v in sig.items()):
            traffic_reason = "profiles/traffic_latest.json was collected on another configuration (%s), this run is %s" % (json.dumps(stamp, sort_keys=True), json.dumps(sig, sort_keys=True))
        dom_name = dominant_group(kernels, probe_ms)[0]       # the group whose launches carried event pairs in the timed steps
        table = {}
        for name, k in kernels.items():
            # the dominant group: HIP events of the timed steps; the others: the one untimed step with every stage timed
            src_ms, src_n, nstep = (timed_ms, timed_n, args.steps) if name == dom_name else (stage_ms, stage_n, 1)
            total = sum(src_ms.get(x, 0.0) for x in k["stages"])
            full = k["stages"][0]                      # the full-batch launches (retry launches run a subset)
            avg_ms = src_ms.get(full, 0.0) / max(src_n.get(full, 0.0), 1.0)
            tf = k["flops"] / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            gbs = k["bytes"] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            n_all = sum(src_n.get(x, 0.0) for x in k["stages"])
            table[name] = dict(total_ms_per_step=total / nstep, avg_launch_ms=avg_ms, launches=src_n.get(full, 0.0),
                               avg_launch_ms_all_passes=(total / n_all if n_all > 0 else 0.0),   # first + lambda-retry launches, as rocprofv3 --stats averages them
                               tflops=tf, frac_compute=tf / FP64_PEAK_TFLOPS, gbs=gbs, frac_hbm=gbs / HBM_PEAK_GBS,
                               compute_unit=k["unit"], algorithmic_flops_per_launch=k["flops"], algorithmic_bytes_per_launch=k["bytes"])
        # The linearisation pair, the cost quadratics and the nominal re-rollout run on three streams at the same time: their
        # spans overlap, so the sum of the three is more than the device time the region takes.  "Largest device time" is taken
        # on exclusive-equivalent time: the spans of the concurrent groups are scaled by (longest span / sum of spans), i.e.
        # the region is counted once and split in proportion (rocprofv3's per-kernel totals, profiles/, rank the same way
        # once the contention-inflated durations of these kernels are set against their stand-alone times).
        conc = [n for n in table if n.startswith("k_lin_") or n.startswith("k_quad_kin") or n.startswith("k_rollout")]
        tot = sum(table[n]["total_ms_per_step"] for n in conc)
        scale = (max(table[n]["total_ms_per_step"] for n in conc) / tot) if tot > 0 else 1.0
        for n in table:
            table[n]["exclusive_ms_per_step"] = table[n]["total_ms_per_step"] * (scale if n in conc else 1.0)
        dom_kernel = dom_name
        d = table[dom_kernel]
        # the roof the dominant kernel sits closer to: fp64 compute (78.6 TFLOP/s, vector == matrix peak) or HBM
        if d["frac_compute"] >= d["frac_hbm"]:
            roof = {"bound": "mfma", "achieved": d["tflops"], "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": d["frac_compute"]}
        else:
            roof = {"bound": "hbm", "achieved": d["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["frac_hbm"]}
        members = [measured.get(k) for k in dom_kernel.split("+")]     # a '+'-joined group: the sum of its member kernels
        if traffic_reason is None and all(m is not None for m in members):
            roof["traffic"] = sum(2.0 * 1024.0 * m["FETCH_SIZE_KiB"] + 1024.0 * m["WRITE_SIZE_KiB"] for m in members)
            roof["traffic_source"] = "profiles/traffic_latest.json (builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command: %s)" % json.dumps(stamp, sort_keys=True)
        else:
            roof["traffic"] = None
            roof["traffic_source"] = traffic_reason or "no PMC record for %s" % dom_kernel
        # whole-iteration view (SURVEY.md 8(d) budget per rollout-iteration x iterations/s, against the roofs of `world` GPUs)
        it_scale = N / 25.0
        roof["whole_iteration_frac"] = value * ITER_FLOPS_N25 * it_scale / (FP64_PEAK_TFLOPS * 1e12 * world)
        roof["whole_iteration_frac_hbm"] = value * ITER_BYTES_N25 * it_scale / (HBM_PEAK_GBS * 1e9 * world)
        roof["whole_iteration_frac_counted_flops"] = value * ITER_FLOPS_COUNTED_N25 * it_scale / (FP64_PEAK_TFLOPS * 1e12 * world)
        # what a fixed-mode iteration EXECUTES: the reference's lambda retry (ilqr.cpp:619-644) repeats the backward pass and the line
        # search for the rollouts whose first search failed -- late iterations, nearly all of them.  Its share is read off the profiled
        # step: device time of the retry backward passes / of the first passes (the kernel's time is proportional to the rollouts it is given)
        bw1, bw2 = stage_ms.get("iLQR_backwardPass", 0.0), stage_ms.get("iLQR_backwardPass_retry", 0.0)
        retry_share = bw2 / bw1 if bw1 > 0 else 0.0
        executed = (ITER_FLOPS_COUNTED_N25 + retry_share * 25.0 * (RICCATI_FLOPS_PER_KNOT + ALPHA_TRIAL_FLOPS_PER_KNOT)) * it_scale
        roof["retry_share_of_first_pass"] = retry_share
        roof["executed_flops_per_iteration"] = executed
        roof["whole_iteration_frac_executed_flops"] = value * executed / (FP64_PEAK_TFLOPS * 1e12 * world)
        if dom_kernel in ("k_backward_wave", "k_backward_pack") and d["avg_launch_ms"] > 0:
            # `achieved` / `frac` are ALGORITHMIC-equivalent rates (914 786 flop per knot, the dense minimal-reuse count); the kernel
            # issues fewer, padded products: MFMA count x 2048 flops is what the hardware executes
            variant = riccati_variant()
            ex = MFMA_PER_KNOT[variant] * 2048.0 * N * (B / n_slices)
            roof["executed_mfma_flops_per_launch"] = ex
            roof["frac_executed_mfma"] = ex / (d["avg_launch_ms"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS
            roof["riccati_variant"] = variant
        roof.update({"kernel": dom_kernel, "compute_unit": d["compute_unit"], "avg_launch_ms": d["avg_launch_ms"], "launches": d["launches"],
                     "avg_launch_ms_all_passes": d["avg_launch_ms_all_passes"],
                     "algorithmic_flops_per_launch": d["algorithmic_flops_per_launch"],
                     "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
                     "frac_compute": d["frac_compute"], "frac_hbm": d["frac_hbm"],
                     "kernel_total_ms_per_step": d["total_ms_per_step"], "kernel_exclusive_ms_per_step": d["exclusive_ms_per_step"],
                     "note": "achieved / frac = ALGORITHMIC flops (or bytes) of one launch / its average duration -- an algorithmic-equivalent rate, not a count of issued "
                             "instructions (executed_mfma_flops_per_launch / frac_executed_mfma give the issued MFMA flops of the Riccati kernel); "
                             "whole_iteration_frac = iterations/s x SURVEY 8(d)'s 27.8 MFLOP (2.43 MB) per rollout-iteration against the fp64 (HBM) roof, "
                             "whole_iteration_frac_counted_flops the same with the op-counted per-stage figures (config.flop_source: 33.0 MFLOP at N = 25), whole_iteration_frac_executed_flops with the lambda-retry passes a fixed-mode iteration really executes added (executed_flops_per_iteration = counted + retry_share_of_first_pass x (Riccati + line-search trial)); "
                             "full-batch launches of the timed steps only (HIP events on the launch stream; this kernel group was picked on an untimed probe "
                             "step with every stage timed, and is the only one that carries event pairs inside the timed region); traffic = 2 x FETCH_SIZE + WRITE_SIZE of a full-batch launch from "
                             "separate rocprofv3 --pmc passes of this command (see traffic_source), null when no record matches this run; "
                             "linearisation, cost quadratics and nominal re-rollout overlap on three streams: their spans include contention, the dominant "
                             "kernel is chosen on exclusive-equivalent time (kernels.*.exclusive_ms_per_step)"})
        out = {
            "metric": "iLQR iterations/sec (H1 nx=51 nu=19 N=%d)" % N, "value": value, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload_label(args, B, N, iters, world, prob["gravity"]),
                       "flop_source": FLOP_SOURCE,
                       "batch_per_gpu": B, "global_batch": B * world, "horizon": N, "iterations_per_solve": iters,
                       "jacobians": "analytic (constrained step)" if args.contact else "analytic", "contact_mode": bool(args.contact), "batch_slices": n_slices, "gather": "u0+cost" + ("+K0" if args.gather_gains else ""), "gather_check": gather_check,
                       "collective": ("ilqr_hip_gather_first_knot, one rank: device copy (no RCCL)" if world == 1 else
                                      "RCCL grouped send/recv behind the C ABI (ilqr_hip_gather_first_knot)" if rccl else
                                      "ilqr_hip_gather_first_knot per rank + gloo, host-staged (one-GPU rehearsal)" if args.rehearse_single_gpu else
                                      "FALLBACK: ilqr_hip_gather_first_knot per rank + torch.distributed.gather (RCCL communicator did not come up)"),
                       "collective_error": rccl_error,
                       "barrier": ("none (one rank)" if world == 1 else
                                   "torch.distributed process group, backend %s%s: barrier before and after the timed region and the max-over-ranks all_reduce of the elapsed time; "
                                   "the product's own RCCL communicator (ilqr_hip_comm_init) carries only the per-step gather" %
                                   (dist.get_backend(), " (= RCCL on ROCm)" if dist.get_backend() == "nccl" else "")),
                       "launcher": "self-launched by `python bench.py --gpus N` (child torch.distributed.run)" if os.environ.get("ILQR_BENCH_SELF_LAUNCHED") else
                                   ("torch.distributed.run (WORLD_SIZE from the environment)" if "WORLD_SIZE" in os.environ else "single process")},
            "roofline": roof,
            "kernels": {n: {k: (round(v, 6) if isinstance(v, float) else v) for k, v in t.items() if k in
                            ("total_ms_per_step", "exclusive_ms_per_step", "avg_launch_ms", "frac_compute", "frac_hbm")} for n, t in table.items()},
            "stage_ms_per_step": {k: stage_ms[k] for k in stage_ms},
            "stage_ms_source": "one untimed step after the timed region with event pairs around every stage (inside the timed region only the roofline kernel is timed: "
                               "the event records of all eight stages cost about 1.3 ms per step)",
            "early_exit": {"value": ee_iters / ee_elapsed, "unit": "iterations/s", "mean_iterations_per_solve": ee_iters / (world * B * ee_steps),
                           "ms_per_step": 1e3 * ee_elapsed / ee_steps, "speculative_retry_iterations_last_step": ee_spec,
                           "note": "same step with the reference's convergence exit (|dJ| < 1e-4) enabled: executed iterations per second; not the headline; "
                                   "once at most 512 rollouts are still active the lambda retry of ilqr.cpp:619-644 runs beside the first pass "
                                   "(same results, ilqr_hip_get_speculative_iterations; ILQR_SPEC=0 for the sequential order)"},
        }
        if contact_line is not None:
            out["contact"] = contact_line
        if limits_line is not None:
            out["joint_limits"] = limits_line
        if fd_line is not None:
            out["fd"] = fd_line
        if dedup_line is not None:
            out["dedup_saturated_retry"] = dedup_line
        if args.workload == "config4":
            out["cpu_baseline"] = None     # (the oracle's OpenMP batch shares ONE reference set; per-rollout windows are checked rollout by rollout in the -m gpu tests)
        elif not args.no_cpu_baseline and world == 1:   # the CPU baseline is timed on rank 0 of the one-GPU run only
            try:
                out["cpu_baseline"] = cpu_baseline(pkg, prob, x0, ui, iters, args.cpu_seconds, contact=args.contact)
            except Exception as e:  # the oracle is optional test infrastructure
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    s.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
