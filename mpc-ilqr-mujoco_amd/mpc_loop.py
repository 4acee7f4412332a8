"""Closed-loop batched MPC on loaded references, with the reference's CSV logs -- SURVEY.md 8(f) row f3.

Mirrors (host side; the solve itself is the HIP library):
  * runSimulation / MPC::stepOnce       /root/reference/main/humanoid_mpc.cpp:126-190, src/ilqr/mpc.cpp:40-127
      per step: reference window -> warm start (shifted previous solution) -> solve -> u = ubar0 + K0 (x - xbar0)
      -> plant step.  The plant here is the same dynamics as the model (`BatchedILQR.step`, or `step_stance` with the
      stance flags of the current schedule row when the solver is in contact mode); the
      reference steps MuJoCo with contacts and clobbers its plant state while solving
       -- neither is reproduced.
  * MPC::initCSVLog / logCurrentStep    src/ilqr/mpc.cpp:181-262   main log, one row per step
  * MPC::logAppliedOptimal              src/ilqr/mpc.cpp:271-343   q_optimal.csv, u_optimal.csv
      (`step,time_sec,q_0..q_25` / `step,time_sec,u_0..u_18`, first knot of the optimised trajectory; the step index
      is the one AFTER the increment in stepOnce, i.e. 1-based), so plotter.py-style tooling keeps working.
  * profiler keys and table             include/common/profiler.hpp, main/humanoid_mpc.cpp:195-226
      `MPCRunner.prof[key]` = list of milliseconds per call for MPC_stepOnce, MPC_extractReference, MPC_warmStart,
      MPC_iLQR_solve, MPC_computeControl (host clock, as the reference measures them) and, with `profile_stages=True`, the
      iLQR_* keys from the device events of the solve; `profiling_table()` formats them as `printProfilingResults` does.
One log set per logged rollout (`log_rollouts`), the reference being single-trajectory.
"""
import os
import time

import numpy as np

NQ, NV, NX, NU = 26, 25, 51, 19


class MPCLogs:
    """The three CSV files of one rollout, in the reference's formats."""

    def __init__(self, directory, dt, log_name="mpc_log.csv"):
        os.makedirs(directory, exist_ok=True)
        self.dt = dt
        self.main = open(os.path.join(directory, log_name), "w")
        self.q = open(os.path.join(directory, "q_optimal.csv"), "w")
        self.u = open(os.path.join(directory, "u_optimal.csv"), "w")
        self.main.write("time_index,time_sec,solve_cost,solve_time_ms" + "".join(",x_%d" % i for i in range(NX)) + "".join(",u_%d" % i for i in range(NU))
                        + "".join(",x_ref_%d" % i for i in range(NX)) + "".join(",u_ref_%d" % i for i in range(NU)) + "\n")
        self.q.write("step,time_sec" + "".join(",q_%d" % i for i in range(NQ)) + "\n")
        self.u.write("step,time_sec" + "".join(",u_%d" % i for i in range(NU)) + "\n")

    @staticmethod
    def _fmt(v):
        return "%.6g" % v          # std::ostream default formatting (6 significant digits)

    def log(self, t_idx, solve_cost, solve_ms, x_measured, u_applied, x_ref0, u_ref0, x_opt0, u_opt0):
        f = self._fmt
        self.main.write(",".join([str(t_idx), f(t_idx * self.dt), f(solve_cost), f(solve_ms)] + [f(v) for v in x_measured] + [f(v) for v in u_applied]
                                 + [f(v) for v in x_ref0] + [f(v) for v in u_ref0]) + "\n")
        self.q.write(",".join([str(t_idx), f(t_idx * self.dt)] + [f(v) for v in x_opt0[:NQ]]) + "\n")
        self.u.write(",".join([str(t_idx), f(t_idx * self.dt)] + [f(v) for v in u_opt0]) + "\n")

    def close(self):
        for fh in (self.main, self.q, self.u):
            fh.flush(); fh.close()


class MPCRunner:
    """Batched closed loop: `solver` = BatchedILQR, `refs` = ReferenceData, `base_problem` = weights etc. (scenario.make_problem)."""

    def __init__(self, solver, refs, base_problem, log_dir=None, log_rollouts=(0,), follow_schedule=False, profile_stages=False):
        self.s, self.refs, self.base = solver, refs, base_problem
        self.prof = {}
        self.profile_stages = profile_stages
        if profile_stages:
            solver.enable_profiling(True)
        self.follow_schedule = follow_schedule
        self.t_idx, self.has_prev = 0, False
        self.logs = {b: MPCLogs(os.path.join(log_dir, "rollout_%d" % b), base_problem["dt"]) for b in log_rollouts} if log_dir else {}
        self.last_cost = None

    def _add(self, key, t_a, t_b):
        self.prof.setdefault(key, []).append(1e3 * (t_b - t_a))

    def profiling_table(self):
        """The table of printProfilingResults (main/humanoid_mpc.cpp:195-226): Function, Calls, Total / Avg / Min / Max in ms."""
        lines = ["", "=== Performance Profiling ===", "", "--- Timing Summary ---",
                 "%-20s%8s%12s%12s%12s%12s" % ("Function", "Calls", "Total(ms)", "Avg(ms)", "Min(ms)", "Max(ms)"), "-" * 76]
        for key in sorted(self.prof):
            t = self.prof[key]
            if t:
                lines.append("%-20s%8d%12.2f%12.2f%12.2f%12.2f" % (key, len(t), sum(t), sum(t) / len(t), min(t), max(t)))
        return "\n".join(lines)

    def step_once(self, x_measured, u_init=None):
        t0 = time.perf_counter()
        prob = self.refs.problem_at(self.t_idx, self.s.N, self.base, follow_schedule=self.follow_schedule)   # extractReferenceWindow
        self.s.set_problem(prob)
        t1 = time.perf_counter(); self._add("MPC_extractReference", t0, t1)
        self.last_stance0 = prob["stance"][0, 0]
        if self.has_prev:
            self.s.initialize_warm_resident(x_measured)       # ilqr.cpp:68-80
        else:
            self.s.initialize(x_measured, u_init)            # cold start, ilqr.cpp:82-116
        t2 = time.perf_counter(); self._add("MPC_warmStart", t1, t2)
        self.last_cost = self.s.solve(x_measured)
        t3 = time.perf_counter(); self._add("MPC_iLQR_solve", t2, t3)
        if self.profile_stages:
            ms, _ = self.s.stage_ms()
            for key, val in (("iLQR_forwardRollout", ms["iLQR_computeCost+forwardRollout"]), ("iLQR_linearization", ms["iLQR_linearization"]),
                             ("iLQR_costQuadratics", ms["iLQR_costQuadratics"]), ("iLQR_backwardPass", ms["iLQR_backwardPass"] + ms["iLQR_backwardPass_retry"]),
                             ("iLQR_lineSearch", ms["iLQR_lineSearch"] + ms["iLQR_lineSearch_retry"])):
                self.prof.setdefault(key, []).append(float(val))
        u = self.s.compute_control(x_measured)                # mpc.cpp:97-101
        t4 = time.perf_counter(); self._add("MPC_computeControl", t3, t4)
        self.has_prev = True
        self.t_idx += 1
        if self.logs:
            ms = 1e3 * (time.perf_counter() - t0)
            xb, ub = self.s.xbar(), self.s.ubar()
            for b, lg in self.logs.items():
                lg.log(self.t_idx, self.last_cost[b], ms, x_measured[b], u[b], prob["x_ref"][0, 0], prob["u_ref"][0, 0], xb[b, 0], ub[b, 0])
        self._add("MPC_stepOnce", t0, time.perf_counter())
        return u

    def run(self, x0, steps, u_init=None):
        """`steps` closed-loop steps from x0 [B,51]; returns the visited states [steps+1,B,51] and controls [steps,B,19]."""
        x = np.array(x0, dtype=np.float64)
        xs, us = [x.copy()], []
        for _ in range(steps):
            u = self.step_once(x, u_init)
            if getattr(self.s, "contact_mode", 0):    # contact row: the plant holds the feet in stance now
                st = np.asarray(self.last_stance0).reshape(-1)
                x = self.s.step_stance(x, u, int(st[0]), int(st[1]))
            else:
                x = self.s.step(x, u)
            xs.append(x.copy()); us.append(u.copy())
        return np.array(xs), np.array(us)

    def close(self):
        for lg in self.logs.values():
            lg.close()
