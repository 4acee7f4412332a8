"""Diagnostic: does the -O3 -march=native build of the oracle run on this host?  python tools/probes/native_oracle_probe.py <nthreads> [portable]"""
import faulthandler, os, subprocess, sys, time
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import oracle_lib as ol
import __graft_entry__ as ge
pkg = ge._load_package(); sc = pkg.scenario
nt = int(sys.argv[1]); native = len(sys.argv) < 3
prob = sc.make_problem(ol.reference_kinematics, N=25)
o0 = ol.Oracle(25, 0.02)
ug = o0.grav_comp(sc.standing_state())
x0, ui = sc.synthetic_batch(max(nt, 8), 25, 0, ug)
o = ol.Oracle(25, 0.02, native=native); o.set_problem(prob); o.set_options(max_iter=3, early_exit=0, jac_mode=1, fd_eps=1e-5)
print("flags:", ol.lib_native()[1] if native else "portable", flush=True)
t = time.perf_counter(); tot, cost, *_ = o.batch_solve(x0, ui, nthreads=nt); dt = time.perf_counter() - t
print("threads", nt, "it/s", tot / dt, cost[:2], flush=True)
