import os, sys
import numpy as np
sys.path.insert(0, "tests")
import torch  # noqa
import test_gpu_configs as T
ol = T.ol
B = 12
prob, x0, ui = T.standing(B, seed=53)
os.environ["ILQR_SPEC"] = sys.argv[1] if len(sys.argv) > 1 else "0"
s = T._solver(B); s.set_problem(prob); s.set_options(early_exit=False); s.set_max_iterations(10)
s.initialize(x0, ui); cost = s.solve(x0)
tc, ta, tl = s.trace()
o = ol.Oracle(prob["N"], prob["dt"]); o.set_problem(prob); o.set_options(max_iter=10, early_exit=0)
for b in range(B):
    o.initialize(x0[b], ui[b]); ok, c = o.solve(x0[b])
    n, oc, oa, olam = o.trace()
    print(b, "alpha gpu", ta[b], "orc", oa)
    print("   cost rel", np.abs(tc[b] - oc) / np.abs(oc))
    print("   lam gpu", tl[b], "orc", olam)
