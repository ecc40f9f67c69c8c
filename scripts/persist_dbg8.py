import os, sys
sys.path.insert(0, ".")
import numpy as np
from lunar_module_ascent_trajectory_optimiser_amd import _lib
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass()
os.environ["ASCENT_PIPELINE"] = "persist"
r = A.solve_batch(S, 200, tol=1e-9, want_traj=False)
print("hist", np.bincount(r.iters)[20:40], "worst", np.argsort(r.iters)[-5:], np.sort(r.iters)[-5:], "status", np.bincount(r.status))
w = int(np.argmax(r.iters))
_lib._lib = None
_lib.LIB_PATH = os.path.abspath("dbglib/libascent_dbg.so")
q = A.solve_batch(S[w:w + 1], 200, tol=1e-9, max_iter=40)
print(q.iters, q.status)
