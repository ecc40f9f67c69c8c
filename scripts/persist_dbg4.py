import os, sys
sys.path.insert(0, ".")
import numpy as np
from lunar_module_ascent_trajectory_optimiser_amd import _lib
_lib.LIB_PATH = os.path.abspath("dbglib/libascent_dbg.so")
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass()[::64][:16]
os.environ["ASCENT_PIPELINE"] = "persist"
r = A.solve_batch(S[11:12], 200, tol=1e-12, max_iter=30, coarse_nodes=-1)
print(r.iters, r.status)
