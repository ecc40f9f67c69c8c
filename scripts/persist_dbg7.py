import os, sys
sys.path.insert(0, ".")
import numpy as np
from lunar_module_ascent_trajectory_optimiser_amd import _lib
_lib.LIB_PATH = os.path.abspath("dbglib/libascent_dbg.so")
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass()[::64][:64]
os.environ["ASCENT_PIPELINE"] = "persist"
r = A.solve_batch(S[15:16], 200, tol=1e-9)
print(r.iters, r.status)
