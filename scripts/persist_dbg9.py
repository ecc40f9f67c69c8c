import os, sys
sys.path.insert(0, ".")
import numpy as np
from lunar_module_ascent_trajectory_optimiser_amd import _lib
_lib.LIB_PATH = os.path.abspath("dbglib/libascent_dbg.so")
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle
S = A.sweep_isp_drymass()
os.environ["ASCENT_PIPELINE"] = "persist"
q = A.solve_batch(S[2467:2468], 200, tol=1e-9, max_iter=22, want_blob=True)
print(q.iters, q.status)
os.environ["ASCENT_PIPELINE"] = "split"
r = A.solve_batch(S[2467:2468], 200, tol=1e-9, want_blob=True)
print("split", r.iters, r.status, "oracle E0 of split solution", c_oracle.kkt_error(S[2467], 200, np.ascontiguousarray(r.blob[:, 0]), 0.0), "of persist iterate", c_oracle.kkt_error(S[2467], 200, np.ascontiguousarray(q.blob[:, 0]), 0.0))
