"""Developer tool: time the config-3 solve with a variant build of the library (LIB=path/to/lib.so), e.g. one compiled with other
constants:  hipcc ... -DNESTED_MU_FIRST_V=1e-7 -o dbglib/v.so"""
import os, sys
sys.path.insert(0, ".")
import numpy as np
from lunar_module_ascent_trajectory_optimiser_amd import _lib
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["LIB"])
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass()
ms = []
for i in range(6):
    r = A.solve_batch(S, 200, tol=1e-9, want_traj=False)
    ms.append(r.kernel_ms)
print(os.environ.get("LIB", "default"), "kernel ms", np.round(ms, 3), "iters", r.iters.min(), r.iters.mean(), r.iters.max(), "converged", (r.status == 0).sum())
