"""Shader cycles per phase of the persistent kernel (diagnostic -DPERSIST_PROFILE build in dbglib/), wavefront 0, batch 4096."""
import os, sys
sys.path.insert(0, ".")
from lunar_module_ascent_trajectory_optimiser_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("LIB", "dbglib/libascent_dbg.so"))
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass()[:int(os.environ.get("B", "4096"))]
os.environ["ASCENT_PIPELINE"] = "persist"
A.solve_batch(S, 200, tol=1e-9, want_traj=False)
r = A.solve_batch(S, 200, tol=1e-9, want_traj=False)
print("kernel ms", r.kernel_ms, "iters", r.iters[:4])
