"""Shader cycles per phase of the persistent kernel (diagnostic -DPERSIST_PROFILE build in dbglib/), wavefront 0, batch 4096.
MP=1: with the move penalty (the reference's DCOST = 1e-5)."""
import os, sys
sys.path.insert(0, ".")
from lunar_module_ascent_trajectory_optimiser_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("LIB", "dbglib/libascent_dbg.so"))
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass()[:int(os.environ.get("B", "4096"))]
mp = os.environ.get("MP", "0") == "1"
scheme = int(os.environ.get("SCHEME", "0"))      # 2: the Hermite-Simpson kernel (ascent_hs.hip)
S[:, 15] = 1e-5
os.environ["ASCENT_PIPELINE"] = "persist"
A.solve_batch(S, 200, tol=1e-9, want_traj=False, move_penalty=mp, scheme=scheme)
r = A.solve_batch(S, 200, tol=1e-9, want_traj=False, move_penalty=mp, scheme=scheme)
print("move_penalty", mp, "kernel ms", r.kernel_ms, "iters", r.iters[:4])
