"""Developer tool: three config-3 solves for rocprofv3 (ASCENT_FACTOR selects the sweep kernels; PROF_LIB
substitutes a diagnostic build of the library; PROF_MAXIT caps the iterations for timing-only builds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lunar_module_ascent_trajectory_optimiser_amd as A
from lunar_module_ascent_trajectory_optimiser_amd import _lib
if os.environ.get("PROF_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["PROF_LIB"])
S = A.sweep_isp_drymass()[:int(os.environ.get("PROF_BATCH", "4096"))]
os.environ["ASCENT_PIPELINE"] = "split"
for _ in range(3):
    r = A.solve_batch(S, 200, want_traj=False, max_iter=int(os.environ.get("PROF_MAXIT", "500")))
print(r.iters.mean(), A.last_kernel_ms())
