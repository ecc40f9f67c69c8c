"""Developer tool: time the config-3 solve with a variant build of the library (LIB=path/to/lib.so, B=batch)."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
from lunar_module_ascent_trajectory_optimiser_amd import _lib
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["LIB"])
import lunar_module_ascent_trajectory_optimiser_amd as A
for B in [int(b) for b in os.environ.get("B", "4096").split(",")]:
    S = A.sweep_isp_drymass()[:: max(1, 4096 // B)][:B]
    os.environ["ASCENT_SMALL_BATCH"] = "off"
    mp = os.environ.get("MP", "0") == "1"          # MP=1: with the move penalty (the reference's DCOST = 1e-5)
    S[:, 15] = 1e-5
    ms = []
    for i in range(8):
        r = A.solve_batch(S, 200, tol=1e-9, want_traj=False, move_penalty=mp)
        ms.append(r.kernel_ms)
    print(f"{os.environ.get('LIB', 'default'):24s} B={B:5d} kernel ms min {min(ms):.3f} median {np.median(ms):.3f} iters {r.iters.min()}-{r.iters.max()} converged {(r.status == 0).sum()}", flush=True)
