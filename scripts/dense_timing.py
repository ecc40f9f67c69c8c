"""Developer tool: timings of the dense-block path (Riccati form): Hermite-Simpson and the move penalty over batch sizes (LIB= variant build)."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
from lunar_module_ascent_trajectory_optimiser_amd import _lib
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["LIB"])
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass().copy(); S[:, 15] = 1e-5
os.environ["ASCENT_DENSE_NEWTON"] = "riccati"
for B in (16, 256, 1024, 4096):
    Sb = S[:: 4096 // B][:B]
    for name, kw in (("HS", dict(scheme=2)), ("move penalty", dict(move_penalty=True)), ("BE dense", dict(path="dense"))):
        ms = [A.solve_batch(Sb, 200, tol=1e-9, want_traj=False, max_iter=500, **kw).kernel_ms for _ in range(3)]
        r = A.solve_batch(Sb, 200, tol=1e-9, want_traj=False, max_iter=500, **kw)
        print(f"{os.environ.get('LIB', 'default'):20s} B={B:5d} {name:13s}: {min(ms):8.2f} ms  converged {(r.status == 0).sum()}/{B} iters {r.iters.min()}-{r.iters.max()} tf sum {r.tf.sum():.12f}", flush=True)
r = A.solve_batch(A.AscentParams(), 2000, tol=1e-9, scheme=2, terminal="ellipse", max_iter=500)
ms = [A.solve_batch(np.tile(A.AscentParams().as_row(), (256, 1)), 2000, tol=1e-9, scheme=2, terminal="ellipse", max_iter=500, want_traj=False).kernel_ms for _ in range(2)]
print(f"N=2000 HS ellipse, 256 NLPs (Riccati): {min(ms):.1f} ms", flush=True)
