import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass()
t = A.solve_batch(S, 200, want_traj=False)
c = A.solve_batch(S, 18, tol=1e-3, want_traj=False, coarse_nodes=-1)
f = t.iters - c.iters
print("coarse hist", np.bincount(c.iters)[6:]); print("fine hist", dict(zip(*np.unique(f, return_counts=True))))
os.environ["ASCENT_DEBUG_ROUNDS"] = "1"
r = A.solve_batch(S, 200, want_traj=False)
print("fine-level rounds hist", dict(zip(*np.unique(r.iters, return_counts=True))))
