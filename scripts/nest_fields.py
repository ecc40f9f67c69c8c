import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass(9, 8)
a = A.solve_batch(S, 200, tol=1e-9, coarse_nodes=-1)
b = A.solve_batch(S, 200, tol=1e-9)
names = "x y xdot ydot xdd ydd angle angledot u mass".split()
for f, n in enumerate(names):
    d = np.abs(a.traj[f] - b.traj[f]); print(f"{n:9s} max diff {d.max():.3e}  at node {np.unravel_index(d.argmax(), d.shape)[0]}  scale {np.abs(a.traj[f]).max():.3e}")
c = A.solve_batch(S, 200, tol=1e-12, coarse_nodes=-1, max_iter=500); d_ = A.solve_batch(S, 200, tol=1e-12, max_iter=500)
print("tol 1e-12: status", np.bincount(c.status), np.bincount(d_.status), "u max diff", np.abs(c.traj[8]-d_.traj[8]).max(), "angle", np.abs(c.traj[6]-d_.traj[6]).max(), "tf", np.abs(c.tf-d_.tf).max())
print("u diff between tol 1e-9 and 1e-12 single grid:", np.abs(a.traj[8]-c.traj[8]).max())
