import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass()
os.environ["ASCENT_PIPELINE"] = "split"
r = A.solve_batch(S, 200, want_traj=False)
os.environ["ASCENT_DEBUG_ROUNDS"] = "1"
q = A.solve_batch(S, 200, want_traj=False)
it, rd = r.iters, q.iters
print("iters  hist", np.bincount(it)[20:])
print("rounds hist", np.bincount(rd)[20:])
extra = rd - it
print("extra rounds hist", np.bincount(extra))
worst = np.argsort(rd)[-12:]
for w in worst: print("problem", w, "isp idx", w // 64, "dry idx", w % 64, "iters", it[w], "rounds", rd[w])
