import os, sys
sys.path.insert(0, ".")
import numpy as np
from lunar_module_ascent_trajectory_optimiser_amd import _lib
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass()[::64][:64]
os.environ["ASCENT_PIPELINE"] = "persist"
r = A.solve_batch(S, 200, tol=1e-9)
w = int(np.argmax(r.iters)); print("worst", w, r.iters[w], "alone:")
# alone, and in its wave-mates' company
for sel in ([w], list(range(4 * (w // 4), 4 * (w // 4) + 4))):
    q = A.solve_batch(S[sel], 200, tol=1e-9)
    print(sel, q.iters, q.status)
