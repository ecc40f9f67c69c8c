import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle as O
B, nt = 130, 37
S = A.sweep_isp_drymass()[:: max(1, 4096 // B)][:B]
os.environ["ASCENT_PIPELINE"] = "persist"
b = A.solve_batch(S, nt, tol=1e-9, scheme=1, max_iter=500)
ref = O.solve_batch(S, nt, 500, 1e-9, scheme=1); O.set_scheme(0)
bad = np.nonzero(np.abs(b.iters.astype(int) - ref["iters"]) > 1)[0]
print("bad", bad, b.iters[bad], ref["iters"][bad])
i = bad[0]
K = nt - 1
# follow the oracle's path and compare the Newton steps of the persistent kernel at its iterates
for j in range(2, int(ref["iters"][i]) + 1, 2):
    r = O.solve_batch(S[i:i+1], nt, j, 1e-9, want_blob=True, coarse_nodes=-1, scheme=1); O.set_scheme(0)
    blob = r["blob"][0]
    for mu in (1e-1, 1e-3, 1e-6, 1e-9):
        step, inertia = A.kkt_step(S[i:i+1], blob[:, None], mu, 0.0, nt, path="persist", scheme=1)
        rc, rs = O.newton_step(S[i], nt, blob, mu, 0.0, scheme=1); O.set_scheme(0)
        err = np.abs(step[:, 0] - rs).max() / max(1.0, np.abs(rs).max())
        print(f"iterate after {j} its, mu {mu:g}: inertia {inertia[0]}/{rc} rel err {err:.1e}")
