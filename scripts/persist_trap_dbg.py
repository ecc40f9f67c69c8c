"""Developer study: accuracy of the persistent kernel's Newton step along the C restatement's path towards the solution
(late iterates are ill-conditioned: sigma = z/s reaches 1e14), per scheme and path."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle as O
nt = 200
S = A.sweep_config4()[::4099][:64]
i = int(sys.argv[1]) if len(sys.argv) > 1 else 0
K = nt - 1
for scheme in (0, 1):
    full = O.solve_batch(S[i:i+1], nt, 500, 1e-11, scheme=scheme); O.set_scheme(0)
    nit = int(full["iters"][0])
    print(f"scheme {scheme}: oracle needs {nit} iterations (all levels)")
    # final-level iterates: warm chain is inside the oracle; take blobs of capped solves on the single fine grid from the converged point
    sol = O.solve_batch(S[i:i+1], nt, 500, 1e-8, want_blob=True, scheme=scheme); O.set_scheme(0)
    blob = sol["blob"][0]
    for mu in (1e-8, 1e-9, 1e-10, 1e-11):
        row = []
        for path in ("persist", "split_wide"):
            step, inertia = A.kkt_step(S[i:i+1], blob[:, None], mu, 0.0, nt, path=path, scheme=scheme)
            rc, rs = O.newton_step(S[i], nt, blob, mu, 0.0, scheme=scheme); O.set_scheme(0)
            segs = ((0, 7 * K), (7 * K, 8 * K), (8 * K, 15 * K), (15 * K, 21 * K), (21 * K, 21 * K + 10))
            err = [np.abs(step[lo:hi, 0] - rs[lo:hi]).max() for lo, hi in segs]
            row.append(" ".join(f"{e:.1e}" for e in err))
        print(f"   mu {mu:g}: |step - oracle| (dz du dl dzb scal)  persist {row[0]} | split_wide {row[1]}   (max |step| {np.abs(rs).max():.1e})")
