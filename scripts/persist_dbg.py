import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
nt = int(os.environ.get("NT", "40")); K = nt - 1
S = A.sweep_isp_drymass(2, 2)[:int(os.environ.get("B", "1"))]
for mi in (1, 2, 3, 4, 6):
    out = {}
    for mode in ("split", "persist"):
        os.environ["ASCENT_PIPELINE"] = mode
        out[mode] = A.solve_batch(S, nt, tol=1e-9, max_iter=mi, coarse_nodes=-1, want_blob=True)
    a, b = out["split"].blob, out["persist"].blob
    d = np.abs(a - b)
    print(f"max_iter {mi}: iters {out['split'].iters} {out['persist'].iters} | z,u {d[:8*K].max():.2e} lam {d[8*K:15*K].max():.2e} zb {d[15*K:21*K].max():.2e} scal {d[21*K:].max(axis=1)}")
