import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass()[::64][:int(os.environ.get("B", "16"))]
for cn in (-1, 0):
    out = {}
    for mode in ("split", "persist"):
        os.environ["ASCENT_PIPELINE"] = mode
        out[mode] = A.solve_batch(S, 200, tol=1e-9, coarse_nodes=cn, want_blob=True)
    print("coarse", cn, "split iters", out["split"].iters, "persist iters", out["persist"].iters, "status", out["persist"].status)
# which iteration do they part at (single grid, the worst problem)?
os.environ["ASCENT_PIPELINE"] = "persist"
full = A.solve_batch(S, 200, tol=1e-9, coarse_nodes=-1)
w = int(np.argmax(full.iters))
P = S[w:w + 1]
K = 199
for mi in range(6, 40, 2):
    o = {}
    for mode in ("split", "persist"):
        os.environ["ASCENT_PIPELINE"] = mode
        o[mode] = A.solve_batch(P, 200, tol=1e-9, max_iter=mi, coarse_nodes=-1, want_blob=True)
    d = np.abs(o["split"].blob - o["persist"].blob)
    print(f"problem {w} max_iter {mi}: iters {o['split'].iters} {o['persist'].iters} status {o['split'].status} {o['persist'].status} | z,u {d[:8*K].max():.2e} lam {d[8*K:15*K].max():.2e} zb {d[15*K:21*K].max():.2e} th {d[21*K,0]:.2e}")
    if d[:8*K].max() > 1e-6: break
