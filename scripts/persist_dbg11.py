import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass()
P = S[2467:2468]; K = 199
os.environ["ASCENT_PIPELINE"] = "persist"
q = A.solve_batch(P, 200, tol=1e-9, max_iter=22, want_blob=True)
g = np.ascontiguousarray(q.blob)
o = {}
for mode in ("split", "persist"):
    os.environ["ASCENT_PIPELINE"] = mode
    o[mode] = A.solve_batch(P, 200, tol=1e-12, max_iter=1, guess=g, warm_start=2, mu_init=1e-10, want_blob=True)
a, b = o["split"].blob[:, 0], o["persist"].blob[:, 0]
for name, lo, hi, w in (("z", 0, 7*K, 7), ("u", 7*K, 8*K, 1), ("lam", 8*K, 15*K, 7), ("zb", 15*K, 21*K, 6)):
    da = (a[lo:hi] - g[lo:hi, 0]).reshape(K, w); db = (b[lo:hi] - g[lo:hi, 0]).reshape(K, w)
    print(name, "split step max", np.abs(da).max(0), "persist step max", np.abs(db).max(0))
    dd = np.abs(da - db)
    print("    diff max", dd.max(0), "at node", dd.argmax(0))
print("scal split", a[21*K:] - g[21*K:, 0]); print("scal persist", b[21*K:] - g[21*K:, 0])
