import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle
S = A.sweep_isp_drymass()
P = S[2467:2468]; K = 199
os.environ["ASCENT_PIPELINE"] = "persist"
b = {}
for mi in (20, 21, 22, 23):
    b[mi] = np.ascontiguousarray(A.solve_batch(P, 200, tol=1e-12, max_iter=mi, want_blob=True).blob[:, 0])
for mi in (20, 21, 22):
    took = b[mi + 1] - b[mi]
    rc, st = c_oracle.newton_step(P[0], 200, b[mi], 1e-10, 0.0)
    print(f"iteration {mi}->{mi+1}: rc {rc} E0 before {c_oracle.kkt_error(P[0], 200, b[mi], 0.0):.3e}")
    for name, lo, hi, w in (("z", 0, 7*K, 7), ("u", 7*K, 8*K, 1), ("lam", 8*K, 15*K, 7), ("zb", 15*K, 21*K, 6)):
        t = took[lo:hi].reshape(K, w); o = st[lo:hi].reshape(K, w)
        print("   ", name, "took max", np.abs(t).max(0), "\n         oracle max", np.abs(o).max(0), "\n         diff max", np.abs(t - o).max(0), "at", np.abs(t - o).argmax(0))
    print("    scal took", took[21*K:], "\n    oracle  ", st[21*K:])
