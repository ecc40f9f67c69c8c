import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle
S = A.sweep_isp_drymass()
P = S[2467:2468]; K = 199
os.environ["ASCENT_PIPELINE"] = "persist"
q = A.solve_batch(P, 200, tol=1e-9, max_iter=22, want_blob=True)
blob = np.ascontiguousarray(q.blob[:, 0])
print("persist iterate: oracle E0", c_oracle.kkt_error(P[0], 200, blob, 0.0), "E(mu)", c_oracle.kkt_error(P[0], 200, blob, 1e-10))
rc, st = c_oracle.newton_step(P[0], 200, blob, 1e-10, 0.0)
print("oracle newton step at that iterate: rc", rc, "max |dz,du|", np.abs(st[:8*K]).max(), "max |dlam|", np.abs(st[8*K:15*K]).max(), "dzb", np.abs(st[15*K:21*K]).max(), "scal", st[21*K:])
os.environ.pop("ASCENT_PIPELINE")
s2, inert = A.kkt_step(P, blob[:, None], 1e-10, 0.0, 200, path="split_wide")
print("split_wide step: inertia", inert, "max |dz,du|", np.abs(s2[:8*K]).max(), "max |dlam|", np.abs(s2[8*K:15*K]).max(), "scal", s2[21*K:, 0])
lam = blob[8*K:15*K].reshape(K, 7); dl = st[8*K:15*K].reshape(K, 7)
print("largest dlam at node/field", np.unravel_index(np.abs(dl).argmax(), dl.shape), "lam there", lam[np.unravel_index(np.abs(dl).argmax(), dl.shape)])
zb = blob[15*K:21*K].reshape(K, 6); z = blob[:7*K].reshape(K, 7); u = blob[7*K:8*K]
dist = np.stack([z[:, 4], np.pi/3 - z[:, 4], z[:, 6], 1 - z[:, 6], u + 1, 1 - u], 1)
print("complementarity min/max", (zb * dist).min(), (zb * dist).max(), "min dist", dist.min(0))
