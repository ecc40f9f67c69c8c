import sys, os
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle as O
full = A.sweep_config4()
S = np.ascontiguousarray(full[::8])
r = A.solve_batch(S, 200, tol=1e-9, scheme=1, want_traj=False)
print("config-4 every 8th, trapezoid, persist:", np.bincount(r.status, minlength=4), "iters", r.iters.min(), r.iters.mean(), r.iters.max(), f"{r.kernel_ms:.1f} ms -> {len(S)/r.kernel_ms:.0f}k NLPs/s")
idx = np.linspace(0, len(S) - 1, 32).astype(int)
o = O.solve_batch(S[idx], 200, 300, 1e-9, scheme=1); O.set_scheme(0)
print("vs oracle: iters equal", np.array_equal(r.iters[idx], o["iters"]), "max dtf", np.abs(r.tf[idx] - o["tf"]).max())
r2 = A.solve_batch(S[:64], 2000, tol=1e-9, scheme=1, want_traj=False, max_iter=500)
o2 = O.solve_batch(S[:4], 2000, 500, 1e-9, scheme=1); O.set_scheme(0)
print("N=2000 trapezoid 64 NLPs:", np.bincount(r2.status, minlength=4), r2.iters.min(), r2.iters.max(), f"{r2.kernel_ms:.1f} ms", "vs oracle", r2.iters[:4], o2["iters"], np.abs(r2.tf[:4] - o2["tf"]).max())
