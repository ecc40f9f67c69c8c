"""Developer check: other tolerances through the default dispatch (and, PIPE=split, the split pipeline) against the C restatement."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle as O
if os.environ.get("PIPE"):
    os.environ["ASCENT_PIPELINE"] = os.environ["PIPE"]
S = A.sweep_config4()[::4099][:64]
for tol in (1e-4, 1e-6, 1e-8, 1e-9, 1e-10, 1e-11, 1e-12):
    for scheme in (0, 1):
        r = A.solve_batch(S, 200, tol=tol, scheme=scheme, max_iter=500)
        o = O.solve_batch(S, 200, 500, tol, scheme=scheme); O.set_scheme(0)
        print(f"tol {tol:g} scheme {scheme}: status gpu {np.bincount(r.status, minlength=4)} oracle {np.bincount(o['status'], minlength=4)} iters gpu {r.iters.min()}-{r.iters.max()} oracle {o['iters'].min()}-{o['iters'].max()} "
              f"max |diters| {np.abs(r.iters.astype(int) - o['iters']).max()} max |dtf| {np.abs(r.tf - o['tf']).max():.1e}", flush=True)
