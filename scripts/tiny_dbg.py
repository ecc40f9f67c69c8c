import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle
for nt in (3, 4, 6, 12):
    S = A.sweep_isp_drymass(1, 1)
    ref = c_oracle.solve_batch(S, nt, 300, 1e-8)
    for mode in ("riccati", "pcr"):
        os.environ["ASCENT_DENSE_NEWTON"] = mode
        r = A.solve_batch(S, nt, tol=1e-8, path="dense")
        print(nt, mode, r.status, r.iters, r.tf, "oracle", ref["status"], ref["iters"], ref["tf"])
