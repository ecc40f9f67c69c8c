import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle
S = A.sweep_isp_drymass()[::64][:16]
P = S[11:12]
for mi in range(20, 34):
    o = {}
    for mode in ("split", "persist"):
        os.environ["ASCENT_PIPELINE"] = mode
        o[mode] = A.solve_batch(P, 200, tol=1e-12, max_iter=mi, coarse_nodes=-1, want_blob=True)
    e = {m: c_oracle.kkt_error(P[0], 200, np.ascontiguousarray(o[m].blob[:, 0]), 0.0) for m in o}
    em = {m: c_oracle.kkt_error(P[0], 200, np.ascontiguousarray(o[m].blob[:, 0]), 1e-10) for m in o}
    print(mi, "iters", o["split"].iters, o["persist"].iters, "E0 split %.3e persist %.3e | E(1e-10) %.3e %.3e" % (e["split"], e["persist"], em["split"], em["persist"]),
          "tf diff %.2e" % abs(o["split"].tf[0] - o["persist"].tf[0]))
