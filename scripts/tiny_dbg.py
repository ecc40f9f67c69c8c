"""Developer check: small batches through the automatic dispatch (dense blocks + PCR) vs the hand-tuned kernels vs the CPU restatement."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle as O
S = A.sweep_isp_drymass()[:: 4096 // 16][:8]
for cn in (26, 0, 60, 18):
    for env in ({}, {"ASCENT_SMALL_BATCH": "off"}, {"ASCENT_DENSE_NEWTON": "riccati"}):
        for k in ("ASCENT_SMALL_BATCH", "ASCENT_DENSE_NEWTON"):
            os.environ.pop(k, None)
        os.environ.update(env)
        r = A.solve_batch(S, 201, tol=1e-9, coarse_nodes=cn, path="dense" if "ASCENT_DENSE_NEWTON" in env else "auto")
        print(cn, env, r.iters, r.status, f"{r.kernel_ms:.2f} ms")
    o = O.solve_batch(S, 201, 300, 1e-9, coarse_nodes=cn)
    print(cn, "oracle", o["iters"])
