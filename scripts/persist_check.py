"""Persistent kernel vs the split pipeline: answers, iteration counts, time."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle

for B in (5, 64, 1024, 4096):
    S = A.sweep_isp_drymass()[:: max(1, 4096 // B)][:B]
    out = {}
    for mode in ("split", "persist"):
        os.environ["ASCENT_PIPELINE"] = mode
        A.solve_batch(S, 200, tol=1e-9, want_traj=False)
        t = time.time()
        r = A.solve_batch(S, 200, tol=1e-9)
        out[mode] = (r, time.time() - t)
    a, b = out["split"][0], out["persist"][0]
    print(f"B={B}: split {a.kernel_ms:.2f} ms, persist {b.kernel_ms:.2f} ms | status {np.bincount(b.status, minlength=4)} iters equal {np.array_equal(a.iters, b.iters)} "
          f"({a.iters.min()}-{a.iters.max()} vs {b.iters.min()}-{b.iters.max()}) max tf diff {np.abs(a.tf - b.tf).max():.2e} traj diff {np.abs(a.traj - b.traj).max():.2e}", flush=True)
    if B <= 64:
        ref = c_oracle.solve_batch(S, 200, 300, 1e-9)
        print("   vs oracle: tf", np.abs(b.tf - ref["tf"]).max(), "iters equal", np.array_equal(b.iters, ref["iters"]))
