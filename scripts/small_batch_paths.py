"""Small batches: the hand-tuned sparse path vs the dense-block path with the PCR Newton solve (kernel ms per solve)."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A

print(f"{'nt':>5s} {'B':>4s} {'hand-tuned':>11s} {'dense riccati':>14s} {'dense pcr':>10s}")
for nt in (200, 600, 2000):
    for B in (1, 2, 4, 8, 16, 64):
        P = np.vstack([A.AscentParams().as_row()[None], A.sweep_isp_drymass(16, 8)])[:B]
        row = []
        for path, env in (("auto", None), ("dense", "riccati"), ("dense", "pcr")):
            if env: os.environ["ASCENT_DENSE_NEWTON"] = env
            os.environ["ASCENT_SMALL_BATCH"] = "off"
            A.solve_batch(P, nt, tol=1e-9, path=path, max_iter=500)
            r = A.solve_batch(P, nt, tol=1e-9, path=path, max_iter=500)
            assert np.all(r.status == 0)
            row.append(r.kernel_ms)
            os.environ.pop("ASCENT_DENSE_NEWTON", None)
        print(f"{nt:5d} {B:4d} {row[0]:11.2f} {row[1]:14.2f} {row[2]:10.2f}", flush=True)
