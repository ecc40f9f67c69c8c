"""BASELINE config 5 as a batch through the persistent Hermite-Simpson kernel and through the dense-block path: N = 2000, terminal 1
(periapsis of the ellipse) and terminal 2 (burnout anywhere on the ellipse)."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A

for term in ("ellipse", "ellipse_free"):
    for B in (1, 16, 256):
        S = np.vstack([A.AscentParams().as_row()[None], A.sweep_isp_drymass(16, 16)])[:B]
        res = {}
        for path in ("persist", "dense"):
            if path == "dense" and term == "ellipse_free" and B == 256:
                continue
            os.environ["ASCENT_PIPELINE"] = path
            A.solve_batch(S, 2000, tol=1e-9, scheme=2, terminal=term, max_iter=500, want_traj=False)
            r = A.solve_batch(S, 2000, tol=1e-9, scheme=2, terminal=term, max_iter=500)
            o = r.orbit()
            res[path] = r
            print(f"terminal {term:12s} B={B:3d} {path:8s}: {r.kernel_ms:8.2f} ms  status {np.bincount(r.status, minlength=4)} iters {r.iters.min()}-{r.iters.max()} "
                  f"t_f[0] {r.final_time()[0]:.6f}  orbit err {np.abs(o['periapsis_alt'] - 17703).max():.1e} {np.abs(o['apoapsis_alt'] - 88615).max():.1e} m", flush=True)
        if len(res) == 2:
            print(f"      max |tf persist - dense| {np.abs(res['persist'].tf - res['dense'].tf).max():.2e}")
del os.environ["ASCENT_PIPELINE"]
