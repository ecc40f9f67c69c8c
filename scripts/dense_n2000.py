"""BASELINE config 5 on the GPU: N = 2000 Hermite-Simpson, ellipse-proper terminal constraints, coast arc."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A

for B in (1, 16, 256):
    S = np.vstack([A.AscentParams().as_row()[None], A.sweep_isp_drymass(16, 16)])[:B]
    t = time.time()
    r = A.solve_batch(S, 2000, tol=1e-9, scheme=2, terminal="ellipse", max_iter=500)
    dt = time.time() - t
    o = r.orbit()
    c = r.coast(2000)
    print(f"B={B}: status {np.bincount(r.status, minlength=4)} iters {r.iters.min()}-{r.iters.max()} t_f[0]={r.final_time()[0]:.6f} "
          f"orbit[0] {o['periapsis_alt'][0]:.3f} x {o['apoapsis_alt'][0]:.3f} m, max orbit err {np.abs(o['periapsis_alt']-17703).max():.2e} "
          f"{np.abs(o['apoapsis_alt']-88615).max():.2e} m; coast {c['tf'][0]*470:.2f} s; wall {dt:.2f} s, kernel {r.kernel_ms:.1f} ms", flush=True)
