"""Developer experiment (GPU): barrier starts of the warm-started grid levels (ASCENT_NESTED_MU=first,next) over the whole config-3
sweep -- kernel time and the per-level iteration maxima (at 4096 NLPs every SIMD holds one wavefront: the slowest one is the time)."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass()
pairs = [tuple(float(x) for x in a.split(",")) for a in sys.argv[1:]] or [(1e-6, 1e-9), (1e-7, 1e-10), (1e-7, 3e-10), (1e-6, 3e-10), (3e-7, 1e-9), (1e-6, 2e-9), (1e-6, 5e-10), (3e-6, 1e-9)]
for mf, mn in pairs:
    os.environ["ASCENT_NESTED_MU"] = f"{mf},{mn}"
    ms = [A.solve_batch(S, 200, tol=1e-9, want_traj=False).kernel_ms for _ in range(6)]
    r = A.solve_batch(S, 200, tol=1e-9, want_traj=False)
    tot = r.iters.astype(int)
    u60 = A.solve_batch(S, 60, tol=1e-3, want_traj=False).iters.astype(int)     # (a 60-node solve nests 17 -> 60 with mu first)
    u17 = A.solve_batch(S, 17, tol=1e-3, want_traj=False, coarse_nodes=-1).iters.astype(int)
    fine, mid = tot - u60, u60 - u17
    print(f"mu {mf:g},{mn:g}: kernel ms min {min(ms):.3f} | converged {(r.status == 0).sum()} | 60-level mean {mid.mean():.2f} max {mid.max()} | fine mean {fine.mean():.2f} hist from {fine.min()} {np.bincount(fine)[fine.min():].tolist()}", flush=True)
