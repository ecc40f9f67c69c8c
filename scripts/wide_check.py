"""Developer tool: the 16-lanes-per-NLP factorisation (ASCENT_FACTOR=wide) against the one-lane-per-NLP one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A

S = A.sweep_isp_drymass()
os.environ["ASCENT_PIPELINE"] = "split"
res = {}
for mode in ("lane", "wide"):
    os.environ["ASCENT_FACTOR"] = mode
    A.solve_batch(S[:256], 200, want_traj=False)
    t = []
    for _ in range(3):
        t0 = time.perf_counter(); r = A.solve_batch(S, 200, want_traj=False); t.append(time.perf_counter() - t0)
    res[mode] = r
    print(mode, "status", np.bincount(r.status), "iters mean %.3f max %d" % (r.iters.mean(), r.iters.max()),
          "kernel ms %.2f" % A.last_kernel_ms(), "wall ms %.2f" % (1e3 * min(t)), flush=True)
a, b = res["lane"], res["wide"]
print("tf max rel diff %.3e" % np.max(np.abs(a.tf - b.tf) / a.tf), "iters differ on", int((a.iters != b.iters).sum()), "of", len(a.iters))
from lunar_module_ascent_trajectory_optimiser_amd.params import pack
P1 = pack([A.AscentParams(r_peri=53108.4, r_apo=53108.4, mass_scalar=2576.0)] * 3)
out = {}
for mode in ("lane", "wide"):
    os.environ["ASCENT_FACTOR"] = mode
    out[mode] = A.solve_batch(P1, 200, want_traj=False, formulation="v1", max_iter=500)
print("v1", out["lane"].final_time()[0], out["wide"].final_time()[0], out["lane"].iters[0], out["wide"].iters[0], out["wide"].status)
