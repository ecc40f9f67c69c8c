import sys, numpy as np, time
sys.path.insert(0, ".")
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_config4(); S[:, 15] = 1e-5
for mp in (True, False):
    t = time.time()
    r = A.solve_batch(S, 200, tol=1e-9, want_traj=False, move_penalty=mp)
    print("config 4 whole box, move_penalty", mp, ": converged", (r.status == 0).sum(), "of", len(S), "status counts", np.bincount(r.status), "iters", r.iters.min(), r.iters.mean(), r.iters.max(), "kernel ms %.1f" % r.kernel_ms, "NLPs/s %.0f" % (len(S) / r.kernel_ms * 1e3))
