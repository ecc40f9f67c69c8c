"""Three solves of 1024 config-3 problems with the l1 move penalty (ascent_opts.move_penalty = 1, dense-block path) and nothing else:
the command behind profiles/*kernel_stats_move_penalty*.csv (rocprofv3 --kernel-trace --stats -- python3 scripts/prof_move_penalty.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass()[::4].copy()
S[:, 15] = 1e-5
for _ in range(3):
    r = A.solve_batch(S, 200, tol=1e-9, want_traj=False, move_penalty=True, max_iter=500)
print("path", A.default_path(len(S), 200, move_penalty=True), "iters", r.iters.mean(), "converged", int((r.status == 0).sum()), "kernel ms", r.kernel_ms)
