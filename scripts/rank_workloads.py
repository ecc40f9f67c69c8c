import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench
import lunar_module_ascent_trajectory_optimiser_amd as A
for world in (2, 4, 8):
    for r in range(world):
        P = bench.rank_params(4096, r, world)
        x = A.solve_batch(P, 200, want_traj=False)
        print(f"world {world} rank {r}: converged {int(x.converged.sum())}/4096 iters {x.iters.min()}-{x.iters.max()} kernel {x.kernel_ms:.2f} ms", flush=True)
