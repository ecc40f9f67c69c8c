import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle as co
S = A.sweep_isp_drymass(4, 4)
for nt in (30, 63, 64, 65, 100, 128, 150, 333, 700):
    g = A.solve_batch(S, nt, tol=1e-9, max_iter=500)
    o = co.solve_batch(S, nt, 500, 1e-9)
    s = A.solve_batch(S, nt, tol=1e-9, max_iter=500, coarse_nodes=-1)
    print(f"nt={nt:4d}: gpu status {np.bincount(g.status, minlength=1)} iters {g.iters.min()}-{g.iters.max()} | oracle iters {o['iters'].min()}-{o['iters'].max()} equal {np.array_equal(g.iters, o['iters'])} | max rel tf diff gpu-oracle {np.abs(g.tf-o['tf']).max()/o['tf'].max():.1e} | vs single grid {np.abs(g.tf-s.tf).max():.1e} (single iters {s.iters.min()}-{s.iters.max()}) kernel {g.kernel_ms:.1f} vs {s.kernel_ms:.1f} ms")
