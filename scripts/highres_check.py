import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle as co
P = A.AscentParams()
for nt in (400, 2000):
    t = time.time(); o = co.solve_batch(P.as_row()[None], nt, 500, 1e-9); tc = time.time() - t
    t = time.time(); r = A.solve_batch(P, nt, tol=1e-9, max_iter=500); tg = time.time() - t
    print(f"nt={nt}: oracle status {o['status'][0]} iters {o['iters'][0]} t_f {o['tf'][0]*470:.5f} s ({tc:.2f}s) | gpu status {r.status[0]} iters {r.iters[0]} t_f {r.final_time()[0]:.5f} s kernel {r.kernel_ms:.1f} ms | rel diff {abs(r.tf[0]-o['tf'][0])/o['tf'][0]:.1e}")
S = A.sweep_isp_drymass(16, 16)
r = A.solve_batch(S, 2000, tol=1e-9, max_iter=500, want_traj=False)
print("batch 256 @ nt=2000:", np.bincount(r.status, minlength=4), "iters", r.iters.min(), r.iters.mean(), r.iters.max(), "kernel ms", r.kernel_ms)
