import os, sys
sys.path.insert(0, '/root/repo')
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass()
os.environ["ASCENT_PIPELINE"] = "split"
A.solve_batch(S[:64], 200, want_traj=False)
os.environ["ASCENT_DEBUG"] = "1"
r = A.solve_batch(S, 200, want_traj=False)
print(A.last_kernel_ms())
