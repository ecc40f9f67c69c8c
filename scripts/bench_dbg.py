import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import lunar_module_ascent_trajectory_optimiser_amd as A
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
P_t = torch.from_numpy(A.sweep_isp_drymass()).to(dev)
out = {}
conv = torch.zeros((), dtype=torch.int64, device=dev)
for i in range(6):
    t0 = time.perf_counter()
    A.solve_batch_torch(P_t, 200, tol=1e-9, want_traj=True, out=out)
    t1 = time.perf_counter()
    conv += (out["status"] == 0).sum()
    t2 = time.perf_counter()
    k = A.last_kernel_ms(0)
    t3 = time.perf_counter()
    print(i, "solve %.2f ms, count %.2f ms, last_kernel_ms call %.2f ms -> %.2f" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, k))
