import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import lunar_module_ascent_trajectory_optimiser_amd as A
dev = torch.device("cuda", 0)
P_t = torch.from_numpy(A.sweep_isp_drymass()).to(dev)
out = {}
for mode in ("persist", "split"):
    os.environ["ASCENT_PIPELINE"] = mode
    for cn in (0, -1):
        A.solve_batch_torch(P_t, 200, tol=1e-9, out=out, sync=True, coarse_nodes=cn)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            A.solve_batch_torch(P_t, 200, tol=1e-9, out=out, coarse_nodes=cn)
            t1 = time.perf_counter()
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3, A.last_kernel_ms(0)))
        print(mode, "coarse", cn, "call ms / total ms / kernel ms:", [tuple(round(x, 2) for x in t) for t in ts])
