"""Throughput of the default dispatch over batch sizes (N=200, backward Euler, cold start, tol 1e-9), with and without the reference's
DCOST: kernel ms and NLPs/s (round 3: the persistent kernel at every size; one NLP per wavefront up to 1024 NLPs)."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A

full = A.sweep_config4(); full[:, 15] = 1e-5
S0 = A.sweep_isp_drymass(); S0[:, 15] = 1e-5
print(f"{'batch':>7s} {'DCOST applied':>26s} {'DCOST not applied':>26s}")
for B in (1, 4, 16, 64, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 262144):
    S = S0[:: max(1, 4096 // B)][:B] if B <= 4096 else np.ascontiguousarray(full[:: len(full) // B][:B])
    row = []
    for mp in (True, False):
        A.solve_batch(S, 200, tol=1e-9, want_traj=False, move_penalty=mp)
        r = A.solve_batch(S, 200, tol=1e-9, want_traj=False, move_penalty=mp)
        ok = int((r.status == 0).sum())
        row.append(f"{r.kernel_ms:9.2f} ms {ok / r.kernel_ms:8.1f}k/s" + ("" if ok == B else "!"))
    print(f"{B:7d} " + " ".join(f"{x:>26s}" for x in row), flush=True)
