"""Hermite-Simpson persistent kernel: one NLP per wavefront against four, over batch sizes (N = 200)."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
S0 = A.sweep_isp_drymass()
for B in (64, 256, 512, 768, 1024, 1536, 2048, 4096):
    S = np.ascontiguousarray(S0[:: 4096 // B][:B]) if 4096 % B == 0 else np.ascontiguousarray(S0[:B])
    row = []
    for wide in ("1", "0"):
        os.environ["ASCENT_PERSIST_WIDE"] = wide
        A.solve_batch(S, 200, tol=1e-9, scheme=2, want_traj=False)
        r = A.solve_batch(S, 200, tol=1e-9, scheme=2, want_traj=False)
        row.append(f"{r.kernel_ms:7.2f} ms" + ("" if (r.status == 0).all() else "!"))
    print(f"B={B:5d}  one NLP per wavefront {row[0]}   four {row[1]}", flush=True)
