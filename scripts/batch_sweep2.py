"""Throughput of the solver paths over batch sizes (N=200, backward Euler, cold start, tol 1e-9): kernel ms and NLPs/s."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A

full = A.sweep_config4()
print(f"{'batch':>7s} " + " ".join(f"{n:>22s}" for n in ("persist", "split wide", "split lane", "fused")))
for B in (16, 64, 256, 1024, 2048, 4096, 8192, 16384, 24576, 32768, 65536):
    S = A.sweep_isp_drymass()[:B] if B <= 4096 else np.ascontiguousarray(full[:: len(full) // B][:B])
    row = []
    for mode, fac in (("persist", None), ("split", "wide"), ("split", "lane"), ("fused", None)):
        if (mode == "fused" and B < 1024) or (fac == "wide" and B > 16384):
            row.append("-"); continue
        os.environ["ASCENT_PIPELINE"] = mode
        if fac: os.environ["ASCENT_FACTOR"] = fac
        else: os.environ.pop("ASCENT_FACTOR", None)
        A.solve_batch(S, 200, tol=1e-9, want_traj=False)
        r = A.solve_batch(S, 200, tol=1e-9, want_traj=False)
        ok = int((r.status == 0).sum())
        row.append(f"{r.kernel_ms:8.2f} ms {ok / r.kernel_ms:7.1f}k/s" + ("" if ok == B else "!"))
    print(f"{B:7d} " + " ".join(f"{x:>22s}" for x in row), flush=True)
