"""Developer tool: throughput of the solver paths (split pipeline with one-lane / 16-lane sweeps, fused kernel) over batch sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
sizes = [int(x) for x in sys.argv[1:]] or [64, 1024, 4096, 8192, 16384, 32768, 65536]
for B in sizes:
    n = int(round(B ** 0.5))
    while B % n: n -= 1
    S = A.sweep_isp_drymass(n, B // n)
    row = [f"{B:6d}"]
    for mode, fac in (("split", "lane"), ("split", "wide"), ("fused", "lane")):
        os.environ["ASCENT_PIPELINE"] = mode
        os.environ["ASCENT_FACTOR"] = fac
        A.solve_batch(S, 200, want_traj=False)
        r = A.solve_batch(S, 200, want_traj=False)
        row.append(f"{mode}/{fac} {r.kernel_ms:8.1f} ms {B/(r.kernel_ms*1e-3):10.0f} NLP/s conv {int(r.converged.sum())}")
    print(" | ".join(row), flush=True)
