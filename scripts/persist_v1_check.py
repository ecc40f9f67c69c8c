"""Developer check: the v1 formulation through the persistent kernel against the split pipeline and the C restatement."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle as O
base = A.AscentParams(r_peri=53108.4, r_apo=53108.4, mass_scalar=2576.0)
for B, nt in ((1, 200), (6, 200), (64, 200), (130, 37), (4096, 200), (9, 1000)):
    n = int(round(B ** 0.5)); 
    while B % n: n -= 1
    S = A.sweep_isp_drymass(n, B // n, base=base)
    out = {}
    for mode in ("split", "persist"):
        os.environ["ASCENT_PIPELINE"] = mode
        A.solve_batch(S, nt, tol=1e-9, formulation="v1", want_traj=False, max_iter=500)
        out[mode] = A.solve_batch(S, nt, tol=1e-9, formulation="v1", max_iter=500)
    a, b = out["split"], out["persist"]
    print(f"B={B} nt={nt}: split {a.kernel_ms:.2f} ms, persist {b.kernel_ms:.2f} ms | status {np.bincount(b.status, minlength=4)} (split {np.bincount(a.status, minlength=4)}) iters equal {np.array_equal(a.iters, b.iters)} "
          f"({a.iters.min()}-{a.iters.max()} vs {b.iters.min()}-{b.iters.max()}) max tf diff {np.abs(a.tf - b.tf).max():.2e} traj diff {np.abs(a.traj - b.traj).max():.2e}", flush=True)
    if B <= 130:
        ref = O.solve_batch(S, nt, 500, 1e-9, formulation=1)
        O.set_formulation(0)
        print("   vs oracle: tf", np.abs(b.tf - ref["tf"]).max(), "iters equal", np.array_equal(b.iters, ref["iters"]), "max diff", np.abs(b.iters.astype(int) - ref["iters"]).max(), "nominal tf*470", b.tf[0] * 470)
