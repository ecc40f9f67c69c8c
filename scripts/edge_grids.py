"""Developer check: unusual grid sizes (partial chunks, tiny grids) through the default dispatch against the C restatement."""
import sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle as O
S = A.sweep_isp_drymass(3, 3)
SCHEME = int(sys.argv[1]) if len(sys.argv) > 1 else 0
bad = 0
for nt in (3, 4, 5, 8, 14, 16, 17, 18, 32, 33, 34, 49, 65, 199, 200, 201, 257, 640):
    r = A.solve_batch(S, nt, tol=1e-9, max_iter=500, scheme=SCHEME)
    o = O.solve_batch(S, nt, 500, 1e-9, scheme=SCHEME); O.set_scheme(0)
    ok = np.array_equal(r.status, o["status"])
    conv = (r.status == 0) & (o["status"] == 0)
    dtf = np.abs(r.tf[conv] - o["tf"][conv]).max() if conv.any() else float("nan")
    dit = np.abs(r.iters.astype(int) - o["iters"]).max()
    flag = "" if ok and (not conv.any() or dtf < 1e-8) else "   <-- CHECK"
    bad += bool(flag)
    print(f"nt {nt:4d} path {A.default_path(len(S), nt, scheme=SCHEME):8s} status gpu {np.bincount(r.status, minlength=4)} oracle {np.bincount(o['status'], minlength=4)} max |dtf| {dtf:.1e} max |diters| {dit}{flag}", flush=True)
print("problems:", bad)
