import sys; sys.path.insert(0, ".")
import numpy as np, json
import lunar_module_ascent_trajectory_optimiser_amd as A
fx = json.load(open("tests/golden/hs_fixtures.json"))
c = [c for c in fx["cases"] if c["nt"] == 2000][0]
print("fixture tf", c["tf"], c["iters"])
for tol in (1e-8, 1e-9, 1e-10, 1e-11, 1e-12):
    for cn in (0, -1):
        r = A.solve_batch(A.AscentParams(), 2000, tol=tol, scheme=2, terminal="ellipse", max_iter=800, coarse_nodes=cn)
        print(tol, "coarse", cn, r.status, r.iters, repr(r.tf[0]), r.tf[0] - c["tf"])
