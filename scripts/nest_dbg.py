import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle as co
P = A.AscentParams().as_row()[None]
for nt, nc in ((400, 36), (200, 18)):
    gc = A.solve_batch(P, nc, tol=1e-9, coarse_nodes=-1, want_blob=True)
    oc = co.solve_batch(P, nc, 300, 1e-9, coarse_nodes=-1, want_blob=True)
    print(nt, "coarse iters", gc.iters, oc["iters"], "blob max rel diff", np.abs(gc.blob[:, 0] - oc["blob"][0]).max() / np.abs(oc["blob"][0]).max())
    pg = co.prolong(gc.blob[:, 0], nc, nt); po = co.prolong(oc["blob"][0], nc, nt)
    print("   prolonged diff", np.abs(pg - po).max())
    fo_from_o = co.solve_batch(P, nt, 300, 1e-9, guess_blob=po[None], warm_start=2, mu_init=1e-5)
    fo_from_g = co.solve_batch(P, nt, 300, 1e-9, guess_blob=pg[None], warm_start=2, mu_init=1e-5)
    fg_from_o = A.solve_batch(P, nt, tol=1e-9, guess=po[:, None], warm_start=2, mu_init=1e-5)
    fg_from_g = A.solve_batch(P, nt, tol=1e-9, guess=pg[:, None], warm_start=2, mu_init=1e-5)
    print("   fine iters: oracle(from o) %d oracle(from g) %d gpu(from o) %d gpu(from g) %d" % (fo_from_o["iters"][0], fo_from_g["iters"][0], fg_from_o.iters[0], fg_from_g.iters[0]))
    full_g = A.solve_batch(P, nt, tol=1e-9); full_o = co.solve_batch(P, nt, 300, 1e-9)
    print("   nested total: gpu %d oracle %d" % (full_g.iters[0], full_o["iters"][0]))
