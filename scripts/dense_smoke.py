"""Quick GPU check of the dense-block path: Newton step vs the C oracle (schemes 0/1), solves vs the hand-tuned path, HS."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle

nt = 60; K = nt - 1
S = A.sweep_isp_drymass(2, 2)
for scheme in (0, 1):
    blobs = []
    for b, row in enumerate(S):
        r = c_oracle.solve_batch(row[None], nt, 3 + b, 1e-9, want_blob=True, coarse_nodes=-1, scheme=scheme)
        blobs.append(r["blob"][0])
    blobs = np.stack(blobs, 1)
    mu = np.array([0.1, 0.02, 1e-3, 0.05]); dw = np.array([0.0, 0.0, 1e-2, 1.0])
    step, inertia = A.kkt_step(S, blobs, mu, dw, nt, path="dense", scheme=scheme)
    for b in range(4):
        rc, ref = c_oracle.newton_step(S[b], nt, np.ascontiguousarray(blobs[:, b]), mu[b], dw[b], scheme=scheme)
        d = np.abs(step[:, b] - ref)
        print("scheme", scheme, "nlp", b, "inertia", inertia[b], rc, "max diff z/u", d[:8*K].max(), "lam", d[8*K:15*K].max(), "zb", d[15*K:21*K].max(), "scal", d[21*K:].max(), "ref max", np.abs(ref).max())
c_oracle.set_scheme(0)
for scheme in (0, 1, 2):
    t = time.time()
    r = A.solve_batch(S, 200, tol=1e-9, scheme=scheme, path="dense")
    print("dense solve scheme", scheme, r.status, r.iters, r.final_time(), f"{time.time()-t:.2f}s kernel {r.kernel_ms:.1f} ms")
    if scheme < 2:
        h = A.solve_batch(S, 200, tol=1e-9, scheme=scheme)
        print("   hand-tuned      ", h.status, h.iters, np.abs(h.tf - r.tf).max())
r = A.solve_batch(A.AscentParams(), 200, tol=1e-9, scheme=2, terminal="ellipse")
print("HS ellipse:", r.status, r.iters, r.final_time(), r.orbit()["periapsis_alt"], r.orbit()["apoapsis_alt"])
c = r.coast(100)
print("coast:", c["tf"] * 470, c["periapsis_alt"], c["apoapsis_alt"], "end radius alt", np.hypot(c["traj"][0, -1] * 17703, c["traj"][1, -1] * 17703 + 1738100) - 1738100,
      "r.v end", c["traj"][0, -1] * c["traj"][2, -1] + (c["traj"][1, -1] + 1738100 / 17703) * c["traj"][3, -1])
